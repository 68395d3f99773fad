#!/usr/bin/env python3
"""Predict the strong-scaling curve of bench.py on ONE GPU.

bench.py --gpus N shards the fixed pop=40 generation over N ranks by LPT on closed-form FLOPs and the step
time is the slowest rank's bucket.  The ranks never talk during a step (one all_gather of 5 doubles per
candidate at the end), so the N-GPU step time can be predicted here by timing each rank's bucket alone on the
single GPU of the box:  T_N = max_r T(bucket_r),  speed-up = T_1 / T_N.  The gap to N is (a) FLOP imbalance
(tiny: LPT balances to <1 %), (b) fewer candidates in flight per GPU (a bucket of 4 has at most 4 streams),
(c) the tail where the bucket's biggest candidate runs alone.

usage: python tools/scaling_emulate.py [--clips 30000] [--epochs 2] [--worlds 1,2,4,8]
prints one JSON line.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pop", type=int, default=40)
    ap.add_argument("--clips", type=int, default=30000)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--slots", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--plan", default="lpt", choices=["lpt", "queue"],
                    help="lpt: static LPT buckets (schedule='static'); queue: the default schedule's deterministic deal "
                         "(evaluator.queue_plan) -- only worlds whose whole generation is dealt (n <= world * W) can be emulated, "
                         "each bucket runs with the W worker threads the rank would use")
    args = ap.parse_args()

    import torch
    import bench
    from cmoop_audio_processing_amd import EvalConfig, PopulationEvaluator, frontend, genes as G

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wav, y = bench.synth_waveforms(args.clips, args.classes, 1234, dev)
    feats = frontend.log_mel(wav)
    del wav
    n_tr, n_va = int(args.clips * 0.8), int(args.clips * 0.1)
    Xtr, ytr = feats[:n_tr].contiguous(), y[:n_tr].contiguous()
    Xva, yva = feats[n_tr:n_tr + n_va].contiguous(), y[n_tr:n_tr + n_va].contiguous()
    del feats
    frontend.prepare_dataset(Xtr, Xva, None, mode="refit")
    cfg = EvalConfig.preset("nsga_penalty", variant="A", classes=args.classes, epochs=args.epochs, early_stop=False,
                            seed=args.seed, n_slots=args.slots, profile_every=0)
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    rng = random.Random(args.seed)
    pop = [G.random_hparams(rng) for _ in range(args.pop)]
    out = {"pop": args.pop, "n_train": n_tr, "epochs": args.epochs, "slots": args.slots, "plan": args.plan, "worlds": {}}
    t1 = None
    gl = [G.normalize_hparams(hp) for hp in pop]
    costs = [float(G.fwd_flops_per_sample(g, G.VARIANT_A, args.classes, ev.T, ev.F)) for g in gl]
    for world in [int(w) for w in args.worlds.split(",")]:
        ev_w = ev
        if args.plan == "queue" and world > 1:
            from dataclasses import replace
            from cmoop_audio_processing_amd.evaluator import queue_plan
            _, W, buckets = queue_plan(costs, world, args.slots)
            if sum(len(b) for b in buckets) != len(gl):
                print(f"[emulate] world {world}: {len(gl) - sum(len(b) for b in buckets)} candidates would go through the "
                      "shared counter; not emulated", file=sys.stderr, flush=True)
                continue
            ev_w = PopulationEvaluator(Xtr, ytr, Xva, yva, replace(cfg, n_slots=W))
        else:
            buckets = G.lpt_assign(costs, world)        # the same assignment sharded_map makes on every rank
        times = []
        for r, idx in enumerate(buckets):
            sub = [gl[i] for i in idx]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if sub:
                ev_w.evaluate_genes(sub, [args.seed + i for i in idx])
            torch.cuda.synchronize()
            times.append(round(time.perf_counter() - t0, 3))
            print(f"[emulate] world {world} rank {r}: {len(sub)} candidates, {times[-1]} s", file=sys.stderr, flush=True)
        tn = max(times)
        if world == 1:
            t1 = tn
        out["worlds"][str(world)] = {"bucket_s": times, "bucket_sizes": [len(b) for b in buckets], "step_s": tn,
                                     "evals_per_hour": round(args.pop / tn * 3600, 1),
                                     "speedup_vs_1": round(t1 / tn, 3) if t1 else None}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
