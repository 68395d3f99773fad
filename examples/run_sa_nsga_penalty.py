#!/usr/bin/env python3
"""End-to-end example: the reference's sa_nsga_penalty.py workflow (surrogate-assisted NSGA-II, optionally the full
memetic method of ablation_study/sa_nsga_local.py) on the MI355X evaluator.

Synthetic 1 s clips with --classes classes -> HIP log-mel front end -> (no scaler: quirk Q2 of sa_nsga_penalty.py:61-85)
-> stratified 50/25/25 split (sa_nsga_penalty.py:71-85) -> SA-NSGA-II on the host: the Kriging surrogate predicts every
offspring, only max(1, int(pop * infill)) of them per generation get a TRUE evaluation on the GPU(s).

    python examples/run_sa_nsga_penalty.py --pop 8 --gen 2 --clips 1200 --epochs 6                       # smoke-sized
    python examples/run_sa_nsga_penalty.py --pop 40 --gen 20 --classes 35                                 # BASELINE configs[2]
    python examples/run_sa_nsga_penalty.py --pop 64 --gen 20 --infill 0.334 --memetic --compute bf16      # BASELINE configs[4]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/run_sa_nsga_penalty.py --pop 40 --gen 20
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_waveforms  # noqa: E402
from cmoop_audio_processing_amd import EvalConfig, PopulationEvaluator, datasets, frontend, nsga, surrogate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pop", type=int, default=8)
    ap.add_argument("--gen", type=int, default=2)
    ap.add_argument("--infill", type=float, default=0.2, help="INFILL_PERCENT, sa_nsga_penalty.py:566")
    ap.add_argument("--clips", type=int, default=1200)
    ap.add_argument("--classes", type=int, default=11, help="11 = the BirdCLEF subset of sa_nsga_penalty.py; 35 = GSC-35")
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--memetic", action="store_true", help="LHS initial population + Lamarckian LCB local search (sa_nsga_local.py:351-433)")
    ap.add_argument("--compute", default="fp32", choices=["fp32", "bf16x3", "bf16"])
    ap.add_argument("--out", default="sa_nsga_generations.csv")
    ap.add_argument("--trace", default="", help="write a JSON trace: per evaluate call wall-clock, epochs run, hypervolume")
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    wav, y = synth_waveforms(a.clips, a.classes, 1234, dev)
    feats = frontend.log_mel(wav).cpu().numpy()                 # [N, 101, 40]; no StandardScaler in this script (Q2)
    Xtr, ytr, Xva, yva, _, _ = datasets.stratified_50_25_25(feats, y.cpu().numpy(), random_state=42)
    preset = "sa_nsga_local" if a.memetic else "sa_nsga_penalty"
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, EvalConfig.preset(preset, classes=a.classes, epochs=a.epochs, seed=a.seed,
                                                                   verbose=(rank == 0), compute=a.compute))
    calls = []
    t_start = time.perf_counter()

    def evaluate(population):
        t0 = time.perf_counter()
        res = ev.compute_objectives_and_constraints(population)
        calls.append({"candidates": len(population), "seconds": round(time.perf_counter() - t0, 3),
                      "wall_clock_s": round(time.perf_counter() - t_start, 3), "epochs_run": list(ev.last_epochs_run)})
        if rank == 0:
            print(f"[search] true evaluation {len(calls)}: {len(population)} candidates in {calls[-1]['seconds']} s",
                  file=sys.stderr, flush=True)
        return res
    pareto, hist, true_evals = surrogate.sa_nsga2(evaluate, a.pop, a.gen, infill_percent=a.infill, seed=a.seed,
                                                  init="lhs" if a.memetic else "random", local_search=a.memetic)
    if rank == 0:
        nsga.write_records_csv(a.out, hist)
        fronts = [[[-r["Accuracy"], r["Size_MB"], r["FPR"]] for r in h] for h in hist]
        ref = nsga.shared_reference_point(fronts)
        hv = [nsga.hypervolume(f, ref) for f in fronts]
        for g, v in enumerate(hv):
            print(f"generation {g}: hypervolume {v:.6f}")
        if a.trace:
            with open(a.trace, "w") as fh:
                json.dump({"pop": a.pop, "gen": a.gen, "infill": a.infill, "memetic": a.memetic, "classes": a.classes,
                           "clips": a.clips, "gpus": world, "compute": a.compute, "true_evaluations": true_evals,
                           "reference_point": [float(v) for v in ref], "hypervolume_per_generation": hv,
                           "evaluate_calls": calls}, fh)
        print(f"{len(pareto)} feasible Pareto solutions; {true_evals} true evaluations of {a.pop * (a.gen + 1)} candidates seen; "
              f"records -> {a.out}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
