#!/usr/bin/env python3
"""End-to-end example: the reference's nsga_penalty.py workflow on the MI355X evaluator.

Synthetic 1 s clips -> HIP log-mel front end -> StandardScaler -> constrained NSGA-II
(host) whose fitness evaluations run on the GPU; writes per-generation records (the
reference's column schema) and prints the hypervolume of each generation's population
against one shared reference point (compare.ipynb semantics).

    python examples/run_nsga_penalty.py --pop 4 --gen 2 --clips 2000 --epochs 3
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/run_nsga_penalty.py --pop 40 --gen 20
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_waveforms  # noqa: E402
from cmoop_audio_processing_amd import EvalConfig, PopulationEvaluator, frontend, nsga  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pop", type=int, default=4)
    ap.add_argument("--gen", type=int, default=2)
    ap.add_argument("--clips", type=int, default=2000)
    ap.add_argument("--epochs", type=int, default=300)      # EPOCHS, nsga_penalty.py:177 (early stopping active)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="nsga_generations.csv")
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    wav, y = synth_waveforms(a.clips, 10, 1234, dev)
    feats = frontend.log_mel(wav)
    n_tr, n_va = int(a.clips * 0.8), int(a.clips * 0.1)
    Xtr, Xva = feats[:n_tr].contiguous(), feats[n_tr:n_tr + n_va].contiguous()
    frontend.prepare_dataset(Xtr, Xva, None, mode="refit")
    ev = PopulationEvaluator(Xtr, y[:n_tr], Xva, y[n_tr:n_tr + n_va],
                             EvalConfig.preset("nsga_penalty", epochs=a.epochs, seed=a.seed, verbose=(rank == 0)))
    pareto, hist = nsga.nsga2(ev.compute_objectives_and_constraints, a.pop, a.gen, seed=a.seed)
    if rank == 0:
        nsga.write_records_csv(a.out, hist)
        fronts = [[[-r["Accuracy"], r["Size_MB"], r["FPR"]] for r in h] for h in hist]
        ref = nsga.shared_reference_point(fronts)
        for g, f in enumerate(fronts):
            print(f"generation {g}: hypervolume {nsga.hypervolume(f, ref):.6f}")
        print(f"{len(pareto)} feasible Pareto solutions; {ev.evals_done} true evaluations; records -> {a.out}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
