#!/usr/bin/env python3
"""Step latency of ONE candidate alone on the GPU (the real-protocol tail: with early stopping a generation ends when
its slowest-converging candidate does, and that is often a small net).  Synthetic features of the bench's shape, one
stream, fixed epochs; prints train steps/s and the evaluation's wall time.  Run it under
`rocprofv3 --kernel-trace --stats` to get the sum of kernel durations and the launch count of the same command.

  python tools/lone_candidate.py 16,3,0,1,1,0 [--variant A] [--clips 6000] [--epochs 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmoop_audio_processing_amd import EvalConfig, PopulationEvaluator, genes as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("genes", nargs="+", help="one or more genes f,k,bn,R,fc,dr; each is timed alone")
    ap.add_argument("--variant", default="A")
    ap.add_argument("--clips", type=int, default=6000)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    n_tr, n_va = int(args.clips * 0.8), int(args.clips * 0.1)
    X = torch.randn((n_tr + n_va, 101, 40), device="cuda", generator=g)
    y = (torch.arange(n_tr + n_va, device="cuda") % 10).to(torch.int32)
    cfg = EvalConfig.preset("nsga_penalty", variant=args.variant, epochs=args.epochs, early_stop=False, n_slots=1, seed=0)
    ev = PopulationEvaluator(X[:n_tr], y[:n_tr], X[n_tr:], y[n_tr:], cfg)
    steps = args.epochs * ((n_tr + cfg.batch - 1) // cfg.batch)
    for gs in args.genes:
        gene = tuple(int(v) for v in gs.split(","))
        hp = G.gene_to_hparams(gene)
        best = 1e30
        for _ in range(args.repeat):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev.evaluate_individual(hp)
            best = min(best, time.perf_counter() - t0)
        fl = G.eval_flops(gene, G.VARIANT_NAMES[args.variant], 10, 101, 40, n_tr, n_va, args.epochs, 1)
        print(json.dumps({"gene": gene, "variant": args.variant, "train_steps": steps, "wall_s": round(best, 4),
                          "steps_per_s": round(steps / best, 1), "ms_per_step_incl_val": round(best / steps * 1e3, 4),
                          "tflops": round(fl / best / 1e12, 2)}), flush=True)


if __name__ == "__main__":
    main()
