#!/usr/bin/env python3
"""Per-candidate validation accuracy of the bench population under each GEMM arithmetic and under a second
fp32 weight-init seed: how much of the accuracy spread between modes is the run-to-run spread of training
itself (different init / dropout streams), and how much the arithmetic.

usage: python tools/mode_accuracy_report.py [--clips 30000] [--epochs 2]   -> one JSON line
"""
import argparse
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pop", type=int, default=40)
    ap.add_argument("--clips", type=int, default=30000)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--runs", default="fp32:0,fp32:1000,bf16x3:0,bf16:0", help="compute:seed pairs")
    args = ap.parse_args()

    import torch
    import bench
    from cmoop_audio_processing_amd import EvalConfig, PopulationEvaluator, frontend, genes as G

    dev = torch.device("cuda", 0)
    wav, y = bench.synth_waveforms(args.clips, 10, 1234, dev)
    feats = frontend.log_mel(wav)
    del wav
    n_tr, n_va = int(args.clips * 0.8), int(args.clips * 0.1)
    Xtr, ytr = feats[:n_tr].contiguous(), y[:n_tr].contiguous()
    Xva, yva = feats[n_tr:n_tr + n_va].contiguous(), y[n_tr:n_tr + n_va].contiguous()
    del feats
    frontend.prepare_dataset(Xtr, Xva, None, mode="refit")
    rng = random.Random(0)
    genes = [G.normalize_hparams(G.random_hparams(rng)) for _ in range(args.pop)]
    out = {"clips": args.clips, "epochs": args.epochs, "genes": [list(g) for g in genes], "runs": {}}
    for spec in args.runs.split(","):
        compute, seed = spec.split(":")
        cfg = EvalConfig.preset("nsga_penalty", epochs=args.epochs, early_stop=False, seed=int(seed), compute=compute)
        ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
        res = ev.evaluate_genes(genes, [int(seed) + i for i in range(len(genes))])
        acc = res[:, 0]
        out["runs"][spec] = {"mean": round(float(acc.mean()), 4), "min": round(float(acc.min()), 4),
                             "below_0.9": int((acc < 0.9).sum()), "acc": [round(float(a), 4) for a in acc],
                             "seconds": round(float(res[:, 4].max()), 1)}
        print(f"[modes] {spec}: mean {acc.mean():.4f} min {acc.min():.4f} below 0.9: {(acc < 0.9).sum()}", file=sys.stderr, flush=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
