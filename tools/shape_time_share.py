#!/usr/bin/env python3
"""Where the headline generation's GEMM time goes, by layer shape: the seeded pop-40 population of bench.py x the isolated
per-shape kernel table (tools/kernel_bench.py output, e.g. profiles/r02_kernel_bench_isolated.txt).  CPU only.

    python tools/shape_time_share.py profiles/r02_kernel_bench_isolated.txt [--n-train 24000 --n-val 3000 --epochs 2]
"""
import argparse, ast, collections, os, random, re, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmoop_audio_processing_amd import genes as G


def read_table(path):
    tab = {}
    for line in open(path):
        m = re.match(r"\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)\s+([\d.]+) \|\s+([\d.]+)\s+([\d.]+) \|\s+([\d.]+)\s+([\d.]+) \|\s+([\d.]+)\s+([\d.]+)", line)
        if m:
            v = [float(x) for x in m.groups()]
            tab[tuple(int(x) for x in v[1:6])] = {"gflop": v[6], "fwd": v[8], "dgrad": v[10], "wgrad": v[12]}
    return tab


def conv_shapes(g, T=101, F=40):
    """(H, W, Cin, Cout, KS, stride, needs_dgrad) of every MFMA conv of a variant-A candidate, forward order."""
    f, k, bn, R, fc, _ = g
    out = [(T, F, f, f, k, 1, True)]           # conv2; conv1 (C_in = 1) is a VALU kernel
    h, w, c = (T + 1) // 2, (F + 1) // 2, f    # max-pool
    for r in range(R):                          # nsga_penalty.py:276-301: skip 1x1/2, conv c->2c, conv 2c->2c, pool
        out.append((h, w, c, 2 * c, 1, 2, True))
        out.append((h, w, c, 2 * c, k, 1, True))
        out.append((h, w, 2 * c, 2 * c, k, 1, True))
        h, w, c = (h + 1) // 2, (w + 1) // 2, 2 * c
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("table")
    ap.add_argument("--pop", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--n-train", type=int, default=24000)
    ap.add_argument("--n-val", type=int, default=3000)
    ap.add_argument("--epochs", type=int, default=2)
    a = ap.parse_args()
    tab = read_table(a.table)
    rng = random.Random(a.seed)
    genes = [G.normalize_hparams(G.random_hparams(rng)) for _ in range(a.pop)]
    train_batches = a.epochs * a.n_train / 64.0
    fwd_only_batches = a.epochs * a.n_val / 64.0
    share = collections.defaultdict(lambda: [0.0, 0.0])   # (shape, pass) -> [seconds, gflop]
    missing = collections.Counter()
    for g in genes:
        for (h, w, ci, co, k, st, dg) in conv_shapes(g):
            key = (h, w, ci, co, k)
            if st != 1 or key not in tab:
                missing[(h, w, ci, co, k, st)] += 1
                continue
            row = tab[key]
            for p, n in (("fwd", train_batches + fwd_only_batches), ("dgrad", train_batches), ("wgrad", train_batches)):
                share[(key, p)][0] += n * row["gflop"] / row[p] / 1e3
                share[(key, p)][1] += n * row["gflop"]
    tot_s = sum(v[0] for v in share.values())
    tot_g = sum(v[1] for v in share.values())
    print(f"sum of isolated GEMM time {tot_s:.1f} s, {tot_g / 1e3:.0f} TFLOP, weighted {tot_g / tot_s / 1e3:.1f} TFLOP/s; shapes not in the table: {dict(missing)}")
    by_shape = collections.defaultdict(lambda: [0.0, 0.0])
    for (key, p), v in share.items():
        by_shape[key][0] += v[0]; by_shape[key][1] += v[1]
    for key, v in sorted(by_shape.items(), key=lambda kv: -kv[1][0]):
        parts = " ".join(f"{p} {share[(key, p)][1] / share[(key, p)][0] / 1e3:6.1f}" for p in ("fwd", "dgrad", "wgrad"))
        print(f"{100 * v[0] / tot_s:5.1f} %  {v[0]:6.2f} s  {str(key):28s} avg {v[1] / v[0] / 1e3:6.1f} TFLOP/s   {parts}")


if __name__ == "__main__":
    main()
