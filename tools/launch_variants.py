"""Host-only: the launch-path variants (cmoop_conv_launch_plan) the trainer uses for every conv layer of a set of genes.
Usage: python tools/launch_variants.py [--all] [--T 101 --F 40]   (default: the bench's 40 genes of random.Random(0))"""
import argparse
import ctypes as C
import itertools
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmoop_audio_processing_amd import _lib, genes as G  # noqa: E402


def plan(op, B, H, W, Ci, Co, KS, st, stats=0):
    buf = C.create_string_buffer(200)
    _lib.check(_lib.lib().cmoop_conv_launch_plan(op, B, H, W, Ci, Co, KS, st, stats, buf, 200))
    return buf.value.decode()


def conv_layers(gene, variant, T, F):
    """(H, W, Cin, Cout, KS, stride, feeds_bn) of every implicit-GEMM conv of a candidate (Net::build_plan's walk)."""
    f, k, bn, R, fc, dr = gene
    out = []
    if variant == 0:
        out.append((T, F, f, f, k, 1, bn))
    h, w, c = (T + 1) // 2, (F + 1) // 2, f
    for _ in range(R):
        out.append((h, w, c, 2 * c, 1, 2, 0))
        out.append((h, w, c, 2 * c, k, 1, bn))
        if variant == 0:
            out.append((h, w, 2 * c, 2 * c, k, 1, bn))
        h, w, c = (h + 1) // 2, (w + 1) // 2, 2 * c
    return out


def variants_of(genes, variant, T, F, batch=64, eval_batch=256, partial=(37,)):
    used = {}
    for g in genes:
        for (H, W, Ci, Co, KS, st, bn) in conv_layers(g, variant, T, F):
            for B in (batch,) + tuple(partial):
                used.setdefault(plan(0, B, H, W, Ci, Co, KS, st, bn), set()).add(("fwd", B, H, W, Ci, Co, KS, st))
                used.setdefault(plan(1, B, H, W, Ci, Co, KS, st), set()).add(("dgrad", B, H, W, Ci, Co, KS, st))
                used.setdefault(plan(2, B, H, W, Ci, Co, KS, st), set()).add(("wgrad", B, H, W, Ci, Co, KS, st))
            used.setdefault(plan(0, eval_batch, H, W, Ci, Co, KS, st, 0), set()).add(("fwd", eval_batch, H, W, Ci, Co, KS, st))
    return used


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--T", type=int, default=101)
    ap.add_argument("--F", type=int, default=40)
    ap.add_argument("--variant", default="A")
    a = ap.parse_args()
    if a.all:
        genes = list(itertools.product((16, 32, 64), (3, 5), (0, 1), (1, 2, 3), (1,), (0,)))
    else:
        rng = random.Random(0)
        genes = [G.normalize_hparams(G.random_hparams(rng)) for _ in range(40)]
    used = variants_of(genes, G.VARIANT_NAMES[a.variant], a.T, a.F)
    for k in sorted(used):
        print(f"{k:60s} {len(used[k]):3d}  e.g. {sorted(used[k])[0]}")
