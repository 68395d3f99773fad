#!/usr/bin/env python3
"""Isolated timing of the MFMA implicit-GEMM kernels on the layer shapes of topology A/B
(batch 64, 101x40 features).  Prints TFLOP/s per shape and mode (fwd / dgrad / wgrad)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmoop_audio_processing_amd import _lib  # noqa: E402

SHAPES = []
for f in (16, 32, 64):
    for k in (3, 5):
        SHAPES.append((64, 101, 40, f, f, k))          # conv2 of topology A
        h, w, c = 51, 20, f
        for r in range(3):
            SHAPES.append((64, h, w, c, 2 * c, k))     # res conv1
            SHAPES.append((64, h, w, 2 * c, 2 * c, k))  # res conv2 (topology A)
            h, w, c = (h + 1) // 2, (w + 1) // 2, 2 * c
SHAPES = sorted(set(SHAPES), key=lambda s: (s[5], s[3], s[4], -s[1]))


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    L = _lib.lib()
    print(f"{'B,H,W,Cin,Cout,KS':28s} {'GFLOP':>8s} | " + " | ".join(f"{m:>7s} ms   TF/s" for m in ("fwd", "dgrad", "wgrad")))
    for (B, H, W, Cin, Cout, KS) in SHAPES:
        if only and not any(o in f"{Cin},{Cout},{KS}" for o in only.split("|")):   # "64,64,|128,128,5": substrings of "Cin,Cout,KS"
            continue
        x = torch.randn((B, H, W, Cin), device="cuda")
        w = torch.randn((Cout, KS, KS, Cin), device="cuda") * 0.05
        b = torch.randn((Cout,), device="cuda")
        y = torch.randn((B, H, W, Cout), device="cuda")
        torch.cuda.synchronize()
        fl = 2.0 * B * H * W * Cout * KS * KS * Cin
        out = []
        for mode in (0, 1, 2):
            best = 1e30
            for _ in range(3):          # min of 3 x 30 back-to-back launches (DVFS / neighbour noise)
                ms = C.c_double()
                _lib.check(L.cmoop_conv_time(mode, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), B, H, W, Cin, Cout, KS, 30, C.byref(ms)))
                best = min(best, ms.value)
            out.append(f"{best:8.3f} {fl / best / 1e9:6.1f}")
        print(f"{str((B, H, W, Cin, Cout, KS)):28s} {fl / 1e9:8.2f} | " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
