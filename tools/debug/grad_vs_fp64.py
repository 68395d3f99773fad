#!/usr/bin/env python3
"""Per-tensor gradient error of the HIP path and of the fp32 oracle, both against the float64 oracle (one train step).
usage: grad_vs_fp64.py f,k,bn,R,fc,dr variant T F B [B2 ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmoop_audio_processing_amd import EvalConfig, genes as G
from cmoop_audio_processing_amd.session import NetSession
from oracle import net as ON

gene = tuple(int(v) for v in sys.argv[1].split(","))
variant, T, F = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
v = G.VARIANT_NAMES[variant]
rs = np.random.RandomState(77)
y = rs.randint(0, 10, size=64).astype(np.int32)
proto = rs.randn(10, T, F).astype(np.float32)
X = (0.8 * proto[y] + rs.randn(64, T, F)).astype(np.float32)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
for B in [int(b) for b in sys.argv[5:]]:
    cfg = EvalConfig(variant=variant, classes=10, batch=64, eval_batch=64)
    ocfg = ON.OracleConfig(variant=v, classes=10, batch=64)
    o32, o64 = ON.OracleNet(gene, ocfg, 9), ON.OracleNet(gene, ocfg, 9, dtype=torch.float64)
    o32.train_step(X[:B], y[:B]); o64.train_step(X[:B], y[:B])
    with NetSession(gene, cfg, T, F, 9) as net:
        net.train_step(Xd, yd, None, row0=0, B=B)
        g = net.get_grads().astype(np.float64)
    g32, g64 = o32.grads_flat().astype(np.float64), o64.grads_flat()
    gmax = np.abs(g64).max()
    off = 0
    print(f"B={B} gene={gene} {variant} {T}x{F}")
    for name, shape, role in G.param_tensors(gene, v, 10):
        n = int(np.prod(shape))
        ref = g64[off:off + n]
        den = max(np.abs(ref).max(), 1e-4 * gmax, 1e-30)
        e_hip = np.abs(g[off:off + n] - ref).max() / den
        e_o32 = np.abs(g32[off:off + n] - ref).max() / den
        e_x = np.abs(g[off:off + n] - g32[off:off + n]).max() / den
        if role in ("kernel", "bias", "gamma", "beta"):
            print(f"  {name:24s} |ref|max {np.abs(ref).max():.3e}  hip-vs-fp64 {e_hip:.2e}  oracle32-vs-fp64 {e_o32:.2e}  hip-vs-oracle32 {e_x:.2e}")
        off += n
