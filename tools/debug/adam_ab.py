#!/usr/bin/env python3
"""Bitwise A/B of the optimiser launch: dumps params and grads after 1 and 4 train steps of a BN + dropout candidate;
run once plain and once with CMOOP_ADAM_UNFUSED=1, then `--compare a.npz b.npz`."""
import sys, numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = a[k].view(np.uint32) != b[k].view(np.uint32)
        print(k, "differing words", int(d.sum()), "of", d.size, "max |d|", float(np.abs(a[k] - b[k]).max()),
              "first", np.flatnonzero(d)[:8].tolist())
    sys.exit(0)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cmoop_audio_processing_amd import EvalConfig
from cmoop_audio_processing_amd.session import NetSession
gene = (16, 3, 1, 1, 2, 1)
cfg = EvalConfig.preset("sa_nsga_penalty", classes=11, epochs=25, patience=2, batch=32, eval_batch=64, seed=11, n_slots=1)
rs = np.random.RandomState(0)
X = torch.from_numpy(rs.randn(128, 21, 12).astype(np.float32)).cuda()
y = torch.from_numpy(rs.randint(0, 11, 128).astype(np.int32)).cuda()
out = {}
with NetSession(gene, cfg, 21, 12, 11) as net:
    for step in range(4):
        net.train_step(X, y, None, row0=32 * step, B=32)
        if step in (0, 3):
            out[f"params_{step + 1}"] = np.array(net.get_params())
            out[f"grads_{step + 1}"] = np.array(net.get_grads())
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], {k: v.shape for k, v in out.items()})
