#!/usr/bin/env python3
"""Step-by-step re-synchronised comparison of one candidate's first epoch (GPU vs oracle, state copied GPU -> oracle before
EVERY step): per-step loss and per-tensor gradient deviation.  Debug aid for an epoch whose end-of-epoch validation loss
deviates more than the oracle's own twins do.
  python tools/debug/resync_steps.py 64,3,0,2,2,1 --seed 6"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cmoop_audio_processing_amd import EvalConfig, genes as G  # noqa: E402
from cmoop_audio_processing_amd.session import NetSession, epoch_permutation  # noqa: E402
from oracle import net as ON  # noqa: E402
from test_gpu_net import make_split, ocfg, per_tensor_err  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("gene")
ap.add_argument("--seed", type=int, default=6)
ap.add_argument("--classes", type=int, default=35)
ap.add_argument("--preset", default="sa_nsga_penalty")
ap.add_argument("--epochs", type=int, default=1)
a = ap.parse_args()
gene = tuple(int(v) for v in a.gene.split(","))
cfg = EvalConfig.preset(a.preset, classes=a.classes, epochs=6, patience=2, batch=32, eval_batch=64, seed=5, n_slots=1)
Xtr, ytr, Xva, yva = make_split(420, 140, 21, 12, a.classes, 61, noise=0.3, label_noise=0.1)
Xd, yd = torch.from_numpy(Xtr).cuda(), torch.from_numpy(ytr).cuda()
Xvd, yvd = torch.from_numpy(Xva).cuda(), torch.from_numpy(yva).cuda()
v = G.VARIANT_NAMES[cfg.variant]
torch.set_num_threads(8)
with NetSession(gene, cfg, 21, 12, a.seed) as net:
    o32 = ON.OracleNet(gene, ocfg(cfg), a.seed)
    o64 = ON.OracleNet(gene, ocfg(cfg), a.seed, dtype=torch.float64)
    for epoch in range(a.epochs):
        perm = epoch_permutation(a.seed, epoch, len(Xtr))
        idx = torch.from_numpy(perm).cuda()
        for s in range(0, len(Xtr), cfg.batch):
            st = net.get_state()
            o32.set_state(st)
            o64.set_state(st)
            b = min(cfg.batch, len(Xtr) - s)
            net.train_step(Xd, yd, idx, row0=s, B=b)
            rows = perm[s:s + b]
            l32, _ = o32.train_step(Xtr[rows], ytr[rows])
            l64, _ = o64.train_step(Xtr[rows], ytr[rows])
            lg, _ = net.train_metrics()
            e_hip = per_tensor_err(gene, v, a.classes, net.get_grads(), o64.grads_flat())
            e_o32 = per_tensor_err(gene, v, a.classes, o32.grads_flat(), o64.grads_flat())
            wh, wo = max(e_hip, key=e_hip.get), max(e_o32, key=e_o32.get)
            dp_h = np.abs(net.get_params().astype(np.float64) - o64.get_flat()).max()
            dp_o = np.abs(o32.get_flat().astype(np.float64) - o64.get_flat()).max()
            print(f"epoch {epoch} step {s // cfg.batch:2d} B={b}: loss gpu {lg / b:.7f} o32 {l32 / b:.7f} o64 {l64 / b:.7f} | grad err vs f64: gpu worst {wh} {e_hip[wh]:.1e}, "
                  f"o32 worst {wo} {e_o32[wo]:.1e} | max |param - f64| after the step: gpu {dp_h:.1e} o32 {dp_o:.1e}", flush=True)
        st = net.get_state()
        lg, ag, _ = net.evaluate(Xvd, yvd)
        print(f"epoch {epoch}: gpu val loss {lg:.7f}")
