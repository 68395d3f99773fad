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
ap.add_argument("--data", default="split35", help="split35: the configs[2] test's data; hard: the hard-synthetic-set test's data (HIP front end)")
a = ap.parse_args()
gene = tuple(int(v) for v in a.gene.split(","))
if a.data == "hard":
    import bench
    from cmoop_audio_processing_amd import frontend
    a.classes, a.preset = 10, "nsga_penalty"
    wav, y = bench.synth_waveforms(480, 10, 7, torch.device("cuda"), n_samples=4000, chunk=160, hard=True, hard_snr_db=-6.0)
    feats = frontend.log_mel(wav)
    Xtr_d, Xva_d = feats[:320].contiguous(), feats[320:].contiguous()
    frontend.prepare_dataset(Xtr_d, Xva_d, None, mode="refit")
    Xtr, Xva = Xtr_d.cpu().numpy(), Xva_d.cpu().numpy()
    ytr, yva = y[:320].cpu().numpy(), y[320:].cpu().numpy()
    cfg = EvalConfig.preset("nsga_penalty", epochs=12, patience=2, batch=32, eval_batch=64, seed=3, n_slots=1, fpr_variant="v1")
else:
    cfg = EvalConfig.preset(a.preset, classes=a.classes, epochs=6, patience=2, batch=32, eval_batch=64, seed=5, n_slots=1)
    Xtr, ytr, Xva, yva = make_split(420, 140, 21, 12, a.classes, 61, noise=0.3, label_noise=0.1)
T_, F_ = int(Xtr.shape[1]), int(Xtr.shape[2])
Xd, yd = torch.from_numpy(Xtr).cuda(), torch.from_numpy(ytr).cuda()
Xvd, yvd = torch.from_numpy(Xva).cuda(), torch.from_numpy(yva).cuda()
v = G.VARIANT_NAMES[cfg.variant]
torch.set_num_threads(8)
tensors = G.param_tensors(gene, G.VARIANT_NAMES[cfg.variant], a.classes)
with NetSession(gene, cfg, T_, F_, a.seed) as net:
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
            ph, p32, p64 = net.get_params().astype(np.float64), o32.get_flat().astype(np.float64), o64.get_flat()
            dp_h = np.abs(ph - p64).max()
            dp_o = np.abs(p32 - p64).max()
            off, sv = 0, []
            for name, shape, role in tensors:            # moving_var after the step: |x - f64| / max|f64|, worst tensor, gpu / o32
                n = int(np.prod(shape))
                if role == "moving_var":
                    mx = max(np.abs(p64[off:off + n]).max(), 1e-3)
                    sv.append((np.abs(ph[off:off + n] - p64[off:off + n]).max() / mx, np.abs(p32[off:off + n] - p64[off:off + n]).max() / mx, name))
                off += n
            if sv:
                w = max(sv)
                print(f"      moving_var worst: gpu {w[0]:.1e} o32 {w[1]:.1e} ({w[2]})")
            print(f"epoch {epoch} step {s // cfg.batch:2d} B={b}: loss gpu {lg / b:.7f} o32 {l32 / b:.7f} o64 {l64 / b:.7f} | grad err vs f64: gpu worst {wh} {e_hip[wh]:.1e}, "
                  f"o32 worst {wo} {e_o32[wo]:.1e} | max |param - f64| after the step: gpu {dp_h:.1e} o32 {dp_o:.1e}", flush=True)
        st = net.get_state()
        lg, ag, _ = net.evaluate(Xvd, yvd)
        print(f"epoch {epoch}: gpu val loss {lg:.7f}")
