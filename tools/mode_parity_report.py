import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from cmoop_audio_processing_amd import EvalConfig, genes as G
from cmoop_audio_processing_amd.session import NetSession
from oracle import net as ON
import test_gpu_net as TG
classes, T, F, B, seed = 10, 21, 12, 24, 99
X, y = TG.make_data(64, T, F, classes, 3)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
for gene in [(32, 5, 0, 2, 3, 1), (32, 5, 1, 2, 3, 1), (16, 3, 1, 1, 2, 1), (16, 3, 0, 1, 1, 0)]:
    for compute in ("fp32", "bf16x3", "bf16"):
        cfg = EvalConfig(variant="A", classes=classes, batch=32, eval_batch=16, compute=compute)
        onet = ON.OracleNet(gene, TG.ocfg(cfg), seed)
        o32 = ON.OracleNet(gene, TG.ocfg(EvalConfig(variant="A", classes=classes, batch=32)), seed)
        with NetSession(gene, cfg, T, F, seed) as net:
            net.train_step(Xd, yd, None, row0=8, B=B)
            lo, co = onet.train_step(X[8:8 + B], y[8:8 + B]); o32.train_step(X[8:8 + B], y[8:8 + B])
            lg, cg = net.train_metrics()
            g, go, g32 = net.get_grads().astype(np.float64), onet.grads_flat().astype(np.float64), o32.grads_flat().astype(np.float64)
            e1 = TG.per_tensor_err(gene, 0, classes, g, go); e2 = TG.per_tensor_err(gene, 0, classes, g, g32)
            cos = lambda a, b: float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
            print(f"{gene} {compute:7s} loss gpu {lg:.6f} oracle {lo:.6f} | vs own oracle: max {max(e1.values()):.2e} cos {cos(g, go):.8f} "
                  f"| vs fp32 oracle: max {max(e2.values()):.2e} cos {cos(g, g32):.6f}", flush=True)
