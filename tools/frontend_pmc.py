#!/usr/bin/env python3
"""Three launches of the front end's logmel_kernel on the bench's 30 000 synthetic clips -- the target of the front-end PMC
passes of tools/pmc_round3.sh (a counter pass serialises every dispatch, hence few launches); prints the plain timing too."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmoop_audio_processing_amd import frontend  # noqa: E402

wav = torch.randn((30000, 16000), device="cuda")
frontend.log_mel(wav[:256])
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    out = frontend.log_mel(wav)
    best = min(best, time.perf_counter() - t)
print(f"logmel 30000 clips best {best * 1e3:.3f} ms = {(wav.numel() * 4 + out.numel() * 4) / best / 1e9:.0f} GB/s algorithmic")
