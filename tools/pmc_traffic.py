#!/usr/bin/env python3
"""Three launches each of the forward / dgrad / wgrad MFMA kernel on one layer shape -- the target of a
`rocprofv3 --pmc FETCH_SIZE WRITE_SIZE` pass (tools/pmc_traffic.sh).  Few launches on purpose: a counter pass
serialises every dispatch."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmoop_audio_processing_amd import _lib  # noqa: E402

B, H, W, Cin, Cout, KS = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,51,20,128,128,5").split(",")]
L = _lib.lib()
x = torch.randn((B, H, W, Cin), device="cuda")
w = torch.randn((Cout, KS, KS, Cin), device="cuda") * 0.05
b = torch.randn((Cout,), device="cuda")
y = torch.randn((B, H, W, Cout), device="cuda")
torch.cuda.synchronize()
for mode in (0, 1, 2):
    ms = C.c_double()
    _lib.check(L.cmoop_conv_time(mode, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), B, H, W, Cin, Cout, KS, 3, C.byref(ms)))
    print("mode", mode, "avg ms", ms.value, flush=True)
