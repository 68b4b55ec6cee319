#!/usr/bin/env python3
"""One fused-pair launch sequence at the C3 size of the 32-channel stage (64 x 250368 rows), for rocprofv3 --kernel-trace:
usage: pair_time.py <C> <K> <dil,dil,..> [B] [T]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vispeech_amd import _lib

c, k = int(sys.argv[1]), int(sys.argv[2])
dils = tuple(int(v) for v in sys.argv[3].split(","))
b = int(sys.argv[4]) if len(sys.argv) > 4 else 64
t = int(sys.argv[5]) if len(sys.argv) > 5 else 250368 * 32 // c
lib = _lib.lib()
r = np.random.Generator(np.random.PCG64(1))
x = torch.randn(b, t, c, device="cuda")
out = torch.empty_like(x)
ws = [(r.standard_normal((c, c, k)) / np.sqrt(c * k)).astype(np.float32) for _ in range(2 * len(dils))]
bs = [r.standard_normal(c).astype(np.float32) * 0.1 for _ in range(2 * len(dils))]
hp = lambda arrs: (C.c_void_p * len(arrs))(*[a.ctypes.data_as(C.c_void_p) for a in arrs])
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
darr = (C.c_int * len(dils))(*dils)
for _ in range(3):
    rc = lib.vsp_cl_resblock(stream, b, t, c, k, len(dils), darr, C.c_void_p(x.data_ptr()), hp(ws), hp(bs), 1, 3, C.c_void_p(out.data_ptr()))
    assert rc == 0, rc
torch.cuda.synchronize()
