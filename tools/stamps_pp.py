#!/usr/bin/env python3
"""Per-wave phase timeline of the 128-channel pair kernel g16_pp (gen16_pp.hip) from the -DPP_STAMPS build: lane 0 of every
wave of one block in the middle of the grid.
usage (GPU box): VSP_LIB_PATH=build/ppstamps/libvispeech_hip.so python tools/stamps_pp.py <K> <dil>
Tags: 1 start | 2 requests out | 3 first chunk written | 4 slices landed | 5 loop start | per step: 11 MEM work issued +
slice wait done, 12 barrier, 13 MFMAs issued, 14 barrier | 20 conv1 done | 21 t tiles written | 22 hand-over barriers |
30 conv2 done | 31 residual requested | 32 residual here | 33 stores issued | 34 stores retired."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vispeech_amd import _lib  # noqa: E402

k, dil = int(sys.argv[1]), int(sys.argv[2])
c, b, t = 128, 64, 31296
lib = _lib.lib()
r = np.random.Generator(np.random.PCG64(1))
x = torch.randn(b, t, c, device="cuda")
out = torch.empty_like(x)
ws = [(r.standard_normal((c, c, k)) / np.sqrt(c * k)).astype(np.float32) for _ in range(2)]
bs = [r.standard_normal(c).astype(np.float32) * 0.1 for _ in range(2)]
hp = lambda arrs: (C.c_void_p * len(arrs))(*[a.ctypes.data_as(C.c_void_p) for a in arrs])
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
darr = (C.c_int * 1)(dil)
for _ in range(2):
    assert lib.vsp_cl_resblock(stream, b, t, c, k, 1, darr, C.c_void_p(x.data_ptr()), hp(ws), hp(bs), 1, 3, C.c_void_p(out.data_ptr())) == 0
torch.cuda.synchronize()
fn = C.CDLL(_lib.LIB_PATH).vsp_debug_stamps_pp
NS = 192
buf = np.zeros((8, NS), dtype=np.uint64)
assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
tags = (buf >> np.uint64(56)).astype(np.int64)
tm = (buf & np.uint64(0x00ffffffffffffff)).astype(np.int64)
t0 = tm[tags > 0].min()
print(f"g16_pp K={k} dilation {dil}: us relative to the block's first stamp; waves 0-3 first half, 4-7 second half")
for w in (0, 4):
    n = int((tags[w] > 0).sum())
    print(f"wave {w}:")
    i = 0
    step = 0
    while i < n:
        tg, us = tags[w, i], (tm[w, i] - t0) / 100.0
        if tg == 11 and i + 3 < n:
            a = [(tm[w, i + j] - t0) / 100.0 for j in range(4)]
            prev = (tm[w, i - 1] - t0) / 100.0
            print(f"   step {step:2d}: MEM {a[0] - prev:5.2f} | barrier {a[1] - a[0]:5.2f} | MFMA {a[2] - a[1]:5.2f} | barrier {a[3] - a[2]:5.2f}   (ends {a[3]:6.2f})")
            step += 1
            i += 4
            continue
        print(f"   tag {tg:2d} at {us:6.2f}")
        i += 1
