#!/usr/bin/env python3
"""Per-wave phase timeline of the persistent pair kernel g16_rw (gen16_rw.hip) from the -DRW_STAMPS build: lane 0 of every
wave of the middle block stamps iterations 20 .. 23.
usage (GPU box): VSP_LIB_PATH=build/rwstamps/libvispeech_hip.so python tools/stamps_rw.py <K> <dil>
Tags: 1 iteration start | 2 first reads + window requests out | 3-5 group MFMAs issued | 6-8 group vector work done
(conv1: t image written; conv2: residual added, stores issued) | 9 window pieces landed | 10 window split | 11 at the barrier."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vispeech_amd import _lib  # noqa: E402

k, dil = int(sys.argv[1]), int(sys.argv[2])
c, b, t = 32, 64, 250368
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
fn = C.CDLL(_lib.LIB_PATH).vsp_debug_stamps_rw
buf = np.zeros((8, 64), dtype=np.uint64)
assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
tags = (buf >> np.uint64(56)).astype(np.int64)
tm = (buf & np.uint64(0x00ffffffffffffff)).astype(np.int64)
t0 = tm[tags > 0].min()
print(f"g16_rw<{k}> dilation {dil}: stamps in us relative to the first one; waves 0-3 conv1, 4-7 conv2")
for w in range(8):
    n = int((tags[w] > 0).sum())
    line, prev = [], None
    for i in range(n):
        us = (tm[w, i] - t0) / 100.0
        if tags[w, i] == 1:
            line.append("\n      |")
        line.append(f" {tags[w, i]}:{us:6.2f}")
    print(f"wave {w}:" + "".join(line))
