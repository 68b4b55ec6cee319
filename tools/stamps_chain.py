#!/usr/bin/env python3
"""Where one g16_chain launch spends its time: tagged in-kernel stamps (wave 0 of every 97th block) of the
-DG16_STAMPS build (tools/build_stamps.sh).
usage (GPU box): VSP_LIB_PATH=build/g16stamps/libvispeech_hip.so VSP_STAMP_CHAIN=<n> python tools/stamps_chain.py
n = 0-based index among the chain launches of ONE kernel shape in the first generator call.
Tags: 1 start | 2 x image written | 3 first slice + barrier | 10 convolution's first fragments requested |
11 retire: begin | 16 slice wait done | 17 barrier done | 12 next slice requested | 13 convolution done, reads drained |
14 next image written | 15 barrier | 20 epilogue | 21 stores issued | 22 stores retired."""
import ctypes as C
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vispeech_amd import _lib, config as vcfg           # noqa: E402
from vispeech_amd.models import SynthesizerTrn          # noqa: E402
from vispeech_amd.schema import ModelDims               # noqa: E402
from vispeech_amd.synth import synth_state_dict         # noqa: E402

dims = ModelDims()
a, kw = vcfg.synthesizer_args(vcfg.default_hparams())
net = SynthesizerTrn(*a, **kw).eval()
net.load_state_dict(synth_state_dict(dims, seed=1234, infer_only=True))
B, T = 64, 489
r = np.random.Generator(np.random.PCG64(1))
z = torch.from_numpy(r.standard_normal((B, dims.inter_channels, T)).astype(np.float32)).cuda()
g = torch.from_numpy(r.standard_normal((B, dims.gin_channels)).astype(np.float32)).cuda()
net._engine.generator(z, g)
torch.cuda.synchronize()
lib = _lib.lib()
fn = lib.vsp_debug_stamps_g16
fn.restype = C.c_int
NS, NSTAMP = 64, 256
buf = np.zeros((NS, NSTAMP), dtype=np.uint64)
n = fn(buf.ctypes.data_as(C.c_void_p), NS, 1)
print(f"g16_chain launch #{os.environ.get('VSP_STAMP_CHAIN')}: {n} sampled waves")
NAMES = {(1, 2): "x load + first image", (2, 3): "first slice wait + barrier", (3, 10): "bias + first fragment requests",
         (15, 10): "bias + first fragment requests", (10, 11): "steps", (12, 11): "steps", (10, 13): "steps", (12, 13): "steps",
         (11, 16): "slice wait", (16, 17): "barrier (slice)", (17, 12): "slice request (LDS-DMA issue)",
         (13, 11): "result -> next input (VALU)", (12, 14): "image write", (14, 15): "barrier (image)",
         (13, 20): "-", (20, 21): "epilogue loads + stores issued", (21, 22): "stores retired"}
tot = []
per = defaultdict(list)
for s in buf[:n]:
    tags = (s >> np.uint64(56)).astype(np.int64)
    t = (s & np.uint64(0x00ffffffffffffff)).astype(np.int64)
    k = int((tags > 0).sum())
    if k < 4:
        continue
    acc = defaultdict(float)
    for i in range(1, k):
        acc[NAMES.get((int(tags[i - 1]), int(tags[i])), f"{tags[i-1]}->{tags[i]}")] += (t[i] - t[i - 1]) / 100.0
    for kk, v in acc.items():
        per[kk].append(v)
    tot.append((t[k - 1] - t[0]) / 100.0)
print(f"block lifetime: median {np.median(tot):.2f} us  (min {np.min(tot):.2f}, max {np.max(tot):.2f}), {len(tot)} blocks")
for kk, v in sorted(per.items(), key=lambda kv: -np.median(kv[1])):
    print(f"  {kk:38s} {np.median(v):8.2f} us  {100 * np.median(v) / np.median(tot):5.1f} %")
