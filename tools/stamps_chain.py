#!/usr/bin/env python3
"""Where one g16_chain launch (whole-ResBlock kernel) spends its time: tagged in-kernel stamps (wave 0 of every 197th block) of the
-DG16_STAMPS build (tools/build_stamps.sh).
usage (GPU box): VSP_LIB_PATH=build/g16stamps/libvispeech_hip.so VSP_STAMP_CHAIN=<n> python tools/stamps_chain.py
n = 0-based index among the pair launches of ONE kernel shape (C = 64: k3 0-2, k7 3-5, k11 6-8; C = 32: k7 0-2, k11 3-5).
Tags: 1 start | 2 requests out | 3 x window written | per slice: 10 top, 11 slice landed, 12 barrier, 13 DMA issued, 14 MFMAs issued |
15/16 chunk hand-over barrier / window written | 20 conv1 done | 21-23 image chunk: barrier, written | 30 epilogue | 31 stores issued | 32 retired."""
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
NAMES = {(1, 2): "x tile from HBM, first image written", (2, 3): "first slice wait + barrier", (3, 10): "first fragments requested",
         (10, 11): "MFMA steps", (12, 11): "MFMA steps", (11, 16): "slice wait (vmcnt)", (16, 17): "barrier (slice)", (17, 12): "LDS-DMA issue",
         (12, 13): "MFMA steps", (10, 13): "MFMA steps", (13, 11): "fold + residual (registers)", (12, 14): "image: split + write",
         (13, 14): "image: split + write", (14, 15): "barrier (image)", (15, 10): "first fragments requested", (13, 20): "-",
         (20, 21): "epilogue: stores issued", (21, 22): "stores retired"}
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
