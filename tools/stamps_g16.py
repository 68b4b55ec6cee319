#!/usr/bin/env python3
"""Phase timeline of one g16_conv launch (128-row tile) from the in-kernel stamps of the -DG16_STAMPS build.
usage (GPU box): VSP_LIB_PATH=build/g16stamps/libvispeech_hip.so VSP_STAMP_G16=<n> python tools/stamps_g16.py
n = 0-based index of the big-tile launch in the first generator call: 0 ups0, 1..18 stage 0 (k3: 1-6, k7: 7-12,
k11: 13-18), 19 ups1, 20..37 stage 1, 38 ups2.
Stamps: 0 start | 1 window written | 2 slices landed | 3 barrier | per step: MEM done, barrier, MFMAs issued, barrier |
epilogue start | stores issued | stores retired."""
import ctypes as C
import os
import sys

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
print(f"g16_conv launch #{os.environ.get('VSP_STAMP_G16')}: {n} sampled waves")
s = buf[:n].astype(np.int64)
for half, name in ((1, "first half (wave 0)"), (5, "second half (wave 4)")):
    g_ = s[s[:, NSTAMP - 1] == half][:, :NSTAMP - 1]
    if not len(g_):
        continue
    cnt = int((g_ > 0).sum(axis=1).min())
    rel = (g_[:, :cnt] - g_[:, :1]) / 100.0
    med = np.median(rel, axis=0)
    d = np.diff(med)
    nsteps = (cnt - 4 - 3) // 4
    print(f"-- {name}: {len(g_)} waves, {cnt} stamps, {nsteps} steps, lifetime median {med[-1]:.2f} us")
    print(f"   prologue: window written {med[1]:.2f}, slices landed +{d[1]:.2f}, barrier +{d[2]:.2f}")
    st = d[3:3 + 4 * nsteps].reshape(nsteps, 4)
    print("   per step median: MEM %.2f  barrier %.2f  MFMA %.2f  barrier %.2f   (sum %.2f us)" % (*np.median(st, axis=0), np.median(st.sum(axis=1))))
    print("   per step mean:   MEM %.2f  barrier %.2f  MFMA %.2f  barrier %.2f   (sum %.2f us; all steps %.2f us)" % (*st.mean(axis=0), st.sum(axis=1).mean(), st.sum()))
    print("   tail deltas:", " ".join(f"{x:.2f}" for x in d[3 + 4 * nsteps:]))
    print("   steps (MEM/bar/MFMA/bar):", " | ".join(" ".join(f"{x:.2f}" for x in row) for row in st[:14]))
