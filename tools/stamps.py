#!/usr/bin/env python3
"""Phase timeline of one cl_respair_f16s launch from the in-kernel stamps of the VSP_STAMPS build.
usage (GPU box): VSP_LIB_PATH=build/stamps/libvispeech_hip.so VSP_STAMP_LAUNCH=<n> python tools/stamps.py
n = 0-based index of the pair launch PER TILE TYPE in the first generator call: 0..8 = (k3 d1,d3,d5, k7 ..., k11 ...)
of the 64-channel stage and, in the same run, of the 32-channel stage (the two are told apart by stamp count).
Stamp order: start | loads issued | window written | barrier | per conv1 step: MFMAs done, next slice written, barrier |
t image written, barrier | per conv2 step: MFMAs done, slice written, barrier | epilogue start | end."""
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
if os.environ.get("VSP_STAMP_CL") is not None:
    # cl_conv_f16s (128-column tile): launch index within the first generator call: 0 ups0, 1..18 stage 0
    # (k3: 1-6, k7: 7-12, k11: 13-18), 19 ups1, 20..37 stage 1.  Stamps: start | staged | barrier |
    # per step: MFMAs done, next slice written, barrier | epilogue start | end.
    fn = lib.vsp_debug_stamps_cl
    fn.restype = C.c_int
    NS, NSTAMP = 128, 160
    buf2 = np.zeros((2, NS, NSTAMP), dtype=np.uint64)
    n = fn(buf2.ctypes.data_as(C.c_void_p), 2 * NS, 1)
    buf, cyc = buf2[0], buf2[1]
    print(f"cl_conv_f16s launch #{os.environ['VSP_STAMP_CL']}: {n} sampled blocks (wave {os.environ.get('VSP_STAMP_WAVE', '0')})")
    s = buf[:n].astype(np.int64)
    c = cyc[:n].astype(np.int64)
    nz = (s > 0).sum(axis=1)
    if n:
        k = int(nz.min())
        dt_us = (s[:, k - 1] - s[:, 0]) / 100.0
        dcyc = (c[:, k - 1] - c[:, 0])
        print(f"   shader clock over the block lifetime: median {np.median(dcyc / dt_us):.0f} MHz "
              f"(s_memtime cycles / s_memrealtime us)")
        # clock inside MFMA phases only (stamp 3 -> 4 is the first step's MFMA phase, then every third)
        idx = np.arange(2, k - 3, 3)
        mf = (c[:, idx + 1] - c[:, idx]) / np.maximum((s[:, idx + 1] - s[:, idx]) / 100.0, 1e-9)
        print(f"   shader clock inside the MFMA phases: median {np.median(mf):.0f} MHz")
    counts = (s > 0).sum(axis=1)
    for cnt in sorted(set(counts.tolist())):
        g = s[counts == cnt][:, :cnt]
        rel = (g - g[:, :1]) / 100.0
        med = np.median(rel, axis=0)
        d = np.diff(med)
        print(f"-- {len(g)} blocks with {cnt} stamps: block lifetime median {med[-1]:.2f} us")
        body = d[2:-2]
        k = len(body) // 3 * 3
        b3 = body[:k].reshape(-1, 3)
        print(f"   prologue {d[0]:.2f} + barrier {d[1]:.2f}; per step (median over {len(b3)} steps): MFMA phase "
              f"{np.median(b3[:, 0]):.2f}, slice write {np.median(b3[:, 1]):.2f}, barrier {np.median(b3[:, 2]):.2f}; "
              f"sum of MFMA phases {b3[:, 0].sum():.1f}, writes {b3[:, 1].sum():.1f}, barriers {b3[:, 2].sum():.1f}; "
              f"tail {' '.join(f'{x:.2f}' for x in d[-3:])}")
        print("   deltas us:", " ".join(f"{x:.2f}" for x in d[:40]), "...")
    sys.exit(0)
fn = lib.vsp_debug_stamps
fn.restype = C.c_int
NS, NSTAMP = 256, 64
buf = np.zeros((NS, NSTAMP), dtype=np.uint64)
n = fn(buf.ctypes.data_as(C.c_void_p), NS, 1)
print(f"pair launch #{os.environ.get('VSP_STAMP_LAUNCH')} of each tile type: {n} sampled blocks (wave 0); stamps are 100 MHz wall clock")
s = buf[:n].astype(np.int64)
counts = (s > 0).sum(axis=1)
for cnt in sorted(set(counts.tolist())):
    g = s[counts == cnt][:, :cnt]
    rel = (g - g[:, :1]) / 100.0                  # us since block start
    med = np.median(rel, axis=0)
    d = np.diff(med)
    print(f"-- {len(g)} blocks with {cnt} stamps: block lifetime median {med[-1]:.2f} us "
          f"(p10 {np.percentile(rel[:, -1], 10):.2f}, p90 {np.percentile(rel[:, -1], 90):.2f})")
    print("   deltas us:", " ".join(f"{x:.2f}" for x in d))
