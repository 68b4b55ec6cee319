#!/usr/bin/env python3
"""Is the generator power-bound?  Times Engine.generator (C3 shape: 64 x 489 frames) with the real synthetic weights
and with the vocoder's convolution weights set to ZERO (same kernels, same launches, same bytes moved; the matrix
cores multiply zeros): if the time drops, the difference is what the data-dependent power of the matrix pipe costs.
usage (GPU box): python tools/gen_zero_weights.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vispeech_amd import config as vcfg
from vispeech_amd.models import SynthesizerTrn
from vispeech_amd.schema import ModelDims
from vispeech_amd.synth import synth_state_dict

dims = ModelDims()
a, kw = vcfg.synthesizer_args(vcfg.default_hparams())
B, T = 64, 489
r = np.random.Generator(np.random.PCG64(1))
z = torch.from_numpy(r.standard_normal((B, dims.inter_channels, T)).astype(np.float32)).cuda()
g = torch.from_numpy(r.standard_normal((B, dims.gin_channels)).astype(np.float32)).cuda()
for mode in ("real", "zero_weights", "zero_input", "real"):
    sd = synth_state_dict(dims, seed=1234, infer_only=True)
    if mode == "zero_weights":
        for k in sd:
            if k.startswith("dec.") and ("weight" in k) and not k.endswith("weight_g"):
                sd[k] = np.zeros_like(sd[k]) + (1e-30 if k.endswith("weight_v") else 0.0)
    net = SynthesizerTrn(*a, **kw).eval()
    net.load_state_dict(sd)
    zz = torch.zeros_like(z) if mode == "zero_input" else z
    for _ in range(3):
        net._engine.generator(zz, g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 15
    for _ in range(n):
        net._engine.generator(zz, g)
    torch.cuda.synchronize()
    print(f"{mode:14s} generator {1e3 * (time.perf_counter() - t0) / n:7.2f} ms per call")
    del net
