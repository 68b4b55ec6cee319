#!/usr/bin/env python3
"""Two batches in flight (experiment, round 6): the frame-rate half of batch k + 1 -- ~170 short launches that leave most of
the chip idle -- on a second stream and a second context while the generator of batch k fills the CUs.  Throughput of K
independent C3 batches with one stream / one context (bench.py's headline) against two.  Not the headline: a step there
is ONE batch start to end.  usage (GPU box): python tools/pipeline2.py [steps] [workload]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from vispeech_amd import config as vcfg
from vispeech_amd.models import SynthesizerTrn
from vispeech_amd.schema import ModelDims
from vispeech_amd.synth import synth_state_dict, workload

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
wl = sys.argv[2] if len(sys.argv) > 2 else "C3"
dims = ModelDims()
sd = synth_state_dict(dims, seed=1234, infer_only=True)
a, kw = vcfg.synthesizer_args(vcfg.default_hparams())
dev = torch.device("cuda:0")
batch = workload(wl)
t = lambda x: torch.from_numpy(np.asarray(x)).to(dev)
inp = dict(ph=t(batch["phonemes"]), ln=t(batch["lengths"]), sid=t(batch["sid"]), d=t(batch["duration"]), f0=t(batch["f0"]),
           en=t(batch["energy"]), noise=t(batch["noise"]))
tf = int(batch["frame_lengths"].max())
valid = 512 * int(batch["frame_lengths"].sum())


def run(n_ctx):
    nets = []
    for _ in range(n_ctx):
        m = SynthesizerTrn(*a, device=dev, **kw).eval()
        m.load_state_dict(sd)
        nets.append(m)
    streams = [torch.cuda.Stream(dev) for _ in range(n_ctx)]

    def step(k):
        i = k % n_ctx
        with torch.cuda.stream(streams[i]):
            return nets[i].infer(inp["ph"], inp["ln"], sid=inp["sid"], noise_scale=0.667, noise=inp["noise"], t_f=tf,
                                 duration_control=inp["d"], pitch_control=inp["f0"], energy_control=inp["en"])[0]
    for k in range(2 * n_ctx + 2):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del nets
    torch.cuda.empty_cache()
    return dt / steps * 1e3


res = {}
for rnd in range(2):
    for n in (1, 2):
        res.setdefault(f"{n}_in_flight_ms_per_batch", []).append(round(run(n), 3))
res["workload"] = wl
res["samples_per_s"] = {k: [round(valid / (x * 1e-3) / 1e6, 1) for x in v] for k, v in res.items() if k.endswith("ms_per_batch")}
print(json.dumps(res))
