#!/usr/bin/env python3
"""Time to first audio of the service path: SynthesisService.stream on one ~5.6 s utterance and on
the 60 s utterance of BASELINE config 5, chunk = 64 frames (0.74 s of audio) -- the latency from the call to the first
PCM16 chunk on the HOST, the steady chunk period, the total, beside the one-shot call (SynthesisService.synthesize =
what the reference's /tts prints as "inference time", reference inference_api.py:43-54, and returns only when the whole
waveform exists).  usage (on the GPU box): python tools/ttfa.py [chunk_frames] > gpurun_out/<tag>/ttfa.json"""
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
from vispeech_amd.service import SynthesisService
from vispeech_amd.synth import synth_state_dict, workload


def measure(svc, batch, runs=5):
    noise = torch.from_numpy(batch["noise"]).to(svc.net.device)
    out = []
    for _ in range(runs + 2):                                   # two warm-up passes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stamps, nbytes = [], 0
        for piece in svc.stream(batch, 0, noise=noise):
            stamps.append(time.perf_counter() - t0)
            nbytes += len(piece)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pcm = svc.synthesize(batch, 0, noise=noise)
        t2 = time.perf_counter()
        out.append((stamps, nbytes, t2 - t1, pcm.size))
    out = out[2:]
    first = sorted(s[0][0] for s in out)[len(out) // 2]
    total = sorted(s[0][-1] for s in out)[len(out) // 2]
    n = len(out[0][0])
    period = sorted((s[0][-1] - s[0][0]) / max(n - 1, 1) for s in out)[len(out) // 2]
    oneshot = sorted(s[2] for s in out)[len(out) // 2]
    samples = out[0][1] // 2
    assert samples == out[0][3]
    return {"frames": int(batch["frame_lengths"][0]), "audio_s": samples / 44100.0, "chunks": n,
            "first_chunk_ms": first * 1e3, "chunk_period_ms": period * 1e3, "stream_total_ms": total * 1e3,
            "one_shot_ms": oneshot * 1e3, "chunk_audio_ms": svc.chunk_frames * 512 / 44.1,
            "first_chunk_vs_one_shot": first / oneshot}


def main():
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dims = ModelDims()
    a, kw = vcfg.synthesizer_args(vcfg.default_hparams())
    net = SynthesizerTrn(*a, **kw).eval()
    net.load_state_dict(synth_state_dict(dims, seed=1234, infer_only=True))
    svc = SynthesisService(net, chunk_frames=chunk)
    res = {"chunk_frames": chunk, "what": "SynthesisService.stream: call -> first PCM16 chunk on the host (median of 5 after 2 warm-ups); "
                                          "one_shot = SynthesisService.synthesize of the same request (the reference's 'inference time')"}
    res["one_utterance_C2_first"] = measure(svc, workload("C2", batch=1))
    res["C5_60s"] = measure(svc, workload("C5"))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
