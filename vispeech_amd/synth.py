"""Deterministic synthetic weights and inputs (reference-independent).

No pretrained checkpoint of the reference exists (its apps point at local
paths, reference inference.py:36), and its random init makes the flow an
identity shift (zero-initialised ``post``, reference modules.py:321-322) and
the vocoder output ~1e-6 (``init_weights`` std 0.01, reference
commons.py:8-11), so parity on random-init weights would be vacuous
(SURVEY.md gotcha G18).  This module draws every tensor of the checkpoint
schema from its own ``numpy`` PCG64 stream keyed by ``(seed, crc32(key))`` with
gains chosen so that every stage carries O(1) signal, and generates phoneme /
duration / F0 / energy / noise batches with the statistics of the reference's
``filelists/train.list``.
"""
from __future__ import annotations

import zlib
from typing import Dict, Optional

import numpy as np

from .schema import ModelDims, state_dict_schema

# symbol-id ranges of the reference table (text/symbols.py:39): "_"=0, zh 1..401,
# ja 402..443, en 444..512, punctuation 513..518.
ZH_RANGE = (1, 402)
JA_RANGE = (402, 444)
PU_RANGE = (513, 519)


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode("utf-8"))]))


def _fan_in(shape) -> int:
    n = 1
    for x in shape[1:]:
        n *= x
    return max(n, 1)


def synth_state_dict(dims: ModelDims, seed: int = 1234,
                     infer_only: bool = False) -> Dict[str, np.ndarray]:
    """Synthetic checkpoint ``{key: float32 ndarray}`` following ``state_dict_schema``.

    ``weight_g`` is ``||v|| * U(0.8, 1.2)`` along the weight-norm axis so that folding
    ``g * v / ||v||`` is a non-trivial rescale (exercises the ConvTranspose dim-0 case,
    SURVEY.md gotcha G3).  Must be generated in schema order because ``weight_g`` reads the
    ``weight_v`` drawn just before it.
    """
    from .schema import used_by_infer
    schema = state_dict_schema(dims)
    out: Dict[str, np.ndarray] = {}
    pending_g = {}
    for key, shape in schema.items():
        if infer_only and not used_by_infer(key):
            continue
        r = _rng(seed, key)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "weight_g":
            pending_g[key] = (shape, r)
            continue
        if key == "enc_p.symbol_emb.weight":
            t = r.standard_normal(shape) * dims.hidden_channels ** -0.5
        elif key == "emb_g.weight" or key == "frame_prior_net.emb.weight":
            t = r.standard_normal(shape)
        elif leaf in ("emb_rel_k", "emb_rel_v"):
            t = r.standard_normal(shape) * shape[-1] ** -0.5
        elif leaf in ("gamma",) or key.endswith("layer_norm_1.weight") or key.endswith("layer_norm_2.weight"):
            t = 1.0 + 0.1 * r.standard_normal(shape)
        elif leaf in ("beta",) or key.endswith("layer_norm_1.bias") or key.endswith("layer_norm_2.bias"):
            t = 0.1 * r.standard_normal(shape)
        elif leaf == "bias":
            t = 0.05 * r.standard_normal(shape)
            if key == "duration_predictor.proj.bias":
                t = t + 1.5          # exp(1.5)-1 ~ 3.5 frames / phoneme before duration_control
            if key == "pitch_predictor.proj_f0.bias":
                t = t + 0.8          # LF0 of ~300 Hz
        elif leaf in ("weight", "weight_v"):
            if key.startswith("dec.ups.") and leaf == "weight_v":
                # ConvTranspose1d weight [in, out, k]: each output sample sees in*k/stride taps
                i = int(key.split(".")[2])
                fan = shape[0] * shape[2] / dims.upsample_rates[i]
                gain = 1.0
            elif key.startswith("dec.resblocks."):
                fan, gain = _fan_in(shape), 0.8
            elif key == "dec.conv_post.weight":
                fan, gain = _fan_in(shape), 0.15
            elif key.startswith("flow.") and ".post." in key:
                fan, gain = _fan_in(shape), 0.2
            elif key == "project.proj.weight":
                fan, gain = _fan_in(shape), 0.3
            elif key == "enc_q.proj.weight":
                fan, gain = _fan_in(shape), 0.15   # keeps exp(logs_q) of the 16-layer WN skip sum O(1)
            else:
                fan, gain = _fan_in(shape), 1.0
            t = r.standard_normal(shape) * (gain / np.sqrt(fan))
        else:  # pragma: no cover - schema and this table must stay in sync
            raise KeyError(f"no synthetic rule for {key}")
        out[key] = np.ascontiguousarray(t, dtype=np.float32)
        gk = key[:-1] + "g" if leaf == "weight_v" else None
        if gk is not None and gk in pending_g:
            gshape, gr = pending_g.pop(gk)
            v = out[key].astype(np.float64)
            norm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(axis=1)).reshape(gshape)
            out[gk] = np.ascontiguousarray(norm * gr.uniform(0.8, 1.2, size=gshape), dtype=np.float32)
    # weight_g precedes weight_v in schema order for some modules: resolve the stragglers
    for gk, (gshape, gr) in pending_g.items():
        vk = gk[:-1] + "v"
        v = out[vk].astype(np.float64)
        norm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(axis=1)).reshape(gshape)
        out[gk] = np.ascontiguousarray(norm * gr.uniform(0.8, 1.2, size=gshape), dtype=np.float32)
    # keep schema order
    return {k: out[k] for k in schema if k in out}


def synth_batch(batch: int, seed: int, *, mean_phonemes: float = 40.0, std_phonemes: float = 10.0,
                min_phonemes: int = 10, max_phonemes: int = 80, mean_frames: float = 430.0,
                jitter_frames: float = 60.0, fixed_phonemes: Optional[int] = None,
                fixed_frames: Optional[int] = None, languages: str = "zh",
                hidden: int = 192) -> Dict[str, np.ndarray]:
    """A padded utterance batch with the statistics of SURVEY.md section 8d.

    Returns int64 ``phonemes [B,Tp]``, ``lengths [B]``, ``sid [B]``, float32
    ``duration/f0/energy [B,Tp]`` (zero in the padding), ``frame_lengths [B]`` and the
    reparameterisation ``noise [B,hidden,Tf]`` with ``Tf = max(frame_lengths)``.
    """
    r = np.random.Generator(np.random.PCG64(seed))
    if fixed_phonemes is not None:
        lengths = np.full(batch, fixed_phonemes, dtype=np.int64)
    else:
        lengths = np.clip(np.rint(r.normal(mean_phonemes, std_phonemes, size=batch)),
                          min_phonemes, max_phonemes).astype(np.int64)
    tp = int(lengths.max())
    phonemes = np.zeros((batch, tp), dtype=np.int64)
    duration = np.zeros((batch, tp), dtype=np.float32)
    f0 = np.zeros((batch, tp), dtype=np.float32)
    energy = np.zeros((batch, tp), dtype=np.float32)
    for b in range(batch):
        n = int(lengths[b])
        lang = languages if languages in ("zh", "ja") else ("zh" if b % 2 == 0 else "ja")
        lo, hi = ZH_RANGE if lang == "zh" else JA_RANGE
        ids = r.integers(lo, hi, size=n)
        pu = r.integers(PU_RANGE[0], PU_RANGE[1], size=n)
        is_pu = (np.arange(n) % 8) == 7
        phonemes[b, :n] = np.where(is_pu, pu, ids)
        target = fixed_frames if fixed_frames is not None else \
            int(np.clip(np.rint(mean_frames + r.uniform(-jitter_frames, jitter_frames)), n, None))
        d = r.gamma(2.0, 5.5, size=n)
        d[r.random(n) < 0.014] = 0.0
        if d.sum() <= 0:
            d[:] = 1.0
        d = np.floor(d * (target / d.sum()))
        # hand the rounding remainder to the longest phoneme so sum(d) == target exactly
        d[int(np.argmax(d))] += target - d.sum()
        duration[b, :n] = d
        v = r.uniform(150.0, 400.0, size=n)
        v[r.random(n) < 0.10] = 0.0
        f0[b, :n] = v
        energy[b, :n] = r.uniform(0.0, 100.0, size=n)
    frame_lengths = duration.sum(axis=1).astype(np.int64)
    tf = int(frame_lengths.max())
    noise = r.standard_normal((batch, hidden, tf), dtype=np.float32)
    sid = r.integers(0, 67, size=batch).astype(np.int64)
    return dict(phonemes=phonemes, lengths=lengths, sid=sid, duration=duration, f0=f0,
                energy=energy, frame_lengths=frame_lengths, noise=noise)


# the five BASELINE.json configurations (C1..C5)
WORKLOADS = {
    "C1": dict(batch=1, seed=101, languages="zh"),
    "C2": dict(batch=16, seed=102, languages="zh"),
    "C3": dict(batch=64, seed=103, languages="mixed"),
    "C4": dict(batch=256, seed=104, languages="mixed"),
    "C5": dict(batch=1, seed=105, languages="zh", fixed_phonemes=470, fixed_frames=5168),
}


def workload(name: str, **override) -> Dict[str, np.ndarray]:
    kw = dict(WORKLOADS[name])
    kw.update(override)
    return synth_batch(**kw)
