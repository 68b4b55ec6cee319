#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference, imported read-only;
nothing of it is copied).  Inputs and weights come from this repo's own
seeded generators (vispeech_amd.synth); the reference model is built from its
unchanged configs/config.json, loaded with those weights, and its
``SynthesizerTrn.infer`` (reference models.py:672-722) is run on CPU with
``torch.randn_like`` patched to return the fixture noise.  Every file written
is data: inputs, expected outputs and intermediate stage tensors.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VISPEECH_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from vispeech_amd import config as vcfg                     # noqa: E402
from vispeech_amd.schema import dims_from_ctor, state_dict_schema  # noqa: E402
from vispeech_amd.synth import synth_batch, synth_state_dict       # noqa: E402

import models as ref_models                                  # noqa: E402  (reference)
import transforms as ref_transforms                          # noqa: E402  (reference)
import utils as ref_utils                                    # noqa: E402  (reference)
from text.symbols import symbols as ref_symbols              # noqa: E402  (reference)
from text import cleaned_text_to_sequence                    # noqa: E402  (reference)


def build_reference():
    hps = ref_utils.get_hparams_from_file(os.path.join(REF, "configs", "config.json"))
    # the repo's re-typed defaults must agree with the reference's config file
    mine = vcfg.default_hparams()
    for k, v in mine.model.items():
        assert hps.model[k] == v, ("config drift", k)
    for k in ("sampling_rate", "filter_length", "hop_length", "n_speakers"):
        assert hps.data[k] == mine.data[k], ("config drift", k)
    assert len(ref_symbols) == vcfg.N_SYMBOLS
    args, kwargs = vcfg.synthesizer_args(mine, len(ref_symbols))
    net = ref_models.SynthesizerTrn(*args, **kwargs).eval()
    dims = dims_from_ctor(*args, **kwargs)
    ref_sd = net.state_dict()
    schema = state_dict_schema(dims)
    assert list(schema.keys()) == list(ref_sd.keys()), "schema key order/content differs from the reference"
    for k, s in schema.items():
        assert tuple(ref_sd[k].shape) == tuple(s), (k, ref_sd[k].shape, s)
    sd = synth_state_dict(dims, seed=1234)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net, dims


class _Noise:
    """Patch ``torch.randn_like`` so the reference consumes the fixture noise (models.py:718)."""

    def __init__(self, noise):
        self.noise = torch.from_numpy(noise)

    def __enter__(self):
        self._orig = torch.randn_like
        torch.randn_like = lambda t, *a, **k: self.noise[:, :, :t.shape[2]].to(t.dtype).clone()
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig


def run_case(net, name, batch, *, use_duration=True, use_pitch=True, use_energy=True,
             noise_scale=0.667, max_len=None, scalar_controls=None, dur_3d=False, live_caller=False):
    stages = {}

    def hook(tag):
        def fn(_m, _inp, out):
            o = out[0] if isinstance(out, tuple) else out
            stages[tag] = o.detach().clone().numpy()
        return fn

    hs = [net.enc_p.register_forward_hook(hook("x_enc")),
          net.lr.register_forward_hook(hook("x_frame")),
          net.frame_prior_net.register_forward_hook(hook("h_frame_t"))]
    ph = torch.from_numpy(batch["phonemes"])
    ln = torch.from_numpy(batch["lengths"])
    sid = torch.from_numpy(batch["sid"])
    sc = scalar_controls or {}
    dur = torch.from_numpy(batch["duration"]) if use_duration else sc.get("duration")
    if use_duration and dur_3d:
        dur = dur[:, None, :]
    kw = dict(sid=sid, noise_scale=noise_scale, max_len=max_len,
              duration_control=dur,
              pitch_control=torch.from_numpy(batch["f0"]) if use_pitch else sc.get("pitch"),
              energy_control=torch.from_numpy(batch["energy"]) if use_energy else sc.get("energy"))
    if not use_duration:
        # predicted durations decide T_f: provide ample noise
        noise = np.random.Generator(np.random.PCG64(777)).standard_normal(
            (ph.shape[0], 192, 4096), dtype=np.float32)
    else:
        noise = batch["noise"]
    with torch.no_grad(), _Noise(noise):
        if live_caller:
            # EXACTLY the reference's one live call (train.py:281, 300-301: `for shift, energy_shift in [(1, 1)]`):
            # durations predicted (duration_control left at None), Python-int scalar controls, max_len=1000, the
            # default noise_scale -- on a ragged batch instead of the [:1] slice of train.py:289-293
            shift, energy_shift = 1, 1
            o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy = net.infer(
                ph, ln, max_len=1000, sid=sid, pitch_control=shift, energy_control=energy_shift)
        else:
            o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy = net.infer(ph, ln, **kw)
    for h in hs:
        h.remove()
    tf = x_mask.shape[2]
    out = dict(
        in_phonemes=batch["phonemes"], in_lengths=batch["lengths"], in_sid=batch["sid"],
        in_noise=np.ascontiguousarray(noise[:, :, :tf]), in_noise_scale=np.float32(noise_scale),
        in_max_len=np.int64(-1 if max_len is None else max_len),
        in_use=np.array([use_duration, use_pitch, use_energy]),
        in_scalar=np.array([sc.get("duration", 1) or 1, sc.get("pitch", 1) or 1, sc.get("energy", 1) or 1],
                           dtype=np.float32),
        in_duration=batch["duration"], in_f0=batch["f0"], in_energy=batch["energy"],
        o=o.numpy(), x_mask=x_mask.numpy(), z=z.numpy(), z_p=z_p.numpy(), m_p=m_p.numpy(),
        logs_p=logs_p.numpy(), duration=duration.reshape(ph.shape[0], -1).numpy(), F0=f0.numpy(),
        energy=energy.numpy(), x_enc=stages["x_enc"], x_frame=stages["x_frame"],
        h_frame=np.ascontiguousarray(np.transpose(stages["h_frame_t"], (0, 2, 1))),
    )
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: B={ph.shape[0]} Tp={ph.shape[1]} Tf={tf} samples={o.shape[-1]} "
          f"|o|max={np.abs(out['o']).max():.4f} |z|max={np.abs(out['z']).max():.3f} -> {os.path.getsize(path)/1024:.0f} KiB")
    return out


def spline_case():
    r = np.random.Generator(np.random.PCG64(4242))
    n, nb = 512, 10
    x = r.uniform(-7.0, 7.0, size=(2, 1, n)).astype(np.float32)       # some outside the +-5 tails
    x[0, 0, :4] = [-5.0, 5.0, 0.0, 4.999999]
    uw = r.standard_normal((2, 1, n, nb)).astype(np.float32)
    uh = r.standard_normal((2, 1, n, nb)).astype(np.float32)
    ud = r.standard_normal((2, 1, n, nb - 1)).astype(np.float32)
    res = {}
    for inv in (False, True):
        y, lad = ref_transforms.piecewise_rational_quadratic_transform(
            torch.from_numpy(x), torch.from_numpy(uw), torch.from_numpy(uh), torch.from_numpy(ud),
            inverse=inv, tails="linear", tail_bound=5.0)
        res["y_inv" if inv else "y_fwd"] = y.numpy()
        res["lad_inv" if inv else "lad_fwd"] = lad.numpy()
    np.savez_compressed(os.path.join(HERE, "spline.npz"), x=x, uw=uw, uh=uh, ud=ud, **res)
    print("spline: n=%d" % (2 * n))


def filelist_case(net):
    """C1 plumbing case cut from the reference's filelists/train.list (real ja phonemes,
    MFA durations, per-phoneme F0/energy).  Pick the shortest row."""
    rows = [l.rstrip("\n").split("|") for l in open(os.path.join(REF, "filelists", "train.list"), encoding="utf-8")]
    rows.sort(key=lambda r: sum(int(x) for x in r[3].split()))
    spk, uid, phones, durs, f0s, ens = rows[0]
    ids = np.array([cleaned_text_to_sequence(phones.split())], dtype=np.int64)
    d = np.array([[float(x) for x in durs.split()]], dtype=np.float32)
    n = ids.shape[1]
    tf = int(d.sum())
    noise = np.random.Generator(np.random.PCG64(101)).standard_normal((1, 192, tf), dtype=np.float32)
    batch = dict(phonemes=ids, lengths=np.array([n], dtype=np.int64), sid=np.array([64], dtype=np.int64),
                 duration=d, f0=np.array([[float(x) for x in f0s.split()]], dtype=np.float32),
                 energy=np.array([[float(x) for x in ens.split()]], dtype=np.float32), noise=noise)
    run_case(net, "c1_filelist", batch)


def vallist_case(net):
    """SURVEY 8f row 2, end to end: the two shortest rows of the reference's filelists/val.list go through the
    REFERENCE's own front door (text.cleaned_text_to_sequence, data_utils.py:94-102 field parsing, spk2id of
    configs/config.json) into SynthesizerTrn.infer as one ragged batch.  The fixture keeps the raw rows and the
    symbol table (data) so that the test can drive vispeech_amd.text over the same lines."""
    import json
    lines = [l.rstrip("\n") for l in open(os.path.join(REF, "filelists", "val.list"), encoding="utf-8")]
    lines.sort(key=lambda l: sum(int(x) for x in l.split("|")[3].split()))
    lines = lines[:2]
    spk2id = json.load(open(os.path.join(REF, "configs", "config.json"), encoding="utf-8"))["data"]["spk2id"]
    rows = [l.split("|") for l in lines]
    B, tp = len(rows), max(len(r[2].split(" ")) for r in rows)
    ids = np.zeros((B, tp), np.int64)
    dur, f0, en = (np.zeros((B, tp), np.float32) for _ in range(3))
    lens, sid = np.zeros(B, np.int64), np.zeros(B, np.int64)
    for b, (spk, uid, phones, durs, f0s, ens) in enumerate(rows):
        seq = cleaned_text_to_sequence(phones.split(" "))
        n = len(seq)
        ids[b, :n], lens[b], sid[b] = seq, n, spk2id[spk]
        dur[b, :n] = [int(x) for x in durs.split(" ")]
        f0[b, :n] = [float(x) for x in f0s.strip().split(" ")]
        en[b, :n] = [float(x) for x in ens.strip().split(" ")]
    tf = int(dur.sum(axis=1).max())
    noise = np.random.Generator(np.random.PCG64(8642)).standard_normal((B, 192, tf), dtype=np.float32)
    batch = dict(phonemes=ids, lengths=lens, sid=sid, duration=dur, f0=f0, energy=en, noise=noise)
    out = run_case(net, "val_filelist", batch)
    path = os.path.join(HERE, "val_filelist.npz")
    keep = ("in_phonemes", "in_lengths", "in_sid", "in_noise", "in_noise_scale", "in_duration", "in_f0", "in_energy",
            "o", "x_mask", "z", "m_p", "logs_p")
    np.savez_compressed(path, rows=np.array(lines), symbols=np.array(list(ref_symbols)),
                        spk2id_keys=np.array(list(spk2id.keys())), spk2id_vals=np.array(list(spk2id.values()), dtype=np.int64),
                        **{k: out[k] for k in keep})
    print(f"val_filelist: rows {[r[1] for r in rows]} -> {os.path.getsize(path)/1024:.0f} KiB")


def vc_case(net, dims):
    """SynthesizerTrn.voice_conversion (reference models.py:724-732) on a ragged batch of synthetic
    linear spectrograms (|N(0,1)| magnitudes, the scale of spectrogram_torch output on speech)."""
    r = np.random.Generator(np.random.PCG64(2024))
    lens = np.array([21, 30, 9], dtype=np.int64)
    B, T = len(lens), int(lens.max())
    y = np.abs(r.standard_normal((B, dims.spec_channels, T))).astype(np.float32)
    for b, n in enumerate(lens):
        y[b, :, n:] = 0.0
    sid_src = np.array([3, 17, 40], dtype=np.int64)
    sid_tgt = np.array([8, 17, 2], dtype=np.int64)
    noise = r.standard_normal((B, dims.inter_channels, T)).astype(np.float32)
    stages = {}
    h = net.enc_q.register_forward_hook(lambda _m, _i, out: stages.update(m_q=out[1].detach().numpy().copy(),
                                                                          logs_q=out[2].detach().numpy().copy()))
    with torch.no_grad(), _Noise(noise):
        o_hat, y_mask, (z, z_p, z_hat) = net.voice_conversion(
            torch.from_numpy(y), torch.from_numpy(lens), torch.from_numpy(sid_src), torch.from_numpy(sid_tgt))
    h.remove()
    path = os.path.join(HERE, "voice_conversion.npz")
    np.savez_compressed(path, in_y=y, in_lengths=lens,
                        in_sid_src=sid_src, in_sid_tgt=sid_tgt, in_noise=noise, o_hat=o_hat.numpy(),
                        y_mask=y_mask.numpy(), z=z.numpy(), z_p=z_p.numpy(), z_hat=z_hat.numpy(), **stages)
    print(f"voice_conversion: B={B} T={T} |o|max={np.abs(o_hat.numpy()).max():.4f} |z_p|max={np.abs(z_p.numpy()).max():.3f}"
          f" -> {os.path.getsize(path)/1024:.0f} KiB")


def main():
    """``make_golden.py`` rewrites everything; ``make_golden.py vc`` only voice_conversion.npz, ``vallist`` only
    val_filelist.npz, ``evaluate`` only evaluate_caller.npz."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    net, dims = build_reference()
    only = set(sys.argv[1:])
    if not only or "infer" in only:
        # ragged batch, controls supplied (the throughput configuration)
        b3 = synth_batch(3, seed=11, mean_phonemes=9, std_phonemes=3, min_phonemes=5, max_phonemes=12,
                         mean_frames=34, jitter_frames=6)
        run_case(net, "ragged_controls", b3)
        # same inputs, every predictor on (scalar controls), durations predicted
        run_case(net, "ragged_predictors", b3, use_duration=False, use_pitch=False, use_energy=False,
                 scalar_controls=dict(duration=0.25, pitch=1.1, energy=0.9), noise_scale=0.5)
        # max_len truncation + [B,1,Tp] duration tensor + pitch predicted only
        run_case(net, "maxlen_dur3d", b3, use_pitch=False, max_len=20, dur_3d=True, noise_scale=1.0)
        filelist_case(net)
    if not only or "evaluate" in only:
        # the reference's live caller form (train.py:300-301)
        b3 = synth_batch(3, seed=11, mean_phonemes=9, std_phonemes=3, min_phonemes=5, max_phonemes=12,
                         mean_frames=34, jitter_frames=6)
        run_case(net, "evaluate_caller", b3, use_duration=False, use_pitch=False, use_energy=False,
                 noise_scale=1.0, max_len=1000, live_caller=True)
    if not only or "vallist" in only:
        vallist_case(net)
    if not only or "spline" in only:
        spline_case()
    if not only or "vc" in only:
        vc_case(net, dims)


if __name__ == "__main__":
    main()
