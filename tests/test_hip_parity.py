"""Parity of the HIP path (through the C-ABI) against the golden vectors produced by the real
reference and against the CPU oracle on seeded inputs.  Needs an MI355X: `pytest -m gpu`.

Tolerances: integer-valued outputs exact; floating-point stage boundaries
<= STAGE_TOL * max|ref| ; waveform <= WAVE_TOL * max|ref|.  The kernels compute in f32 on the
matrix core (v_mfma_f32_32x32x2_f32 = exact fmaf chain), so the only difference from the
reference's fp32 CPU path is summation order.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STAGE_TOL = 1e-5
WAVE_TOL = 1e-4

CASES = ["ragged_controls", "ragged_predictors", "maxlen_dur3d", "c1_filelist", "evaluate_caller"]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def to_np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def dims():
    from vispeech_amd.schema import ModelDims
    return ModelDims()


@pytest.fixture(scope="module")
def weights(dims):
    from vispeech_amd.synth import synth_state_dict
    return synth_state_dict(dims, seed=1234, infer_only=True)


@pytest.fixture(scope="module")
def net(dims, weights):
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    m = SynthesizerTrn(*args, **kwargs).eval()
    m.load_state_dict(weights, strict=True)
    return m


@pytest.fixture(scope="module")
def oracle(dims, weights):
    from oracle.vispeech_oracle import Oracle
    return Oracle(weights, dims)


def golden(golden_dir, case):
    return np.load(os.path.join(golden_dir, f"{case}.npz"))


def run_case(net, g):
    use, sc = g["in_use"], g["in_scalar"]
    max_len = int(g["in_max_len"])
    dev = net.device
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    return net.infer(
        t(g["in_phonemes"]), t(g["in_lengths"]), sid=t(g["in_sid"]), noise_scale=float(g["in_noise_scale"]),
        max_len=None if max_len < 0 else max_len,
        duration_control=t(g["in_duration"]) if use[0] else float(sc[0]),
        pitch_control=t(g["in_f0"]) if use[1] else float(sc[1]),
        energy_control=t(g["in_energy"]) if use[2] else float(sc[2]),
        noise=t(g["in_noise"]))


@pytest.mark.parametrize("case", CASES)
def test_infer_matches_reference_golden(net, golden_dir, case):
    """Full SynthesizerTrn.infer through the shim vs the outputs of the real reference."""
    g = golden(golden_dir, case)
    o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy = run_case(net, g)
    assert x_mask.dtype == torch.bool
    np.testing.assert_array_equal(to_np(x_mask), g["x_mask"])
    np.testing.assert_array_equal(to_np(duration).reshape(g["duration"].shape), g["duration"])
    errs = {}
    for name, val in (("m_p", m_p), ("logs_p", logs_p), ("z_p", z_p), ("z", z), ("F0", f0), ("energy", energy)):
        errs[name] = rel_err(to_np(val), g[name])
    errs["o"] = rel_err(to_np(o), g["o"])
    print(case, {k: f"{v:.2e}" for k, v in errs.items()})
    for name in ("m_p", "logs_p", "z_p", "z", "F0", "energy"):
        assert errs[name] <= STAGE_TOL, (case, name, errs[name])
    assert errs["o"] <= WAVE_TOL, (case, errs["o"])


def test_live_caller_form_matches_reference_golden(net, golden_dir):
    """The reference's ONE live call of infer (train.py:281, 300-301), argument for argument:
    `infer(phonemes, phonemes_lengths, max_len=1000, sid=sid, pitch_control=shift, energy_control=energy_shift)` with
    `shift, energy_shift = 1, 1` (Python ints), durations predicted (duration_control omitted), the default noise_scale
    -- predicted durations + integer scalar controls + max_len in one call, on a ragged batch."""
    g = golden(golden_dir, "evaluate_caller")
    assert not g["in_use"].any() and int(g["in_max_len"]) == 1000 and float(g["in_noise_scale"]) == 1.0
    t = lambda a: torch.from_numpy(np.asarray(a)).to(net.device)
    shift, energy_shift = 1, 1
    o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy = net.infer(
        t(g["in_phonemes"]), t(g["in_lengths"]), max_len=1000, sid=t(g["in_sid"]), pitch_control=shift,
        energy_control=energy_shift, noise=t(g["in_noise"]))
    np.testing.assert_array_equal(to_np(duration).reshape(g["duration"].shape), g["duration"])      # predicted durations: exact
    np.testing.assert_array_equal(to_np(x_mask), g["x_mask"])
    # the reference's next line (train.py:302): y_hat_lengths = mask.sum([1, 2]).long() * hop_length
    np.testing.assert_array_equal(to_np(x_mask.sum([1, 2]).long() * 512), g["x_mask"].sum((1, 2)) * 512)
    for name, val in (("m_p", m_p), ("logs_p", logs_p), ("z_p", z_p), ("z", z), ("F0", f0), ("energy", energy)):
        assert rel_err(to_np(val), g[name]) <= STAGE_TOL, name
    assert o.shape == g["o"].shape and rel_err(to_np(o), g["o"]) <= WAVE_TOL


@pytest.mark.parametrize("case", ["ragged_controls", "c1_filelist"])
def test_stage_text_encoder(net, weights, dims, golden_dir, case):
    g = golden(golden_dir, case)
    emb = weights["enc_p.symbol_emb.weight"][g["in_phonemes"]] * np.float32(math.sqrt(dims.hidden_channels))
    x = np.ascontiguousarray(np.transpose(emb, (0, 2, 1)))
    y = net._engine.encoder(0, x, g["in_lengths"])
    e = rel_err(to_np(y), g["x_enc"])
    print("x_enc", case, f"{e:.2e}")
    assert e <= STAGE_TOL


@pytest.mark.parametrize("case", CASES)
def test_stage_length_regulator_exact(net, oracle, golden_dir, case):
    """The expand is a pure gather: bit-exact against the reference's x_frame when fed the
    oracle's (reference-pinned) phoneme-rate tensor."""
    g = golden(golden_dir, case)
    use, sc = g["in_use"], g["in_scalar"]
    enc = oracle.encode(g["in_phonemes"], g["in_lengths"], g["in_sid"],
                        g["in_duration"] if use[0] else float(sc[0]), g["in_f0"] if use[1] else float(sc[1]),
                        g["in_energy"] if use[2] else float(sc[2]))
    d = enc["duration"].reshape(g["in_phonemes"].shape).numpy()
    cum = np.cumsum(np.maximum(np.trunc(d), 0).astype(np.int64), axis=1).astype(np.int32)
    tf = g["x_frame"].shape[2]
    y = net._engine.length_regulate(enc["x_var"], cum, tf)
    np.testing.assert_array_equal(to_np(y), enc["x_frame"].numpy())
    assert rel_err(to_np(y), g["x_frame"]) <= 1e-5


@pytest.mark.parametrize("case", ["ragged_controls", "ragged_predictors"])
def test_stage_frame_prior(net, golden_dir, case):
    g = golden(golden_dir, case)
    lens = g["x_mask"][:, 0, :].sum(axis=1).astype(np.int64)
    y = net._engine.encoder(2, g["x_frame"], lens)
    e = rel_err(to_np(y), g["h_frame"])
    print("h_frame", case, f"{e:.2e}")
    assert e <= STAGE_TOL


@pytest.mark.parametrize("case", ["ragged_controls", "c1_filelist"])
def test_stage_flow(net, weights, golden_dir, case):
    g = golden(golden_dir, case)
    lens = g["x_mask"][:, 0, :].sum(axis=1).astype(np.int64)
    gvec = weights["emb_g.weight"][g["in_sid"]]
    z = net._engine.flow_reverse(g["z_p"], gvec, lens)
    e = rel_err(to_np(z), g["z"])
    print("flow", case, f"{e:.2e}")
    assert e <= STAGE_TOL


@pytest.mark.parametrize("case", CASES)
def test_stage_generator(net, weights, golden_dir, case):
    g = golden(golden_dir, case)
    max_len = int(g["in_max_len"])
    zin = g["z"] * g["x_mask"].astype(np.float32)
    if max_len >= 0:
        zin = zin[:, :, :max_len]
    gvec = weights["emb_g.weight"][g["in_sid"]]
    o = net._engine.generator(np.ascontiguousarray(zin), gvec)
    e = rel_err(to_np(o), g["o"])
    print("generator", case, f"{e:.2e}")
    assert e <= WAVE_TOL


def test_spline_matches_reference_golden(golden_dir):
    from vispeech_amd.engine import rq_spline
    g = np.load(os.path.join(golden_dir, "spline.npz"))
    for inv, yk, lk in ((False, "y_fwd", "lad_fwd"), (True, "y_inv", "lad_inv")):
        y, lad = rq_spline(g["x"], g["uw"], g["uh"], g["ud"], inverse=inv, tail_bound=5.0)
        ey = np.abs(to_np(y) - g[yk]).max()
        el = np.abs(to_np(lad) - g[lk]).max()
        print("spline", inv, f"{ey:.2e} {el:.2e}")
        assert ey <= 1e-4 and el <= 1e-4
    y, _ = rq_spline(g["x"], g["uw"], g["uh"], g["ud"], inverse=False)
    x2, _ = rq_spline(y, g["uw"], g["uh"], g["ud"], inverse=True)
    assert np.abs(to_np(x2) - g["x"]).max() <= 1e-3


def _oracle_vs_hip(net, oracle, batch, **kw):
    dev = net.device
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    ref = oracle.infer(batch["phonemes"], batch["lengths"], batch["sid"], noise=batch["noise"], noise_scale=0.667,
                       duration_control=batch["duration"], pitch_control=batch["f0"], energy_control=batch["energy"], **kw)
    o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy = net.infer(
        t(batch["phonemes"]), t(batch["lengths"]), sid=t(batch["sid"]), noise_scale=0.667,
        duration_control=t(batch["duration"]), pitch_control=t(batch["f0"]), energy_control=t(batch["energy"]),
        noise=t(batch["noise"]), **kw)
    return ref, dict(o=o, x_mask=x_mask, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p)


def test_oracle_parity_medium_batch(net, oracle):
    """Seeded ragged batch larger than the golden cases (multi-tile in every kernel: T_f > 128,
    generator rows > 1024 columns) against the CPU oracle."""
    from vispeech_amd.synth import synth_batch
    batch = synth_batch(3, seed=21, mean_phonemes=24, std_phonemes=6, min_phonemes=12, max_phonemes=40,
                        mean_frames=150, jitter_frames=30)
    ref, out = _oracle_vs_hip(net, oracle, batch)
    np.testing.assert_array_equal(to_np(out["x_mask"]), ref["x_mask"].numpy())
    for name in ("m_p", "logs_p", "z_p", "z"):
        e = rel_err(to_np(out[name]), ref[name].numpy())
        print(name, f"{e:.2e}")
        assert e <= STAGE_TOL, (name, e)
    e = rel_err(to_np(out["o"]), ref["o"].numpy())
    print("o", f"{e:.2e}")
    assert e <= WAVE_TOL


def test_sharded_batch_equals_unsharded(net):
    """SURVEY gotcha G6: with every shard padded to the global T_f, per-utterance results do not
    depend on which shard they ran in -- bit-exact HERE, where both sides are small enough to select the same kernels
    (same tiles per utterance); at other shard sizes within the parity tolerance: the test below."""
    from vispeech_amd.synth import synth_batch
    batch = synth_batch(4, seed=33, mean_phonemes=16, std_phonemes=4, min_phonemes=8, max_phonemes=24,
                        mean_frames=70, jitter_frames=20)
    dev = net.device
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    tf = int(batch["frame_lengths"].max())

    def run(sl):
        return net.infer(t(batch["phonemes"][sl]), t(batch["lengths"][sl]), sid=t(batch["sid"][sl]), noise_scale=0.667,
                         duration_control=t(batch["duration"][sl]), pitch_control=t(batch["f0"][sl]),
                         energy_control=t(batch["energy"][sl]), noise=t(batch["noise"][sl]), t_f=tf)
    full = run(slice(0, 4))
    a, b = run(slice(0, 2)), run(slice(2, 4))
    np.testing.assert_array_equal(to_np(full[0]), np.concatenate([to_np(a[0]), to_np(b[0])], axis=0))
    np.testing.assert_array_equal(to_np(full[2][0]), np.concatenate([to_np(a[2][0]), to_np(b[2][0])], axis=0))


def test_c3_rank_slice_matches_the_unsharded_run(net):
    """Shard-size independence, stated truthfully: utterances [24, 32) of the C3 batch run as the
    rank-3-of-8 slice (global frame padding, its own rows of the noise) against the same rows of the N = 1 run.  The
    frame-rate convolutions pick their kernel from the size of the LAUNCH (conv_mfma.hip launch_conv: the channel-split
    and latency forms on grids that do not fill the chip), and a split-k sum adds in another order -- so a shard is NOT
    bit-equal to the unsharded run in general (it is when both sides select the same kernels: the 4 -> 2 + 2 test
    above).  What holds at every shard size is the parity tolerance itself, with a wide margin: 1e-5 on the latent,
    1e-5 of the peak on the waveform (measured 2e-7 / 1e-6)."""
    from vispeech_amd.sharding import shard_range
    from vispeech_amd.synth import workload
    b = workload("C3")
    dev = net.device
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    tf = int(b["frame_lengths"].max())

    def run(sl):
        o, _, (z, *_), *_ = net.infer(t(b["phonemes"][sl]), t(b["lengths"][sl]), sid=t(b["sid"][sl]), noise_scale=0.667,
                                      duration_control=t(b["duration"][sl]), pitch_control=t(b["f0"][sl]),
                                      energy_control=t(b["energy"][sl]), noise=t(b["noise"][sl]), t_f=tf)
        return to_np(o), to_np(z)
    lo, hi = shard_range(64, 3, 8)
    assert (lo, hi) == (24, 32)
    o_all, z_all = run(slice(0, 64))
    o_sh, z_sh = run(slice(lo, hi))
    assert o_sh.shape == o_all[lo:hi].shape
    ez, eo = rel_err(z_sh, z_all[lo:hi]), rel_err(o_sh, o_all[lo:hi])
    print(f"rank 3 of 8 vs unsharded: z {ez:.2e}  o {eo:.2e}  bit-equal: {np.array_equal(o_sh, o_all[lo:hi])}")
    assert ez <= STAGE_TOL and eo <= 1e-5


def test_valid_region_properties(net):
    """Size-independent properties: outputs finite and within tanh range; frames beyond each
    utterance's length are zero in z; m_p/logs_p masked."""
    from vispeech_amd.synth import synth_batch
    batch = synth_batch(2, seed=44, mean_phonemes=20, std_phonemes=4, min_phonemes=10, max_phonemes=30,
                        mean_frames=200, jitter_frames=40)
    dev = net.device
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    o, x_mask, (z, z_p, m_p, logs_p), *_ = net.infer(
        t(batch["phonemes"]), t(batch["lengths"]), sid=t(batch["sid"]), noise_scale=0.667,
        duration_control=t(batch["duration"]), pitch_control=t(batch["f0"]), energy_control=t(batch["energy"]),
        noise=t(batch["noise"]))
    assert torch.isfinite(o).all() and o.abs().max() <= 1.0
    m = x_mask.to(torch.float32)
    assert (z * (1 - m)).abs().max() == 0
    assert (m_p * (1 - m)).abs().max() == 0 and (logs_p * (1 - m)).abs().max() == 0
    assert x_mask.sum(dim=(1, 2)).cpu().tolist() == batch["frame_lengths"].tolist()
    assert o.shape[-1] == 512 * int(batch["frame_lengths"].max())


def test_f32_generator_mode_matches_golden(dims, weights, golden_dir, monkeypatch):
    """The f32-MFMA channel-major generator (VSP_GENERATOR=f32) stays available as a second,
    independently written implementation of Generator.forward; both must match the reference."""
    monkeypatch.setenv("VSP_GENERATOR", "f32")
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    m = SynthesizerTrn(*args, **kwargs).eval()
    m.load_state_dict(weights, strict=True)
    for case in ("ragged_controls", "maxlen_dur3d"):
        g = golden(golden_dir, case)
        o = run_case(m, g)[0]
        e = rel_err(to_np(o), g["o"])
        print("f32 generator", case, f"{e:.2e}")
        assert e <= WAVE_TOL


def test_long_utterance_matches_oracle(net, oracle):
    """One long utterance (many time tiles per kernel, attention over ~1.2k frames) against the
    CPU oracle -- the long-form configuration (BASELINE C5) at a size the oracle finishes in seconds."""
    from vispeech_amd.synth import synth_batch
    batch = synth_batch(1, seed=105, fixed_phonemes=110, fixed_frames=1200)
    ref, out = _oracle_vs_hip(net, oracle, batch)
    for name in ("m_p", "z"):
        e = rel_err(to_np(out[name]), ref[name].numpy())
        print("long", name, f"{e:.2e}")
        assert e <= STAGE_TOL, (name, e)
    e = rel_err(to_np(out["o"]), ref["o"].numpy())
    print("long o", f"{e:.2e}")
    assert e <= WAVE_TOL


def test_long_form_60s_properties(net):
    """BASELINE C5 shape (470 phonemes, 5168 frames = 60 s): runs in one call, output finite and
    bounded, and the first 1000 frames' audio is independent of what follows beyond the receptive
    field (generator +-13 frames, flow +-32, attention is global so z is compared after masking the
    tail with an identical prefix batch padded to the same frame count)."""
    from vispeech_amd.synth import workload
    b = workload("C5")
    dev = net.device
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    o, x_mask, (z, *_), *_ = net.infer(t(b["phonemes"]), t(b["lengths"]), sid=t(b["sid"]), noise_scale=0.667,
                                       duration_control=t(b["duration"]), pitch_control=t(b["f0"]),
                                       energy_control=t(b["energy"]), noise=t(b["noise"]))
    assert o.shape == (1, 1, 512 * 5168) and torch.isfinite(o).all() and o.abs().max() <= 1.0
    assert int(x_mask.sum()) == 5168
    # vocoder locality: re-run the generator on the first 1100 frames of z; samples of the first
    # 1000 frames (beyond the 13-frame receptive field from the cut) must be identical
    g = net._engine.encode(t(b["phonemes"]), t(b["lengths"]), t(b["sid"]), t(b["duration"]), t(b["f0"]), t(b["energy"]))["g"]
    o_cut = net._engine.generator(z[:, :, :1100].contiguous(), g)
    n = 512 * 1000
    assert torch.equal(o[:, :, :n], o_cut[:, :, :n])


def test_streamed_vocoder_is_bit_identical(net, weights):
    """Chunked generator with the 13-frame halo == one full call (streamed long-form synthesis)."""
    rng = np.random.Generator(np.random.PCG64(9))
    z = rng.standard_normal((1, 192, 700), dtype=np.float32)
    gvec = weights["emb_g.weight"][[3]]
    full = net._engine.generator(z, gvec)
    parts = list(net._engine.generator_stream(z, gvec, chunk_frames=192))
    assert len(parts) == 4 and parts[0].shape[-1] == 192 * 512
    assert torch.equal(torch.cat(parts, dim=-1), full)


@pytest.mark.parametrize("frames", [1, 2, 13, 31, 33, 64, 127, 257])
def test_generator_edge_lengths_match_oracle(net, oracle, weights, dims, frames):
    """Tile-edge and degenerate lengths of the vocoder (shorter than the receptive field, one frame,
    one past a tile boundary) against the CPU oracle's Generator.forward."""
    from oracle.vispeech_oracle import generator as oracle_generator
    rng = np.random.Generator(np.random.PCG64(1000 + frames))
    z = rng.standard_normal((2, 192, frames), dtype=np.float32)
    sid = np.array([5, 41])
    gvec = weights["emb_g.weight"][sid]
    o = net._engine.generator(z, gvec)
    ref = oracle_generator(oracle.w, torch.from_numpy(z), torch.from_numpy(gvec)[:, :, None], dims)
    assert o.shape == ref.shape == (2, 1, 512 * frames)
    e = rel_err(to_np(o), ref.numpy())
    assert e <= WAVE_TOL, (frames, e)


def test_degenerate_batches(net, oracle):
    """Single phoneme, zero-length phonemes in the middle, an utterance that is all padding but one
    frame, and max_len larger than the batch: outputs match the oracle."""
    dev = net.device
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    ph = np.array([[7, 0, 0, 0], [9, 15, 3, 200]], dtype=np.int64)
    ln = np.array([1, 4], dtype=np.int64)
    sid = np.array([0, 66], dtype=np.int64)
    dur = np.array([[1, 0, 0, 0], [3, 0, 5, 2]], dtype=np.float32)
    f0 = np.array([[220, 0, 0, 0], [0, 180, 300, 150]], dtype=np.float32)
    en = np.array([[50, 0, 0, 0], [10, 20, 90, 60]], dtype=np.float32)
    noise = np.random.Generator(np.random.PCG64(5)).standard_normal((2, 192, 10), dtype=np.float32)
    ref = oracle.infer(ph, ln, sid, noise=noise, noise_scale=0.8, max_len=50, duration_control=dur, pitch_control=f0,
                       energy_control=en)
    o, x_mask, (z, *_), *_ = net.infer(t(ph), t(ln), sid=t(sid), noise_scale=0.8, max_len=50, duration_control=t(dur),
                                       pitch_control=t(f0), energy_control=t(en), noise=t(noise))
    np.testing.assert_array_equal(to_np(x_mask), ref["x_mask"].numpy())
    assert x_mask.sum(dim=(1, 2)).tolist() == [1, 10]
    assert rel_err(to_np(z), ref["z"].numpy()) <= STAGE_TOL
    assert rel_err(to_np(o), ref["o"].numpy()) <= WAVE_TOL


def test_abi_error_paths_on_device(net):
    """Workspace too small, missing noise and bad arguments are reported as error codes, never crashes."""
    from vispeech_amd import _lib
    eng = net._engine
    z = torch.zeros(1, 192, 8, device=net.device)
    g = torch.zeros(1, 256, device=net.device)
    o = torch.zeros(1, 1, 8 * 512, device=net.device)
    ws = torch.zeros(1024, dtype=torch.uint8, device=net.device)
    import ctypes as C
    P = lambda x: C.c_void_p(x.data_ptr())
    rc = eng.lib.vsp_generator(eng.ctx, None, 1, 8, P(z), P(g), P(o), P(ws), ws.numel())
    assert rc == -6 and b"workspace" in eng.lib.vsp_last_error(eng.ctx)
    rc = eng.lib.vsp_generator(eng.ctx, None, 0, 8, P(z), P(g), P(o), P(ws), ws.numel())
    assert rc == -1
    with pytest.raises(ValueError):
        net.infer(torch.zeros(1, 3, dtype=torch.int64), torch.tensor([3]), sid=None)
    enc = eng.encode(torch.ones(1, 3, dtype=torch.int64), torch.tensor([3]), torch.tensor([1]),
                     duration_ctl=torch.ones(1, 3), pitch_ctl=torch.ones(1, 3), energy_ctl=torch.ones(1, 3))
    with pytest.raises(ValueError, match="max_len"):
        eng.decode(enc, 3, None, 0.5, max_len=-1, noise_seed=1)   # the C side reads max_len < 0 as "no limit": refused here
    with pytest.raises(ValueError, match="noise_seed"):
        eng.decode(enc, 3, None, 0.5)                      # no noise and no seed: refused (no silent seed 0)
    # vsp_attention: caller-owned workspace (ABI 4), too small -> error code
    qkv = torch.zeros(1, 3 * 192, 16, device=net.device)
    ln = torch.tensor([16], device=net.device)
    out = torch.zeros(1, 192, 16, device=net.device)
    rc = eng.lib.vsp_attention(eng.ctx, None, 0, 0, 1, 16, P(qkv), P(ln), P(out), P(ws), 16)
    assert rc == -6
    assert eng.lib.vsp_attention_workspace_bytes(eng.ctx, 1, 16) > 16
    rc = eng.lib.vsp_wn_layer(eng.ctx, None, 99, 0, 1, 16, P(out), P(g), P(ln), P(qkv), 0, P(ws), ws.numel())
    assert rc == -1


def test_reduced_precision_mode_is_opt_in_and_gated(dims, weights, golden_dir, monkeypatch):
    """VSP_GENERATOR=f16 (plain f16 operands, one MFMA per product: the low-precision variant of
    BASELINE config 3) is NOT the product default; it must still reach >= 30 dB SNR against the
    reference waveform -- and it must NOT pass the fp32 gate by accident."""
    monkeypatch.setenv("VSP_GENERATOR", "f16")
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    m = SynthesizerTrn(*args, **kwargs).eval()
    m.load_state_dict(weights, strict=True)
    g = golden(golden_dir, "ragged_controls")
    o = to_np(run_case(m, g)[0]).astype(np.float64)
    ref = g["o"].astype(np.float64)
    snr = 10 * np.log10((ref ** 2).sum() / ((o - ref) ** 2).sum())
    e = rel_err(o, ref)
    # audio-domain metric of the reference's training loss (mel-L1 on log-mel, mel_processing.py:85-112, train.py:
    # 163-177), computed on the GPU (vsp_spectrogram + vsp_spec_to_mel)
    mel = lambda w: to_np(m._engine.mel_spectrogram(w[:, 0].astype(np.float32), 80, 44100)).astype(np.float64)
    mel_l1 = float(np.abs(mel(o) - mel(ref)).mean())
    print(f"f16 generator: SNR {snr:.1f} dB, max rel err {e:.2e}, log-mel L1 {mel_l1:.2e}")
    assert snr >= 30.0
    assert mel_l1 <= 1e-2
    assert e > 1e-5          # i.e. really the reduced-precision path


@pytest.mark.parametrize("env", [{"VSP_FUSE_PAIRS": "0"}, {"VSP_CHAIN": "0"}, {"VSP_CHAIN": "7"}, {"VSP_PAIR": "ring"},
                                 {"VSP_TIMG": "0"}, {"VSP_FUSE_PAIRS": "0", "VSP_TIMG": "0"}, {"VSP_PP": "0"},
                                 {"VSP_PP": "0", "VSP_TIMG": "0"}, {"VSP_CHAIN_RING": "1"}, {"VSP_RW64": "1"}],
                         ids=["two_launches", "pair_launches", "chains_for_every_kernel_size", "ring_pair_kernel",
                              "fp32_intermediates", "two_launches_fp32_intermediates", "no_128_channel_pair_kernel",
                              "no_128_channel_pair_kernel_fp32_intermediates", "ring_chain_kernel",
                              "register_weights_k3_pairs_64_channels"])
def test_fused_resblock_paths_are_bit_identical(net, dims, weights, monkeypatch, env):
    """The ResBlocks of the 32/64-channel stages run fused (gen16.hip): by default a whole ResBlock of the
    32-channel stage is ONE launch (g16_chain: the running x in registers, every intermediate in LDS, the chain's halo
    recomputed per tile) and a conv pair of the 64-channel stage is one launch (g16_pair).  VSP_CHAIN=0 runs pairs
    everywhere, VSP_FUSE_PAIRS=0 one g16_conv launch per convolution, VSP_CHAIN=7 chains for k3, k7 and k11.  Round 4:
    the k7 / k11 pairs of the 32-channel stage run on the persistent register-weights kernel (g16_rw; VSP_PAIR=ring = the
    LDS-ring pair kernel), and the per-convolution stages hand a pair's intermediate over as an operand image
    (VSP_TIMG=0 = as an fp32 tensor); the k3 / k7 pairs of the 128-channel stage are one launch on the ping-pong tile
    (g16_pp; VSP_PP=0 = two launches).  Every switch is read when a context is created.  Same split products, same
    accumulation order => identical bits, for tiles that start/end anywhere in the utterance."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    two = SynthesizerTrn(*args, **kwargs).eval()
    two.load_state_dict(weights, strict=True)
    r = np.random.Generator(np.random.PCG64(5))
    for B, T in ((3, 37), (2, 1), (1, 130), (5, 64)):
        z = torch.from_numpy(r.standard_normal((B, dims.inter_channels, T)).astype(np.float32))
        g = torch.from_numpy(r.standard_normal((B, dims.gin_channels)).astype(np.float32))
        a = net._engine.generator(z, g)
        b = two._engine.generator(z, g)
        assert torch.equal(a, b), (B, T, float((a - b).abs().max()))


def test_alternative_kernel_paths_match_golden(dims, weights, golden_dir, monkeypatch):
    """Switchable second implementations stay correct: frame-rate convs on the f32 matrix core
    (VSP_FRAME=f32) and the two-pass f32 attention kernel (VSP_ATT=f32)."""
    monkeypatch.setenv("VSP_FRAME", "f32")
    monkeypatch.setenv("VSP_ATT", "f32")
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    m = SynthesizerTrn(*args, **kwargs).eval()
    m.load_state_dict(weights, strict=True)
    for case in ("ragged_predictors", "c1_filelist"):
        g = golden(golden_dir, case)
        out = run_case(m, g)
        assert rel_err(to_np(out[0]), g["o"]) <= WAVE_TOL, case
        assert rel_err(to_np(out[2][2]), g["m_p"]) <= STAGE_TOL, case


@pytest.mark.parametrize("env", [{"VSP_COLS": "0"}, {"VSP_COLS_MIN_BLOCKS": "0"}, {"VSP_COLS_MIN_BLOCKS": "0", "VSP_COLS_BLOCKS": "100000"}])
def test_column_tile_kernels_and_row_tiled_kernels_both_match_golden(net, dims, weights, golden_dir, monkeypatch, env):
    """Round 6's column-tile kernels (conv_cols.hip: mid-size 1x1 convolutions; conv_o + LayerNorm and q | k | v +
    operand packing in one launch each) serve grids of 32 .. 256 column tiles by default (the per-rank slices of a sharded
    batch; the full-size tests of tests/test_baseline_configs.py run the two fused launches).  The goldens are a few
    column tiles, so here the selections are forced: VSP_COLS=0 (row-tiled kernels + separate pack / LayerNorm launches
    everywhere: round 5's path), VSP_COLS_MIN_BLOCKS=0 (column tiles from the first tile on), and with
    VSP_COLS_BLOCKS=100000 at every size -- each against the reference goldens, and against the default selection
    within the stage tolerance (another summation order, not the same bits)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    m = SynthesizerTrn(*args, **kwargs).eval()
    m.load_state_dict(weights, strict=True)
    for case in ("ragged_predictors", "ragged_controls", "evaluate_caller"):
        g = golden(golden_dir, case)
        out = run_case(m, g)
        ref = run_case(net, g)
        np.testing.assert_array_equal(to_np(out[3]).reshape(g["duration"].shape), g["duration"])
        assert rel_err(to_np(out[0]), g["o"]) <= WAVE_TOL, case
        for name, i in (("z", 0), ("z_p", 1), ("m_p", 2), ("logs_p", 3)):
            assert rel_err(to_np(out[2][i]), g[name]) <= STAGE_TOL, (case, name)
            assert rel_err(to_np(out[2][i]), to_np(ref[2][i])) <= STAGE_TOL, (case, name)


def test_single_call_infer_matches_three_call_path(net, golden_dir):
    """vsp_infer (one call, caller-supplied frame padding, no host synchronisation) == vsp_encode +
    vsp_frame_lengths_host + vsp_decode, bit for bit, and both match the reference golden."""
    g = golden(golden_dir, "ragged_controls")
    eng = net._engine
    tf = int(g["x_mask"].shape[2])
    r = eng.infer_padded(g["in_phonemes"], g["in_lengths"], g["in_sid"], tf, g["in_noise"],
                         noise_scale=float(g["in_noise_scale"]), duration_ctl=g["in_duration"],
                         pitch_ctl=g["in_f0"], energy_ctl=g["in_energy"])
    ref = run_case(net, g)
    assert torch.equal(r["o"], ref[0])
    assert torch.equal(r["z"], ref[2][0]) and torch.equal(r["m_p"], ref[2][2])
    np.testing.assert_array_equal(to_np(r["x_mask"]), g["x_mask"])
    assert rel_err(to_np(r["o"]), g["o"]) <= WAVE_TOL
    # a larger padding only appends frames: the common part of the latent is unchanged
    tf2 = tf + 7
    noise2 = np.zeros((g["in_noise"].shape[0], g["in_noise"].shape[1], tf2), dtype=np.float32)
    noise2[:, :, :tf] = g["in_noise"]
    r2 = eng.infer_padded(g["in_phonemes"], g["in_lengths"], g["in_sid"], tf2, noise2,
                          noise_scale=float(g["in_noise_scale"]), duration_ctl=g["in_duration"],
                          pitch_ctl=g["in_f0"], energy_ctl=g["in_energy"])
    np.testing.assert_array_equal(to_np(r2["frame_lengths"]), to_np(r["frame_lengths"]))
    assert rel_err(to_np(r2["m_p"][:, :, :tf]), g["m_p"]) <= STAGE_TOL


@pytest.mark.parametrize("T,lens", [(5, [5, 3]), (37, [37, 20]), (300, [300, 123]), (1500, [1500, 1])])
def test_attention_stage_matches_oracle(net, weights, dims, T, lens):
    """The MFMA attention kernel alone (vsp_attention) against the closed form of reference
    attentions.py:148-179 in the oracle: short sequences (T < window), ragged masks, and T = 1500 where the
    key range is split over the waves of a block."""
    from oracle.vispeech_oracle import rel_attention
    r = np.random.Generator(np.random.PCG64(T))
    B, H = len(lens), dims.hidden_channels
    qkv = r.standard_normal((B, 3 * H, T)).astype(np.float32)
    lens_a = np.array(lens, dtype=np.int64)
    for which, layer, prefix in ((0, 1, "enc_p.encoder"), (2, 3, "frame_prior_net.fft_block")):
        ek = torch.from_numpy(weights[f"{prefix}.attn_layers.{layer}.emb_rel_k"])
        ev = torch.from_numpy(weights[f"{prefix}.attn_layers.{layer}.emb_rel_v"])
        t = torch.from_numpy(qkv)
        mask = (torch.arange(T)[None, :] < torch.from_numpy(lens_a)[:, None])
        ref = rel_attention(t[:, :H], t[:, H:2 * H], t[:, 2 * H:], ek, ev, mask, dims.n_heads, dims.window_size)
        out = net._engine.attention(which, layer, qkv, lens_a)
        # rows of valid queries (masked query rows are finite garbage by design: reference gotcha G9,
        # zeroed by the encoder's final x * mask)
        for b, n in enumerate(lens):
            assert rel_err(to_np(out)[b, :, :n], ref.numpy()[b, :, :n]) <= STAGE_TOL, (T, which, b)


@pytest.mark.parametrize("which,T,lens", [(1, 37, [37, 20, 1]), (3, 130, [130, 64])])
def test_wn_layer_matches_oracle(net, oracle, dims, which, T, lens):
    """vsp_wn_layer (SURVEY 8b's per-stage entry; reference modules.py:148-176): the four layers of a coupling
    layer's WN one call at a time, running x and the skip sum compared with the oracle after EVERY layer (ragged
    masks, odd T)."""
    import torch.nn.functional as F
    from oracle.vispeech_oracle import wn_layer
    r = np.random.Generator(np.random.PCG64(100 * which + T))
    B, h, nl = len(lens), dims.hidden_channels, dims.flow_layers
    x0 = r.standard_normal((B, h, T)).astype(np.float32)
    g = r.standard_normal((B, dims.gin_channels)).astype(np.float32)
    lens_a = np.array(lens, dtype=np.int64)
    mask = (torch.arange(T)[None, :] < torch.from_numpy(lens_a)[:, None]).to(torch.float32)[:, None, :]
    w, prefix = oracle.w, f"flow.flows.{2 * which}.enc"
    gc = F.conv1d(torch.from_numpy(g)[:, :, None], w[f"{prefix}.cond_layer.weight"], w[f"{prefix}.cond_layer.bias"])
    xr = torch.from_numpy(x0) * mask
    outr = torch.zeros_like(xr)
    xg, skg = (torch.from_numpy(x0) * mask).to(net.device), None
    for i in range(nl):
        xr, outr = wn_layer(w, prefix, i, xr, outr, mask, gc, h, nl, dims.flow_kernel)
        xg, skg = net._engine.wn_layer(which, i, xg, g, lens_a, skg)
        expect_skip = outr * mask if i == nl - 1 else outr          # the last layer folds WN's `output * x_mask` in
        assert rel_err(to_np(xg), xr.numpy()) <= STAGE_TOL, (which, i)
        assert rel_err(to_np(skg), expect_skip.numpy()) <= STAGE_TOL, (which, i)


def test_library_noise_draw_is_the_documented_philox_stream(net, oracle, dims):
    """noise == NULL: the library draws the reparameterisation noise itself (vsp_randn: Philox4x32-10 + Box-Muller,
    include/vispeech_hip.h) -- the C caller's torch.randn_like (reference models.py:718).  Checked against the numpy
    restatement in oracle/, for moments, for determinism, and end to end: infer(noise_seed=s) == infer(noise=randn(s))."""
    from oracle.vispeech_oracle import philox_randn
    eng = net._engine
    n = 3 * 192 * 77 + 1                                   # not a multiple of four
    got = to_np(eng.randn(0x1234567890ABCDEF, n))
    ref = philox_randn(0x1234567890ABCDEF, n)
    assert np.abs(got - ref).max() <= 2e-5                 # same words; logf / sincosf differ in the last ulps
    big = to_np(eng.randn(7, 1 << 22))
    assert abs(big.mean()) < 3e-3 and abs(big.std() - 1.0) < 3e-3 and np.isfinite(big).all()
    assert abs(float(np.mean(big ** 3))) < 1e-2 and abs(float(np.mean(big ** 4)) - 3.0) < 3e-2
    assert not np.array_equal(got[:64], to_np(eng.randn(8, 64)))
    from vispeech_amd.synth import synth_batch
    b = synth_batch(2, seed=11, mean_phonemes=8, std_phonemes=1, min_phonemes=6, max_phonemes=10, mean_frames=30, jitter_frames=5)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(net.device)
    kw = dict(sid=t(b["sid"]), noise_scale=0.667, duration_control=t(b["duration"]), pitch_control=t(b["f0"]),
              energy_control=t(b["energy"]))
    a = net.infer(t(b["phonemes"]), t(b["lengths"]), noise_seed=42, **kw)
    tf = a[2][0].shape[2]
    drawn = eng.randn(42, 2, dims.inter_channels, tf)
    c = net.infer(t(b["phonemes"]), t(b["lengths"]), noise=drawn, **kw)
    assert torch.equal(a[0], c[0]) and torch.equal(a[2][1], c[2][1])
    d = net.infer(t(b["phonemes"]), t(b["lengths"]), noise_seed=43, **kw)
    assert not torch.equal(a[2][1], d[2][1])
    ref = oracle.infer(b["phonemes"], b["lengths"], b["sid"], noise=to_np(drawn), noise_scale=0.667,
                       duration_control=b["duration"], pitch_control=b["f0"], energy_control=b["energy"])
    assert rel_err(to_np(a[0]), ref["o"].numpy()) <= WAVE_TOL


def test_library_noise_of_a_shard_does_not_depend_on_the_shard_layout(net, dims):
    """a rank that synthesises utterances [lo, hi) of a global batch with library-drawn noise must draw ITS
    part of the global [B, C, Tf] stream (vsp_randn_at / vsp_set_noise_offset), at any (unaligned) element offset; and a
    Python caller of the one-call path has to name the seed (no silent seed 0)."""
    from oracle.vispeech_oracle import philox_randn
    eng = net._engine
    whole = eng.randn(5, 4099)
    for first, n in ((0, 17), (1, 64), (6, 1021), (4096, 3)):
        part = eng.randn(5, n, first=first)
        assert torch.equal(part, whole[first:first + n]), (first, n)
    np.testing.assert_array_equal(philox_randn(5, 1021, first=6), philox_randn(5, 1027)[6:])
    from vispeech_amd.synth import synth_batch
    b = synth_batch(3, seed=12, mean_phonemes=8, std_phonemes=1, min_phonemes=6, max_phonemes=10, mean_frames=30, jitter_frames=5)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(net.device)
    ctl = dict(duration_control=t(b["duration"]), pitch_control=t(b["f0"]), energy_control=t(b["energy"]))
    full = net.infer(t(b["phonemes"]), t(b["lengths"]), sid=t(b["sid"]), noise_scale=0.667, noise_seed=77, **ctl)
    tf = full[2][0].shape[2]
    lo = 0
    for hi in (1, 3):                                      # shards [0, 1) and [1, 3) under the global padding
        sl = slice(lo, hi)
        part = net.infer(t(b["phonemes"])[sl], t(b["lengths"])[sl], sid=t(b["sid"])[sl], noise_scale=0.667, noise_seed=77,
                         t_f=tf, noise_offset=lo * dims.inter_channels * tf, **{k: v[sl] for k, v in ctl.items()})
        assert torch.equal(part[2][1], full[2][1][sl]) and torch.equal(part[0], full[0][sl])
        lo = hi
    with pytest.raises(ValueError):
        eng.infer_padded(t(b["phonemes"]), t(b["lengths"]), t(b["sid"]), tf, None, noise_scale=0.667,
                         duration_ctl=ctl["duration_control"], pitch_ctl=ctl["pitch_control"], energy_ctl=ctl["energy_control"])


def test_attention_operands_beyond_the_f16_range_stay_finite(net, dims):
    """the split-f16 attention packs q / k / v with fixed scales; magnitudes beyond the f16 range are clamped
    (documented in include/vispeech_hip.h) instead of turning rows into NaN."""
    T, H = 40, dims.hidden_channels
    qkv = np.random.Generator(np.random.PCG64(9)).standard_normal((1, 3 * H, T)).astype(np.float32)
    qkv[0, 5, 7] = 1e6
    qkv[0, H + 9, 3] = -1e7
    qkv[0, 2 * H + 1, 11] = 3e5
    out = to_np(net._engine.attention(0, 0, qkv, np.array([T])))
    assert np.isfinite(out).all()


def test_weights_beyond_the_packed_f16_range_are_refused(weights):
    """The split-f16 kernels take their weights * 2^8 as f16 pairs (kernels.h G16_WSCALE): a folded weight of |w| >= ~254
    cannot be represented and must fail the load loudly instead of turning into inf inside the matrix core."""
    from vispeech_amd import config as vcfg
    from vispeech_amd._lib import VspError
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    m = SynthesizerTrn(*args, **kwargs).eval()
    bad = dict(weights)
    key = "dec.resblocks.0.convs1.0.weight_g"
    bad[key] = bad[key] * 1.0e5
    with pytest.raises(VspError, match="exceeds what the split-f16 matrix path represents"):
        m.load_state_dict(bad, strict=True)
