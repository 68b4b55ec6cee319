"""Trimmed tails (round 5): behind an utterance's last frame the generator's input is exactly zero (reference
models.py:720: `z * x_mask`; `dec` itself applies no mask, models.py:271-290), so the library computes every utterance
only to `length + 13 + 1 + 13` frames (output frame F depends on input frames F - 13 .. F + 13, sample-exact) and fills the rest of the padded waveform -- which the reference still returns --
from the steady state and the computed tensor end (vispeech_amd/csrc/kernels.h, launch_gen_tail_fill).  The FULL padded
output must equal the to-the-padded-length run (VSP_TRIM_TAILS=0, second implementation) bit for bit, whichever kernels
the other switches select, and the oracle within the usual tolerance.  Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HALO = 14          # vsp_generator_halo_frames for configs/config.json (asserted below)


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


def make_net(weights):
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    m = SynthesizerTrn(*args, **kwargs).eval()
    m.load_state_dict(weights, strict=True)
    return m


@pytest.fixture(scope="module")
def net(weights):
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return make_net(weights)


@pytest.fixture(scope="module")
def net_untrimmed(weights):
    """A context created under VSP_TRIM_TAILS=0 (the switch is read by vsp_create): every utterance to the padded length."""
    mp = pytest.MonkeyPatch()
    mp.setenv("VSP_TRIM_TAILS", "0")
    try:
        return make_net(weights)
    finally:
        mp.undo()


def batch_with_frames(frames, seed):
    """A synthetic batch whose utterances have exactly `frames` frames (durations rewritten, noise redrawn)."""
    from vispeech_amd.synth import synth_batch
    frames = np.asarray(frames, dtype=np.int64)
    b = synth_batch(len(frames), seed=seed, mean_phonemes=14, std_phonemes=4, min_phonemes=6, max_phonemes=24,
                    mean_frames=40, jitter_frames=5)
    r = np.random.Generator(np.random.PCG64(seed + 1))
    for i, L in enumerate(frames):
        n = int(b["lengths"][i])
        d = np.zeros(n)
        cut = np.sort(r.integers(0, L + 1, size=n - 1))
        d[:] = np.diff(np.concatenate([[0], cut, [L]]))
        b["duration"][i, :] = 0
        b["duration"][i, :n] = d
    b["frame_lengths"] = frames
    b["noise"] = r.standard_normal((len(frames), 192, int(frames.max())), dtype=np.float32)
    return b


def run(net, b, sl=slice(None), t_f=None, **kw):
    t = lambda a: torch.from_numpy(np.asarray(a)).to(net.device)
    if t_f is not None and t_f > b["noise"].shape[2]:          # a global padding beyond the local maximum: noise to match
        extra = np.random.Generator(np.random.PCG64(99)).standard_normal(
            (b["noise"].shape[0], b["noise"].shape[1], t_f - b["noise"].shape[2]), dtype=np.float32)
        b = dict(b, noise=np.concatenate([b["noise"], extra], axis=2))
    return net.infer(t(b["phonemes"][sl]), t(b["lengths"][sl]), sid=t(b["sid"][sl]), noise_scale=0.667,
                     duration_control=t(b["duration"][sl]), pitch_control=t(b["f0"][sl]), energy_control=t(b["energy"][sl]),
                     noise=t(b["noise"][sl]), t_f=t_f, **kw)


# padded length 150; an utterance is computed to length + 13 + 1 + 13 frames where that is shorter: untrimmed (150, 124,
# 123: length + 27 >= T), the shortest trimmed tails (122: the steady-state frame is the LAST one before the tensor end's
# 13 frames, nothing to fill between; 121: one frame to fill), ordinary, short, and tiny utterances
FRAMES = [150, 124, 123, 122, 121, 120, 96, 60, 31, 7, 1]


def test_halo_constant(net):
    assert net._engine.lib.vsp_generator_halo_frames(net._engine.ctx) == HALO


def test_trimmed_tails_equal_the_padded_run_bit_for_bit(net, net_untrimmed):
    b = batch_with_frames(FRAMES, seed=501)
    o1, m1, (z1, *_), *_ = run(net, b)
    o0, m0, (z0, *_), *_ = run(net_untrimmed, b)
    assert o1.shape == (len(FRAMES), 1, 512 * 150)
    np.testing.assert_array_equal(to_np(z1), to_np(z0))
    a, c = to_np(o1), to_np(o0)
    for i, L in enumerate(FRAMES):       # per utterance, so that a failure names the case
        np.testing.assert_array_equal(a[i], c[i], err_msg=f"utterance {i} ({L} of 150 frames)")
    # the padded tails are NOT zero (gotcha G5: `dec` is unmasked) -- the fill is doing real work
    assert np.abs(a[6, 0, 512 * (60 + HALO + 2):512 * (150 - HALO - 2)]).max() > 0


def test_trimmed_tail_structure(net_untrimmed):
    """What the fill relies on, observed on the UNTRIMMED output (so that a wrong halo constant cannot hide): once HALO
    frames behind an utterance's end and HALO frames before the tensor's end the waveform is periodic in one frame."""
    b = batch_with_frames([150, 60], seed=502)
    o = to_np(run(net_untrimmed, b)[0])[1, 0].reshape(150, 512)
    steady = o[60 + HALO]
    for f in range(60 + HALO, 150 - HALO):
        np.testing.assert_array_equal(o[f], steady, err_msg=f"frame {f}")
    assert not np.array_equal(o[150 - 1], steady)       # the tensor end differs from the steady state


@pytest.mark.parametrize("env", [
    {"VSP_CHAIN": "0"},                          # 32-channel k3 pairs on g16_rw instead of the g16_rc chain
    {"VSP_PAIR": "ring", "VSP_CHAIN_RING": "1"},   # the LDS-ring pair / chain kernels
    {"VSP_FUSE_PAIRS": "0"},                     # one g16_conv launch per convolution everywhere
    {"VSP_TIMG": "0", "VSP_PP": "0"},            # fp32 intermediates, two-launch pairs at 128 channels
    {"VSP_CHAIN": "7"},                          # whole-ResBlock launches for k3, k7 and k11
    {"VSP_RW64": "1"},                           # the 64-channel k3 pairs on the register-weights kernel (opt-in)
    {"VSP_RB_STREAMS": "15"},                    # the stages' ResBlock chains on side streams (opt-in; fork / join by events)
    {"VSP_RB_STREAMS": "5", "VSP_FUSE_PAIRS": "0"},   # the same on stages 0 and 2 only, one launch per convolution
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_trimmed_tails_under_every_kernel_selection(net_untrimmed, weights, monkeypatch, env):
    """Every generator kernel has the per-utterance extent: the second implementations, trimmed, equal the default
    path untrimmed."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    two = make_net(weights)
    b = batch_with_frames([90, 40, 61, 12], seed=503)
    np.testing.assert_array_equal(to_np(run(two, b)[0]), to_np(run(net_untrimmed, b)[0]))


def test_trimmed_tails_in_the_reduced_precision_mode(weights, monkeypatch):
    """VSP_GENERATOR=f16 (opt-in, one MFMA per product) runs the same kernels with TERMS = 1: trimmed == untrimmed there
    too."""
    monkeypatch.setenv("VSP_GENERATOR", "f16")
    a = make_net(weights)
    monkeypatch.setenv("VSP_TRIM_TAILS", "0")
    c = make_net(weights)
    b = batch_with_frames([90, 40, 61, 12], seed=506)
    np.testing.assert_array_equal(to_np(run(a, b)[0]), to_np(run(c, b)[0]))


def test_trimmed_tails_zero_length_and_single_utterance(net, net_untrimmed):
    """An utterance of zero frames (all durations zero) inside a batch is all padding: its whole waveform is the
    steady state + the tensor end; a batch of one has nothing to trim."""
    b = batch_with_frames([64, 0, 33], seed=507)
    np.testing.assert_array_equal(to_np(run(net, b)[0]), to_np(run(net_untrimmed, b)[0]))
    b1 = batch_with_frames([50], seed=508)
    np.testing.assert_array_equal(to_np(run(net, b1)[0]), to_np(run(net_untrimmed, b1)[0]))


@pytest.mark.parametrize("seed", [601, 602, 603, 604])
def test_trimmed_tails_random_batches(net, net_untrimmed, seed):
    """Random batch sizes and lengths (tile boundaries of every generator kernel land anywhere relative to the
    utterance ends): trimmed == untrimmed, bit for bit."""
    r = np.random.Generator(np.random.PCG64(seed))
    nb = int(r.integers(2, 12))
    tmax = int(r.integers(60, 260))
    frames = [int(x) for x in r.integers(1, tmax + 1, size=nb)]
    frames[int(r.integers(0, nb))] = tmax
    b = batch_with_frames(frames, seed=seed)
    o1, o0 = to_np(run(net, b)[0]), to_np(run(net_untrimmed, b)[0])
    for i, L in enumerate(frames):
        np.testing.assert_array_equal(o1[i], o0[i], err_msg=f"seed {seed}: utterance {i} ({L} of {tmax} frames)")


def test_trimmed_tails_with_max_len_and_global_padding(net, net_untrimmed):
    """max_len cuts the frame axis in front of the vocoder (models.py:720): lengths beyond the cut are untrimmed; a
    global padding t_f (sharded batches) beyond the local maximum makes EVERY utterance a trimmed one."""
    b = batch_with_frames([100, 45, 80], seed=504)
    for kw in (dict(max_len=70), dict(t_f=140), dict(t_f=140, max_len=120)):
        o1 = run(net, b, **kw)[0]
        o0 = run(net_untrimmed, b, **kw)[0]
        assert o1.shape == o0.shape
        np.testing.assert_array_equal(to_np(o1), to_np(o0), err_msg=str(kw))


def test_trimmed_tails_match_the_oracle(net, dims, weights):
    """... and the trimmed run's padded tail against the CPU oracle (which computes the padded batch densely)."""
    from oracle.vispeech_oracle import Oracle
    b = batch_with_frames([80, 30], seed=505)
    o = to_np(run(net, b)[0])
    ref = Oracle(weights, dims).infer(b["phonemes"], b["lengths"], b["sid"], noise=b["noise"], noise_scale=0.667,
                                      duration_control=b["duration"], pitch_control=b["f0"], energy_control=b["energy"])
    r = ref["o"].numpy()
    assert np.abs(o - r).max() <= 1e-4 * np.abs(r).max()
    tail = slice(512 * 30, None)
    assert np.abs(o[1, 0, tail] - r[1, 0, tail]).max() <= 1e-4 * np.abs(r).max()


def test_c3_batch_trimmed_equals_untrimmed(net, net_untrimmed):
    """The headline batch (64 mixed zh/ja utterances, 489 padded frames, 12.6 % padding): the full padded waveform."""
    from vispeech_amd.synth import workload
    b = workload("C3")
    o1 = run(net, b)[0]
    o0 = run(net_untrimmed, b)[0]
    assert o1.shape == (64, 1, 512 * int(b["frame_lengths"].max()))
    assert torch.equal(o1, o0), float((o1 - o0).abs().max())


def test_early_frame_count_copy_changes_nothing_but_the_wait(net, weights, monkeypatch):
    """With given durations vsp_encode derives the frame counts first and starts their copy to the host;
    vsp_frame_lengths_host then waits for that copy only (include/vispeech_hip.h).  Same counts, same tensors as the
    copy-and-synchronise form (VSP_EARLY_FL=0), also when the same context alternates between given and predicted durations."""
    monkeypatch.setenv("VSP_EARLY_FL", "0")
    plain = make_net(weights)
    b = batch_with_frames([70, 33, 51], seed=504)
    for _ in range(2):
        o1, m1, (z1, *_), d1, *_ = run(net, b)
        o0, m0, (z0, *_), d0, *_ = run(plain, b)
        assert torch.equal(o1, o0) and torch.equal(m1, m0) and torch.equal(z1, z0) and torch.equal(d1, d0)
        # predicted durations on the same context: the early copy must not be consumed by a call it does not belong to
        t = lambda a: torch.from_numpy(np.asarray(a)).to(net.device)
        kw = dict(sid=t(b["sid"]), noise_scale=0.0, pitch_control=t(b["f0"]), energy_control=t(b["energy"]))
        p1 = net.infer(t(b["phonemes"]), t(b["lengths"]), **kw)
        p0 = plain.infer(t(b["phonemes"]), t(b["lengths"]), **kw)
        assert torch.equal(p1[3], p0[3]) and p1[0].shape == p0[0].shape and torch.equal(p1[0], p0[0])
