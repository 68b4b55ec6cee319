"""Voice conversion through the C-ABI: SynthesizerTrn.voice_conversion
(reference models.py:724-732) = PosteriorEncoder (models.py:212-241) + forward flow with the source
speaker + reverse flow with the target speaker + generator.  Golden vectors come from the real
reference (tests/golden/make_golden.py vc); larger seeded cases are checked against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STAGE_TOL = 1e-5
WAVE_TOL = 1e-4


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
    return synth_state_dict(dims, seed=1234)            # full checkpoint, enc_q.* included


@pytest.fixture(scope="module")
def net(weights):
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


def test_voice_conversion_matches_reference_golden(net, golden_dir):
    g = np.load(os.path.join(golden_dir, "voice_conversion.npz"))
    dev = net.device
    t = lambda x: torch.from_numpy(x).to(dev)
    o_hat, y_mask, (z, z_p, z_hat) = net.voice_conversion(t(g["in_y"]), t(g["in_lengths"]), t(g["in_sid_src"]),
                                                          t(g["in_sid_tgt"]), noise=t(g["in_noise"]))
    assert y_mask.dtype == torch.float32                  # models.py:234: mask cast to x.dtype
    np.testing.assert_array_equal(to_np(y_mask), g["y_mask"])
    for name, v in (("z", z), ("z_p", z_p), ("z_hat", z_hat)):
        assert rel_err(to_np(v), g[name]) <= STAGE_TOL, name
    assert rel_err(to_np(o_hat), g["o_hat"]) <= WAVE_TOL


def test_posterior_encoder_and_flow_forward_stages(net, weights, golden_dir):
    g = np.load(os.path.join(golden_dir, "voice_conversion.npz"))
    eng = net._engine
    gsrc = torch.from_numpy(weights["emb_g.weight"][g["in_sid_src"]])
    z, m, logs = eng.posterior_encoder(g["in_y"], g["in_lengths"], gsrc, g["in_noise"])
    assert rel_err(to_np(m), g["m_q"]) <= STAGE_TOL
    assert rel_err(to_np(logs), g["logs_q"]) <= STAGE_TOL
    assert rel_err(to_np(z), g["z"]) <= STAGE_TOL
    z_p = eng.flow_forward(g["z"], gsrc, g["in_lengths"])
    assert rel_err(to_np(z_p), g["z_p"]) <= STAGE_TOL
    # forward then reverse with the same speaker is the identity on the valid frames
    back = eng.flow_reverse(z_p, gsrc, g["in_lengths"])
    assert rel_err(to_np(back), g["z"]) <= 1e-5


def test_voice_conversion_matches_oracle_larger_batch(net, oracle, dims):
    r = np.random.Generator(np.random.PCG64(77))
    lens = np.array([150, 97, 1, 64, 130], dtype=np.int64)
    B, T = len(lens), int(lens.max())
    y = np.abs(r.standard_normal((B, dims.spec_channels, T))).astype(np.float32)
    for b, n in enumerate(lens):
        y[b, :, n:] = 0.0
    src = r.integers(0, 67, B)
    tgt = r.integers(0, 67, B)
    noise = r.standard_normal((B, dims.inter_channels, T)).astype(np.float32)
    ref = oracle.voice_conversion(y, lens, src, tgt, noise)
    dev = net.device
    t = lambda x: torch.from_numpy(np.asarray(x)).to(dev)
    o_hat, y_mask, (z, z_p, z_hat) = net.voice_conversion(t(y), t(lens), t(src), t(tgt), noise=t(noise))
    for name, v in (("z", z), ("z_p", z_p), ("z_hat", z_hat)):
        assert rel_err(to_np(v), ref[name].numpy()) <= STAGE_TOL, name
    assert rel_err(to_np(o_hat), ref["o_hat"].numpy()) <= WAVE_TOL
    # padded frames of every latent are exactly zero (every stage multiplies by y_mask)
    for b, n in enumerate(lens):
        assert float(z_hat[b, :, n:].abs().max().item() if n < T else 0.0) == 0.0


def test_voice_conversion_needs_posterior_weights(dims):
    """A checkpoint without enc_q.* still loads (infer works) but voice_conversion fails loudly."""
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    from vispeech_amd.synth import synth_state_dict
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    m = SynthesizerTrn(*args, **kwargs).eval()
    m.load_state_dict(synth_state_dict(dims, seed=1234, infer_only=True), strict=True)
    assert not m._engine.has_voice_conversion
    y = torch.zeros(1, dims.spec_channels, 8, device=m.device)
    with pytest.raises(RuntimeError, match="enc_q"):
        m.voice_conversion(y, torch.tensor([8]), torch.tensor([0]), torch.tensor([1]))
    # and the C entry point itself refuses (VSP_ERR_STATE), not only the Python shim
    from vispeech_amd._lib import VspError
    with pytest.raises(VspError, match="STATE"):
        m._engine.voice_conversion(y, torch.tensor([8]), torch.tensor([0]), torch.tensor([1]),
                                   torch.zeros(1, dims.inter_channels, 8))


def test_spectrogram_front_end_matches_oracle(net, dims):
    """vsp_spectrogram (DFT as a split-f16 MFMA conv over reflect-padded frames) against torch.stft in the
    oracle: the linear spectrogram that voice_conversion consumes (reference mel_processing.py:50-69)."""
    from oracle.vispeech_oracle import spectrogram
    r = np.random.Generator(np.random.PCG64(3))
    n_fft, hop = 2 * (dims.spec_channels - 1), dims.hop_length
    for B, L in ((2, 512 * 37), (1, 4097), (3, 1000)):
        t = np.arange(L) / dims.sampling_rate
        audio = (0.4 * np.sin(2 * np.pi * 220.0 * t)[None, :] * r.uniform(0.2, 1.0, (B, 1)) +
                 0.1 * r.standard_normal((B, L))).astype(np.float32).clip(-1, 1)
        ref = spectrogram(audio, n_fft, hop).numpy()
        got = to_np(net._engine.spectrogram(audio))
        assert got.shape == ref.shape == (B, dims.spec_channels, 1 + (L + 2 * ((n_fft - hop) // 2) - n_fft) // hop)
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), (B, L, np.abs(got - ref).max(), np.abs(ref).max())


def test_wav_to_wav_voice_conversion(net, oracle, dims):
    """Audio in, audio out: spectrogram front end + voice_conversion, against the oracle's torch.stft + VC."""
    from oracle.vispeech_oracle import spectrogram
    r = np.random.Generator(np.random.PCG64(9))
    L = 512 * 24
    audio = (0.3 * r.standard_normal((2, L))).astype(np.float32).clip(-1, 1)
    n_fft = 2 * (dims.spec_channels - 1)
    y_ref = spectrogram(audio, n_fft, dims.hop_length)
    lens = np.array([24, 17], dtype=np.int64)
    y_ref = y_ref * (torch.arange(24)[None, None, :] < torch.from_numpy(lens)[:, None, None])
    noise = r.standard_normal((2, dims.inter_channels, 24)).astype(np.float32)
    src, tgt = np.array([5, 6]), np.array([20, 6])
    ref = oracle.voice_conversion(y_ref.numpy(), lens, src, tgt, noise)
    y = net._engine.spectrogram(audio)
    y = y * (torch.arange(24, device=y.device)[None, None, :] < torch.from_numpy(lens).to(y.device)[:, None, None])
    dev = net.device
    t = lambda x: torch.from_numpy(np.asarray(x)).to(dev)
    o_hat, _, (z, z_p, z_hat) = net.voice_conversion(y, t(lens), t(src), t(tgt), noise=t(noise))
    assert rel_err(to_np(z), ref["z"].numpy()) <= 5e-5            # spectrogram error rides on top of the stage error
    assert rel_err(to_np(o_hat), ref["o_hat"].numpy()) <= 2e-4
