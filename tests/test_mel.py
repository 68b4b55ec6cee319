"""Mel front end (reference mel_processing.py:73-112): the library's mel basis (host) and the GPU mel projection /
mel spectrogram against the oracle's restatement.  librosa -- where the reference takes the basis from
(mel_processing.py:79, 99; requirements.txt without a version) -- is not in this image: both sides restate its
published Slaney algorithm; the pin (round 5) is tests/test_oracle_golden.py: both agree to fp32 rounding with
transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney"), a third party's implementation adapted from
librosa.  The known-answer below is the one value librosa's own documentation prints for its default example."""
import numpy as np
import pytest
import torch

from oracle.vispeech_oracle import mel_filterbank as oracle_filterbank
from oracle.vispeech_oracle import mel_spectrogram as oracle_mel
from oracle.vispeech_oracle import spectrogram as oracle_spec


@pytest.fixture(scope="module")
def dims():
    from vispeech_amd.schema import ModelDims
    return ModelDims()


@pytest.fixture(scope="module")
def net(dims):
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    from vispeech_amd.synth import synth_state_dict
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    m = SynthesizerTrn(*args, **kwargs).eval()
    m.load_state_dict(synth_state_dict(dims, seed=1234, infer_only=True), strict=True)
    return m


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(44100, 2048, 80, 0.0, None), (22050, 1024, 80, 0.0, 8000.0),
                                                       (44100, 2048, 128, 30.0, 16000.0), (16000, 512, 40, 0.0, None)])
def test_mel_basis_matches_oracle_and_is_slaney_normalised(sr, n_fft, n_mels, fmin, fmax):
    from vispeech_amd.mel_processing import mel_filterbank
    w = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    ref = oracle_filterbank(sr, n_fft, n_mels, fmin, fmax)
    assert w.shape == ref.shape == (n_mels, n_fft // 2 + 1)
    assert np.abs(w - ref).max() <= 1e-7 * np.abs(ref).max()
    assert (w >= 0).all()
    # triangles: one run of non-zero bins per filter, peaks move up in frequency
    peaks = w.argmax(axis=1)
    assert (np.diff(peaks) >= 0).all()
    for m in range(n_mels):
        nz = np.flatnonzero(w[m])
        assert len(nz) == 0 or (np.diff(nz) == 1).all()


def test_mel_basis_known_answer_and_bad_arguments():
    from vispeech_amd.mel_processing import mel_filterbank
    w = mel_filterbank(22050, 2048, 128)          # librosa.filters.mel(sr=22050, n_fft=2048): its documented example
    assert round(float(w[0, 1]), 3) == 0.016 and w[0, 0] == 0.0 and w[-1, -1] == 0.0
    with pytest.raises(ValueError):
        mel_filterbank(22050, 2047, 80)
    with pytest.raises(ValueError):
        mel_filterbank(22050, 2048, 80, 9000.0, 8000.0)


@pytest.mark.gpu
def test_gpu_mel_matches_oracle(net, dims):
    """spec_to_mel on an arbitrary spectrogram and the full audio -> log-mel path (the reference's training-loss
    metric, train.py:163-177) against the oracle: mel before the log to 1e-5 of its maximum, log-mel to 1e-4 absolute."""
    from vispeech_amd import mel_processing as mp
    eng = net._engine
    r = np.random.Generator(np.random.PCG64(11))
    n_fft, hop, sr = 2 * (dims.spec_channels - 1), dims.hop_length, dims.sampling_rate
    audio = (0.3 * r.standard_normal((3, 7 * hop + 100))).astype(np.float32)
    spec = oracle_spec(audio, n_fft, hop)
    mel = mp.spec_to_mel_torch(spec, n_fft, 80, sr, 0.0, None, engine=eng)
    basis = torch.from_numpy(oracle_filterbank(sr, n_fft, 80))
    lin_ref = (basis @ spec).numpy()
    lin = np.exp(mel.cpu().numpy().astype(np.float64))
    clipped = np.maximum(lin_ref, 1e-5)
    assert np.abs(lin - clipped).max() <= 1e-5 * clipped.max()
    ref = oracle_mel(audio, sr, n_fft, hop, 80).numpy()
    out = mp.mel_spectrogram_torch(audio, n_fft, 80, sr, hop, n_fft, 0.0, None, engine=eng).cpu().numpy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 1e-4
    # a silent signal sits on the clamp: log(1e-5) everywhere is not produced by accident (sqrt(1e-6) spectrum floor)
    z = mp.mel_spectrogram_torch(np.zeros((1, 4 * hop), np.float32), n_fft, 80, sr, hop, n_fft, 0.0, None, engine=eng)
    zr = oracle_mel(np.zeros((1, 4 * hop), np.float32), sr, n_fft, hop, 80)
    assert np.abs(z.cpu().numpy() - zr.numpy()).max() <= 1e-4
    with pytest.raises(ValueError):
        mp.mel_spectrogram_torch(audio, n_fft // 2, 80, sr, hop, n_fft // 2, 0.0, None, engine=eng)
    with pytest.raises(ValueError):
        mp.mel_spectrogram_torch(audio, n_fft, 80, sr, hop, n_fft, 0.0, None, center=True, engine=eng)
