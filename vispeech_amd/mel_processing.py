"""The reference's ``mel_processing`` front end (reference mel_processing.py:50-112) on the MI355X path: same function
names and argument order, tensors in, CUDA tensors out.  The DFT matrix of the spectrogram lives in an engine's weight
arena, so the engine (``net._engine`` of a ``vispeech_amd.models.SynthesizerTrn``) is passed as a keyword; ``n_fft``
and ``win_size`` must be that model's ``filter_length`` (the reference always calls these with
``hps.data.filter_length`` / ``win_length`` of equal value, data_utils.py:60-64).  ``center`` must be False, as in
every call of the reference.

Parity status: the linear spectrogram is pinned to ``torch.stft`` (tests/test_mel.py); the MEL BASIS -- librosa's
``filters.mel`` in the reference (mel_processing.py:14, 78, 96) -- is pinned to a third party's implementation of that
routine, ``transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")`` ("adapted from torchaudio and
librosa"), to fp32 rounding (tests/test_oracle_golden.py).  librosa and torchaudio themselves are absent from the build
image: the reference's own call has not been run beside it."""
from typing import Optional

import torch

from . import _lib


def _check(engine, n_fft: int, win_size: int, center: bool):
    if engine is None:
        raise ValueError("pass engine=net._engine (the spectrogram's DFT matrix is part of the model's weight arena)")
    if center:
        raise ValueError("center=True is not supported (the reference never uses it)")
    want = 2 * (engine.dims.spec_channels - 1)
    if n_fft != want or win_size != want:
        raise ValueError(f"n_fft / win_size must be the model's filter_length {want}")


def spectrogram_torch(y, n_fft: int, sampling_rate: int, hop_size: int, win_size: int, center: bool = False, *,
                      engine=None) -> torch.Tensor:
    """reference mel_processing.py:50-69 -> [B, n_fft // 2 + 1, frames]."""
    _check(engine, n_fft, win_size, center)
    return engine.spectrogram(y, hop_size)


def spec_to_mel_torch(spec, n_fft: int, num_mels: int, sampling_rate: int, fmin: float, fmax: Optional[float], *,
                      engine=None) -> torch.Tensor:
    """reference mel_processing.py:73-82 -> [B, num_mels, frames] (log-mel)."""
    if engine is None:
        raise ValueError("pass engine=net._engine")
    if spec.shape[1] != n_fft // 2 + 1:
        raise ValueError("spec does not have n_fft // 2 + 1 rows")
    return engine.spec_to_mel(spec, num_mels, sampling_rate, fmin, fmax)


def mel_spectrogram_torch(y, n_fft: int, num_mels: int, sampling_rate: int, hop_size: int, win_size: int, fmin: float,
                          fmax: Optional[float], center: bool = False, *, engine=None) -> torch.Tensor:
    """reference mel_processing.py:85-112 -> [B, num_mels, frames] (log-mel)."""
    _check(engine, n_fft, win_size, center)
    return engine.mel_spectrogram(y, num_mels, sampling_rate, fmin, fmax, hop_size)


def mel_filterbank(sampling_rate: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: Optional[float] = None):
    """The basis ``librosa.filters.mel(sampling_rate, n_fft, n_mels, fmin, fmax)`` returns (host, numpy float32
    [n_mels, n_fft // 2 + 1]); computed by the library (vsp_mel_filterbank)."""
    import ctypes as C

    import numpy as np
    w = np.zeros((n_mels, n_fft // 2 + 1), dtype=np.float32)
    rc = _lib.lib().vsp_mel_filterbank(int(sampling_rate), int(n_fft), int(n_mels), float(fmin),
                                       0.0 if fmax is None else float(fmax), w.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("vsp_mel_filterbank: bad argument")
    return w
