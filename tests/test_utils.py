import wave

import numpy as np

from vispeech_amd.utils import write_wav


def test_write_wav_roundtrip(tmp_path):
    x = np.sin(np.linspace(0, 20, 4410)).astype(np.float32) * 0.5
    p = tmp_path / "a.wav"
    write_wav(str(p), x[None, None, :], 44100)
    with wave.open(str(p), "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 44100, 4410)
        pcm = np.frombuffer(w.readframes(4410), dtype="<i2")
    assert np.abs(pcm / 32767.0 - x).max() <= 1.0 / 32767.0
