"""Callers' side of the path ('next' rows 1 and 3): checkpoint loading with the
reference's tolerant semantics and a PCM16 WAV writer.  Pure host code, no GPU arithmetic."""
from __future__ import annotations

import logging
import struct
import wave
from typing import Optional, Tuple

import numpy as np

logger = logging.getLogger(__name__)


def load_checkpoint(checkpoint_path: str, model, optimizer=None) -> Tuple[object, Optional[object], float, int]:
    """Load a reference ``G_*.pth`` (``{'model','iteration','optimizer','learning_rate'}``, reference
    utils.py:67-70) into a ``vispeech_amd.models.SynthesizerTrn``.  Like the reference's
    ``utils.load_checkpoint`` (utils.py:21-51) it is tolerant: keys missing from the file keep the
    model's current value (here: must already be loaded, else the final load fails loudly) and
    mismatched shapes are reported and skipped.  Returns (model, optimizer, learning_rate, iteration)."""
    import torch
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    saved = ckpt["model"] if "model" in ckpt else ckpt
    from .schema import state_dict_schema
    schema = state_dict_schema(model.dims)
    current = {k: v for k, v in getattr(model, "_state", {}).items()}
    new = {}
    for k, shape in schema.items():
        if k in saved and tuple(saved[k].shape) == tuple(shape):
            new[k] = saved[k]
        else:
            if k in saved:
                logger.info("%s: shape %s in checkpoint, %s expected -- skipped", k, tuple(saved[k].shape), tuple(shape))
            else:
                logger.info("%s is not in the checkpoint", k)
            if k in current:
                new[k] = current[k]
    model.load_state_dict(new, strict=True)
    return model, optimizer, float(ckpt.get("learning_rate", 0.0)), int(ckpt.get("iteration", 0))


def write_wav(path: str, audio, sampling_rate: int = 44100) -> None:
    """float waveform in [-1, 1] ([S] or [1, S] or [1, 1, S]; torch or numpy) -> 16-bit PCM WAV
    (what the reference's apps do with scipy.io.wavfile.write, inference_api.py:50)."""
    a = audio.detach().cpu().numpy() if hasattr(audio, "detach") else np.asarray(audio)
    a = np.asarray(a, dtype=np.float32).reshape(-1)
    pcm = np.clip(np.rint(a * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sampling_rate))
        w.writeframes(pcm.tobytes())
