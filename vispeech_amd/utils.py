"""Callers' side of the path ('next' rows 1 and 3): checkpoint loading with the
reference's tolerant semantics and a PCM16 WAV writer.  Pure host code, no GPU arithmetic."""
from __future__ import annotations

import logging
import struct
import wave
from typing import Dict, List, Mapping, Optional, Tuple

import numpy as np

logger = logging.getLogger(__name__)


def merge_checkpoint_state(saved: Mapping, schema: Mapping, current: Mapping) -> Tuple[Dict, List[str], List[str]]:
    """The tolerant key handling of the reference's ``utils.load_checkpoint`` (reference utils.py:32-43): for every
    key of the MODEL's schema take the checkpoint's tensor if it is there and has the model's shape, else keep the
    model's current value and report it.  Returns (new_state, missing_keys, mismatched_keys); a key that is neither
    usable from the checkpoint nor present in ``current`` is left out of new_state (the strict load that follows
    fails loudly for it if ``infer`` needs it -- the reference would keep its random init there)."""
    new, missing, mismatched = {}, [], []
    for k, shape in schema.items():
        if k in saved and tuple(saved[k].shape) == tuple(shape):
            new[k] = saved[k]
            continue
        if k in saved:
            mismatched.append(k)
            logger.info("%s: shape %s in the checkpoint, %s expected -- kept the model's value", k, tuple(saved[k].shape), tuple(shape))
        else:
            missing.append(k)
            logger.info("%s is not in the checkpoint", k)
        if k in current:
            new[k] = current[k]
    return new, missing, mismatched


def load_checkpoint(checkpoint_path: str, model, optimizer=None, skip_optimizer: bool = False,
                    trusted_pickle: bool = False) -> Tuple[object, Optional[object], float, int]:
    """Load a reference ``G_*.pth`` (``{'model','iteration','optimizer','learning_rate'}``, reference
    utils.py:67-70) into a ``vispeech_amd.models.SynthesizerTrn``; same signature and return value as the
    reference's ``utils.load_checkpoint`` (utils.py:21-51): (model, optimizer, learning_rate, iteration).
    The file is read with torch's restricted unpickler (tensors, dicts, numbers: all a reference checkpoint
    holds); ``trusted_pickle=True`` opts out for legacy files that need arbitrary pickles."""
    import os
    import torch
    assert os.path.isfile(checkpoint_path), checkpoint_path           # as the reference (utils.py:22)
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=not trusted_pickle)
    iteration = ckpt["iteration"]
    learning_rate = ckpt["learning_rate"]
    if optimizer is not None and not skip_optimizer:
        optimizer.load_state_dict(ckpt["optimizer"])
    from .schema import state_dict_schema
    new, _, _ = merge_checkpoint_state(ckpt["model"], state_dict_schema(model.dims), getattr(model, "_state", {}))
    model.load_state_dict(new, strict=True)
    logger.info("Loaded checkpoint '%s' (iteration %s)", checkpoint_path, iteration)
    return model, optimizer, learning_rate, iteration


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
