"""Configuration objects for the synthesis path.

The reference reads one JSON file into a recursive attribute-dict
(reference utils.py:281-310 ``HParams``, utils.py:214-224
``get_hparams_from_file``) and splats ``hps.model`` into the
``SynthesizerTrn`` constructor (reference inference.py:26-33).  This module
offers the same two entry points so user config files work unchanged, plus
``default_hparams()`` which re-types the keys of the reference's
configs/config.json that the inference path consumes (the speaker-name table
is training data and is not needed: only ``n_speakers`` is).
"""
from __future__ import annotations

import json
from typing import Any, Dict


class HParams:
    """Nested attribute/dict hybrid (interface of reference utils.py:281-310)."""

    def __init__(self, **kwargs: Any) -> None:
        for k, v in kwargs.items():
            self[k] = HParams(**v) if isinstance(v, dict) else v

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def to_dict(self) -> Dict[str, Any]:
        return {k: (v.to_dict() if isinstance(v, HParams) else v) for k, v in self.items()}

    def __len__(self) -> int:
        return len(self.__dict__)

    def __getitem__(self, key: str) -> Any:
        return getattr(self, key)

    def __setitem__(self, key: str, value: Any) -> None:
        setattr(self, key, value)

    def __contains__(self, key: str) -> bool:
        return key in self.__dict__

    def __repr__(self) -> str:
        return repr(self.__dict__)


def get_hparams_from_file(config_path: str) -> HParams:
    """Load an unmodified reference-style config JSON."""
    with open(config_path, "r") as f:
        return HParams(**json.loads(f.read()))


# Number of phoneme symbols of the reference's table (text/symbols.py:39:
# "_" + 401 zh + 42 ja + 69 en + 6 punctuation).  See vispeech_amd/text.py.
N_SYMBOLS = 519


def default_config_dict() -> Dict[str, Any]:
    """The inference-relevant content of the reference's configs/config.json."""
    return {
        "train": {"seed": 1234, "segment_size": 16384},
        "data": {
            "max_wav_value": 32768.0,
            "sampling_rate": 44100,
            "filter_length": 2048,
            "hop_length": 512,
            "win_length": 2048,
            "n_mel_channels": 80,
            "add_blank": True,
            "n_speakers": 200,
            "cleaned_text": True,
        },
        "model": {
            "inter_channels": 192,
            "hidden_channels": 192,
            "filter_channels": 768,
            "n_heads": 2,
            "n_layers": 4,
            "kernel_size": 3,
            "p_dropout": 0.1,
            "resblock": "1",
            "resblock_kernel_sizes": [3, 7, 11],
            "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
            "upsample_rates": [8, 8, 4, 2],
            "upsample_initial_channel": 512,
            "upsample_kernel_sizes": [16, 16, 4, 4],
            "n_layers_q": 3,
            "use_spectral_norm": False,
            "gin_channels": 256,
            "f0_mean": 171.21,
            "f0_std": 128.9,
            "freeze_textencoder": False,
            "freeze_decoder": False,
        },
    }


def default_hparams() -> HParams:
    return HParams(**default_config_dict())


def synthesizer_args(hps: HParams, n_vocab: int = N_SYMBOLS):
    """Positional/keyword arguments exactly as reference inference.py:26-33 builds them."""
    args = (
        n_vocab,
        hps.data.filter_length // 2 + 1,
        hps.data.hop_length,
        hps.data.sampling_rate,
        hps.train.segment_size // hps.data.hop_length,
    )
    kwargs = dict(n_speakers=hps.data.n_speakers, **hps.model.to_dict())
    return args, kwargs
