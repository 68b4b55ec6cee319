"""Phoneme-ID front door: the callers' side of the path.

The reference maps cleaned phoneme strings to ids with a 519-entry table (``text/symbols.py:39``:
"_" + zh + ja + en + punctuation) and stores training/inference rows as
``spk|id|phones|durations|f0|energy`` (``data_utils.py:52, 94-102``; fields space separated).  The
table is the reference's DATA and is not duplicated here: pass the list (``from text.symbols import
symbols`` in a checkout of the reference, or a JSON/text file with one symbol per line) to
``SymbolTable``.  Everything else -- id lookup, row parsing, padding a list of rows into the tensors
``SynthesizerTrn.infer`` takes -- is implemented here.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Dict, Iterable, List, Sequence

import numpy as np


class SymbolTable:
    def __init__(self, symbols: Sequence[str]):
        self.symbols = list(symbols)
        if len(set(self.symbols)) != len(self.symbols):
            raise ValueError("duplicate symbols")
        self._id: Dict[str, int] = {s: i for i, s in enumerate(self.symbols)}

    @classmethod
    def from_file(cls, path: str) -> "SymbolTable":
        text = open(path, encoding="utf-8").read()
        if path.endswith(".json"):
            return cls(json.loads(text))
        return cls([l.rstrip("\n") for l in text.splitlines() if l != ""])

    def __len__(self) -> int:
        return len(self.symbols)

    def cleaned_text_to_sequence(self, cleaned_text: Iterable[str]) -> List[int]:
        """reference text/__init__.py:9-17 (KeyError on unknown symbols, like the reference)."""
        return [self._id[s] for s in cleaned_text]


@dataclass
class FilelistRow:
    speaker: str
    utt_id: str
    phones: List[str]
    durations: np.ndarray   # int frames per phoneme (MFA, hop 512)
    f0: np.ndarray          # Hz per phoneme (0 = unvoiced)
    energy: np.ndarray


def parse_filelist_row(line: str) -> FilelistRow:
    """``spk|id|phones|durations|f0|energy`` (reference data_utils.py:52, 94-102)."""
    parts = line.rstrip("\n").split("|")
    if len(parts) != 6:
        raise ValueError(f"expected 6 '|'-separated fields, got {len(parts)}")
    spk, uid, phones, durs, f0s, ens = parts
    ph = phones.split(" ")
    d = np.array([int(x) for x in durs.split(" ")], dtype=np.int64)
    f0 = np.array([float(x) for x in f0s.strip().split(" ")], dtype=np.float32)
    en = np.array([float(x) for x in ens.strip().split(" ")], dtype=np.float32)
    if not (len(ph) == len(d) == len(f0) == len(en)):      # the reference asserts the same (data_utils.py:90-91)
        raise ValueError("phones / durations / f0 / energy lengths differ")
    return FilelistRow(spk, uid, ph, d, f0, en)


def collate_rows(rows: Sequence[FilelistRow], table: SymbolTable, spk2id: Dict[str, int]):
    """Pad rows into the arrays ``SynthesizerTrn.infer`` takes with control tensors:
    phonemes [B,Tp] int64, lengths [B], sid [B], duration / f0 / energy [B,Tp] float32 (zero padded)."""
    B = len(rows)
    tp = max(len(r.phones) for r in rows)
    out = dict(phonemes=np.zeros((B, tp), np.int64), lengths=np.zeros(B, np.int64), sid=np.zeros(B, np.int64),
               duration=np.zeros((B, tp), np.float32), f0=np.zeros((B, tp), np.float32),
               energy=np.zeros((B, tp), np.float32))
    for b, r in enumerate(rows):
        n = len(r.phones)
        out["phonemes"][b, :n] = table.cleaned_text_to_sequence(r.phones)
        out["lengths"][b] = n
        out["sid"][b] = spk2id[r.speaker]
        out["duration"][b, :n] = r.durations
        out["f0"][b, :n] = r.f0
        out["energy"][b, :n] = r.energy
    return out
