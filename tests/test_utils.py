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


def test_front_door_row_parsing_and_collation(tmp_path):
    from vispeech_amd.text import SymbolTable, collate_rows, parse_filelist_row
    table = SymbolTable(["_", "a.", "k.", "pau", "sp"])
    rows = [parse_filelist_row("nene|u1|k. a. pau|3 5 0|0 220.5 0|10 40 1\n"),
            parse_filelist_row("nene|u2|a. sp|4 2|180 0|30 2")]
    batch = collate_rows(rows, table, {"nene": 64})
    assert batch["phonemes"].tolist() == [[2, 1, 3], [1, 4, 0]]
    assert batch["lengths"].tolist() == [3, 2] and batch["sid"].tolist() == [64, 64]
    assert batch["duration"].sum(axis=1).tolist() == [8.0, 6.0]
    assert batch["f0"][0, 1] == np.float32(220.5)
    p = tmp_path / "symbols.txt"
    p.write_text("\n".join(table.symbols) + "\n", encoding="utf-8")
    assert SymbolTable.from_file(str(p)).symbols == table.symbols
    import pytest
    with pytest.raises(KeyError):
        table.cleaned_text_to_sequence(["zz"])
    with pytest.raises(ValueError):
        parse_filelist_row("a|b|c")
