"""Service wrapper around the synthesis path.

The reference's web app (``inference_api.py:13, 35-65``) guards its one model with a NON-BLOCKING lock: a request
that arrives while another is being synthesised is answered "busy" at once (``mutex.acquire(blocking=False)``,
:37), otherwise ``infer`` runs and the waveform is written as a 44.1 kHz PCM16 WAV (:50).  ``SynthesisService``
keeps those semantics -- one synthesis in flight per model, callers are refused rather than queued -- and adds
what the MI355X path makes possible: the vocoder output is STREAMED, chunk by chunk, as PCM16 bytes
(``vsp_generator_stream_chunk``: the 13/14-frame-halo streamer, bit-identical to the one-shot waveform), so the
first audio leaves after one chunk instead of after the whole utterance.  Pure host logic; all arithmetic runs in
libvispeech_hip through ``vispeech_amd.models.SynthesizerTrn``.
"""
from __future__ import annotations

import io
import threading
import wave
from typing import Dict, Iterator, Optional, Sequence

import numpy as np


def pcm16(audio) -> np.ndarray:
    """float waveform in [-1, 1] -> little-endian int16 (the conversion of ``utils.write_wav``)."""
    a = audio.detach().cpu().numpy() if hasattr(audio, "detach") else np.asarray(audio)
    return np.clip(np.rint(np.asarray(a, dtype=np.float32).reshape(-1) * 32767.0), -32768, 32767).astype("<i2")


class Busy(RuntimeError):
    """Another synthesis is in flight (the reference answers such a request with a 'server busy' text)."""


class _LockedStream:
    """Iterator over the chunks of one streamed synthesis that OWNS the service's single-flight lock."""

    def __init__(self, service: "SynthesisService", chunks: Iterator[bytes]):
        self._service = service
        self._chunks = chunks
        self._held = True

    def __iter__(self):
        return self

    def __next__(self) -> bytes:
        if not self._held:
            raise StopIteration
        try:
            return next(self._chunks)
        except BaseException:          # exhausted (StopIteration) or failed: either way the synthesis is over
            self.close()
            raise

    def close(self) -> None:
        if self._held:
            self._held = False
            try:
                self._chunks.close()
            finally:
                self._service.release()

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class SynthesisService:
    """One model, one synthesis at a time, never queueing (reference inference_api.py:13, 37)."""

    def __init__(self, net, sampling_rate: int = 44100, chunk_frames: int = 64, noise_scale: float = 0.667, stream=None):
        self.net = net
        self.sampling_rate = int(sampling_rate)
        self.chunk_frames = int(chunk_frames)
        self.noise_scale = float(noise_scale)
        self._lock = threading.Lock()
        self._stream = stream          # a torch.cuda.Stream all of this service's GPU work runs on (None: the caller's)

    def _scope(self):
        """The stream scope of this service's GPU work (``PooledSynthesisService`` gives every slot its own stream)."""
        import contextlib
        if self._stream is None:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self._stream)

    # ------------------------------------------------------------------ single-flight
    def try_acquire(self) -> bool:
        return self._lock.acquire(blocking=False)

    def release(self) -> None:
        self._lock.release()

    @property
    def busy(self) -> bool:
        return self._lock.locked()

    # ------------------------------------------------------------------ one-shot (what the reference's /tts does)
    def synthesize(self, batch: Dict[str, "np.ndarray"], utterance: int = 0, noise=None) -> Optional[np.ndarray]:
        """``batch`` = the arrays of ``vispeech_amd.text.collate_rows`` (phonemes, lengths, sid and optionally
        duration / f0 / energy).  Returns the PCM16 samples of ``utterance`` (valid part only), or ``None`` if
        another request is in flight (the reference returns None -> "busy")."""
        if not self.try_acquire():
            return None
        try:
            with self._scope():
                o, frames = self._infer(batch, noise)
                hop = self.net.dims.total_upsample
                pcm = pcm16(o[utterance, 0, : int(frames[utterance]) * hop])      # (device -> host: the stream has drained)
            self._check_numerics()
            return pcm
        finally:
            self.release()

    def wav_bytes(self, batch, utterance: int = 0, noise=None) -> Optional[bytes]:
        """The reference's response body: a mono PCM16 WAV at the model's sampling rate (inference_api.py:50, 64)."""
        pcm = self.synthesize(batch, utterance, noise)
        if pcm is None:
            return None
        buf = io.BytesIO()
        with wave.open(buf, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(self.sampling_rate)
            w.writeframes(pcm.tobytes())
        return buf.getvalue()

    # ------------------------------------------------------------------ streamed
    def stream(self, batch, utterance: int = 0, noise=None) -> Iterator[bytes]:
        """PCM16 bytes of ``utterance``, one vocoder chunk (``chunk_frames`` frames) at a time.  Raises ``Busy``
        at once when another synthesis is in flight.  The lock is owned by the returned ``_LockedStream`` and is
        released exactly once: when the stream is exhausted, fails, is ``close()``d, or is dropped -- also when it
        was never started (a plain generator that is never advanced would never run its ``finally``).  The
        concatenation equals ``synthesize`` byte for byte."""
        if not self.try_acquire():
            raise Busy("another synthesis is in flight")
        return _LockedStream(self, self._stream_chunks(batch, utterance, noise))

    def _stream_chunks(self, batch, utterance, noise) -> Iterator[bytes]:
        import torch
        net, eng = self.net, self.net._engine
        with self._scope():
            enc, frames, tf = self._encode(batch)
            z_noise = noise if noise is not None else torch.randn(
                enc["x_var"].shape[0], net.dims.inter_channels, tf, dtype=torch.float32, device=eng.device)
            dec = eng.decode(enc, tf, z_noise, self.noise_scale, max_len=0)     # everything but the vocoder
            chunks = eng.generator_stream(dec["z"], enc["g"], self.chunk_frames)
        left = int(frames[utterance]) * net.dims.total_upsample
        while left > 0:
            # (the stream scope is entered per chunk, never held across a yield: the consumer's thread keeps its own stream)
            with self._scope():
                o = next(chunks, None)
                if o is None:
                    break
                piece = pcm16(o[utterance, 0, : min(left, o.shape[2])])
            self._check_numerics()
            left -= piece.size
            yield piece.tobytes()

    # ------------------------------------------------------------------ helpers
    def _check_numerics(self) -> None:
        """After a device -> host copy: raise if a kernel reported values outside the range the split-f16 matrix kernels
        represent (Engine.check_numerics; the audio just copied would be inf / NaN garbage)."""
        eng = getattr(self.net, "_engine", None)
        if eng is not None and hasattr(eng, "check_numerics"):
            eng.check_numerics(sync=False)

    def _controls(self, batch):
        return {k: batch.get(k) for k in ("duration", "f0", "energy")}

    def _encode(self, batch):
        import torch
        eng = self.net._engine
        c = self._controls(batch)
        t = lambda a: None if a is None else torch.as_tensor(np.asarray(a))
        enc = eng.encode(t(batch["phonemes"]), t(batch["lengths"]), t(batch["sid"]), t(c["duration"]), t(c["f0"]),
                         t(c["energy"]))
        frames, tf = eng.frame_lengths_host(enc["frame_lengths"])
        if tf <= 0:
            raise ValueError("all durations are zero: nothing to synthesise")
        return enc, frames, tf

    def _infer(self, batch, noise):
        import torch
        net = self.net
        c = self._controls(batch)
        t = lambda a: None if a is None else torch.as_tensor(np.asarray(a)).to(net.device)
        o, x_mask, *_ = net.infer(t(batch["phonemes"]), t(batch["lengths"]), sid=t(batch["sid"]),
                                  noise_scale=self.noise_scale, duration_control=t(c["duration"]),
                                  pitch_control=t(c["f0"]), energy_control=t(c["energy"]), noise=noise)
        return o, x_mask.sum(dim=(1, 2)).cpu().tolist()


class PooledSynthesisService:
    """Up to N syntheses in flight on one GPU (round 6): one single-flight ``SynthesisService`` per context of an
    ``InFlightPool``, each on its context's stream.  The reference's semantics generalised, not replaced: a request is
    served by the first FREE slot or refused at once (``None`` / ``Busy``) -- never queued (inference_api.py:13, 37 with
    N locks instead of one).  The frame-rate half of one request overlaps the vocoder of another: 3.1 -> 2.1 -> 1.7 ms per
    single-utterance request at 1 / 2 / 3 slots (profiles/r06_batches_in_flight.txt)."""

    def __init__(self, pool, sampling_rate: int = 44100, chunk_frames: int = 64, noise_scale: float = 0.667):
        self.slots = [SynthesisService(net, sampling_rate, chunk_frames, noise_scale, stream=st)
                      for net, st in zip(pool.nets, pool.streams if pool.streams[0] is not None else [None] * len(pool.nets))]

    @property
    def busy(self) -> bool:
        return all(s.busy for s in self.slots)

    def synthesize(self, batch, utterance: int = 0, noise=None) -> Optional[np.ndarray]:
        for s in self.slots:
            pcm = s.synthesize(batch, utterance, noise)          # (None = this slot is taken: try the next)
            if pcm is not None:
                return pcm
        return None

    def wav_bytes(self, batch, utterance: int = 0, noise=None) -> Optional[bytes]:
        for s in self.slots:
            wav = s.wav_bytes(batch, utterance, noise)
            if wav is not None:
                return wav
        return None

    def stream(self, batch, utterance: int = 0, noise=None) -> Iterator[bytes]:
        for s in self.slots:
            try:
                return s.stream(batch, utterance, noise)
            except Busy:
                continue
        raise Busy("every synthesis slot is taken")
