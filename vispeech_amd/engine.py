"""Thin torch-side driver of the C-ABI: owns a ``vsp_ctx``, hands torch CUDA tensors' pointers to
libvispeech_hip and keeps the caller-owned workspaces.  torch is plumbing here (device memory,
the current HIP stream); every arithmetic step runs in the library's kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from . import _lib
from .schema import ModelDims


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev_f32(t, device) -> torch.Tensor:
    return torch.as_tensor(t).to(device=device, dtype=torch.float32).contiguous()


def _dev_i64(t, device) -> torch.Tensor:
    return torch.as_tensor(t).to(device=device, dtype=torch.int64).contiguous()


class Engine:
    def __init__(self, dims: ModelDims, device: "torch.device | str | int" = "cuda:0"):
        self.lib = _lib.lib()                      # raises ImportError if the extension is absent
        self.dims = dims
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda":
            raise RuntimeError("vispeech_amd runs on an MI355X (torch device 'cuda'); there is no CPU path")
        self.cfg = _lib.make_config(dims)
        ctx = C.c_void_p()
        rc = self.lib.vsp_create(C.byref(self.cfg), self.device.index or 0, C.byref(ctx))
        self.ctx = ctx
        _lib.check(rc, ctx, "vsp_create")
        self._arena: Optional[torch.Tensor] = None
        self._ws: Dict[str, torch.Tensor] = {}
        self.ready = False

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.vsp_destroy(self.ctx)
                self.ctx = None
        except Exception:  # pragma: no cover
            pass

    # ------------------------------------------------------------------ weights
    def set_weights(self, state_dict: Mapping[str, "np.ndarray | torch.Tensor"], strict: bool = True):
        """One ``vsp_set_weight[_typed]`` per tensor of a fresh load (``vsp_begin_weights`` first: nothing of an
        earlier load survives).  float16 / bfloat16 / float64 checkpoints and tensors that already live on the
        device are handed over as they are (``vsp_set_weight_typed``); everything else is float32 host data."""
        missing, unexpected = [], []
        _lib.check(self.lib.vsp_begin_weights(self.ctx), self.ctx, "vsp_begin_weights")
        self.ready = False
        for k, v in state_dict.items():
            if torch.is_tensor(v) and str(v.dtype).replace("torch.", "") in _lib.DTYPES and (v.is_cuda or v.dtype != torch.float32):
                t = v.detach().contiguous()
                if t.is_cuda:
                    # the library copies with a blocking hipMemcpy on the NULL stream, which does not order against
                    # torch's (possibly non-blocking) streams: whatever produced `t` must have finished first
                    torch.cuda.synchronize(t.device)
                shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                rc = self.lib.vsp_set_weight_typed(self.ctx, k.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(),
                                                   _lib.DTYPES[str(t.dtype).replace("torch.", "")], int(t.is_cuda))
                if rc == -4:
                    unexpected.append(k)
                    continue
                _lib.check(rc, self.ctx, f"vsp_set_weight_typed({k})")
                continue
            a = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            rc = self.lib.vsp_set_weight(self.ctx, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)
            if rc == -4:
                unexpected.append(k)
                continue
            _lib.check(rc, self.ctx, f"vsp_set_weight({k})")
        n_missing = self.lib.vsp_missing_weights(self.ctx)
        if n_missing:
            missing.append(f"{n_missing} infer-path tensors")
        if strict and (unexpected or n_missing):
            raise RuntimeError(f"load_state_dict: unexpected keys {unexpected[:5]}..., missing {missing}")
        return missing, unexpected

    def arena_bytes(self) -> int:
        return int(self.lib.vsp_weight_arena_bytes(self.ctx))

    def _alloc_arena(self) -> torch.Tensor:
        if self._arena is None:
            self._arena = torch.empty(self.arena_bytes() // 4, dtype=torch.float32, device=self.device)
        return self._arena

    def finalize(self) -> torch.Tensor:
        """Fold, pack and upload; returns the packed arena as a flat float tensor (the object a
        multi-GPU run broadcasts from rank 0)."""
        arena = self._alloc_arena()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vsp_finalize_weights(self.ctx, _ptr(arena)), self.ctx, "vsp_finalize_weights")
        self.ready = True
        return arena

    def adopt(self, arena: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Non-root rank: use an arena whose bytes arrive by broadcast.  The engine is NOT ready until
        ``commit_adopted()`` has checked the header of the received bytes."""
        if arena is None:
            arena = self._alloc_arena()
        assert arena.numel() * 4 == self.arena_bytes() and arena.is_cuda
        self._arena = arena
        _lib.check(self.lib.vsp_adopt_packed_weights(self.ctx, _ptr(arena)), self.ctx, "vsp_adopt_packed_weights")
        self.ready = False
        return arena

    def commit_adopted(self) -> None:
        """After the broadcast: read the arena header (magic, ABI, size, config hash, what rank 0 packed)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vsp_commit_adopted_weights(self.ctx, self._stream()), self.ctx,
                       "vsp_commit_adopted_weights")
        self.ready = True

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, tag: str, nbytes: int) -> torch.Tensor:
        if nbytes < 0:
            _lib.check(int(nbytes), self.ctx, f"{tag} workspace size")
        w = self._ws.get(tag)
        if w is None or w.numel() < nbytes:
            self._ws[tag] = None
            w = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self._ws[tag] = w
        return w

    def _f(self, *shape) -> torch.Tensor:
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ the path
    def encode(self, phonemes, lengths, sid, duration_ctl=None, pitch_ctl=None, energy_ctl=None,
               duration_scale=1.0, pitch_scale=1.0, energy_scale=1.0) -> Dict[str, torch.Tensor]:
        d = self.dims
        ph = _dev_i64(phonemes, self.device)
        ln = _dev_i64(lengths, self.device)
        sd = _dev_i64(sid, self.device)
        B, Tp = ph.shape
        dc = None if duration_ctl is None else _dev_f32(duration_ctl, self.device).reshape(B, -1)
        pc = None if pitch_ctl is None else _dev_f32(pitch_ctl, self.device).reshape(B, -1)
        ec = None if energy_ctl is None else _dev_f32(energy_ctl, self.device).reshape(B, -1)
        for name, t in (("duration", dc), ("pitch", pc), ("energy", ec)):
            if t is not None and t.shape[1] != Tp:
                raise ValueError(f"{name}_control must have {Tp} entries per utterance, got {t.shape[1]}")
        out = dict(x_var=self._f(B, d.hidden_channels, Tp), g=self._f(B, d.gin_channels),
                   duration=self._f(B, Tp), F0=self._f(B, Tp), energy=self._f(B, Tp),
                   frame_lengths=torch.empty(B, dtype=torch.int64, device=self.device),
                   cum_dur=torch.empty(B, Tp, dtype=torch.int32, device=self.device))
        ws = self._workspace("encode", self.lib.vsp_encode_workspace_bytes(self.ctx, B, Tp))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_encode(self.ctx, self._stream(), B, Tp, _ptr(ph), _ptr(ln), _ptr(sd), _ptr(dc), _ptr(pc),
                                     _ptr(ec), float(duration_scale), float(pitch_scale), float(energy_scale),
                                     _ptr(out["x_var"]), _ptr(out["g"]), _ptr(out["duration"]), _ptr(out["F0"]),
                                     _ptr(out["energy"]), _ptr(out["frame_lengths"]), _ptr(out["cum_dur"]), _ptr(ws),
                                     ws.numel())
        _lib.check(rc, self.ctx, "vsp_encode")
        out["_keep"] = (ph, ln, sd, dc, pc, ec)
        return out

    def frame_lengths_host(self, frame_lengths: torch.Tensor):
        B = frame_lengths.numel()
        host = (C.c_int64 * B)()
        mx = C.c_int64()
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_frame_lengths_host(self.ctx, self._stream(), B, _ptr(frame_lengths), host, C.byref(mx))
        _lib.check(rc, self.ctx, "vsp_frame_lengths_host")
        self.check_numerics(sync=False)       # (free: a pinned word; reports what has completed, e.g. the previous call)
        return list(host), int(mx.value)

    # ------------------------------------------------------------------ numeric-range status (vsp_status)
    def status(self, clear: bool = False) -> int:
        """The context's sticky VSP_FLAG_* word (no synchronisation: launches that have completed)."""
        flags = C.c_uint(0)
        _lib.check(self.lib.vsp_status(self.ctx, C.byref(flags), int(clear)), self.ctx, "vsp_status")
        return int(flags.value)

    def check_numerics(self, sync: bool = True) -> None:
        """Raise ``VspError`` if a kernel has reported non-finite values since the last check -- an activation left the
        range the split-f16 matrix kernels represent (include/vispeech_hip.h, vsp_status) or an input was not finite.
        ``sync=True`` waits for the device first, so that the answer covers everything enqueued so far."""
        if sync:
            torch.cuda.synchronize(self.device)
        f = self.status(clear=True)
        if f:
            what = [n for b, n in ((_lib.FLAG_NONFINITE_LATENT, "z_p (phoneme- / frame-rate stages)"),
                                   (_lib.FLAG_NONFINITE_WAVE, "the waveform (flow / generator)")) if f & b]
            raise _lib.VspError("non-finite values in " + " and ".join(what) + ": an activation beyond the split-f16 range "
                                "(|x| > 65504; inside the generator 65504 / 2^VSP_ACT_SCALE_LOG2) or a non-finite input; "
                                "VSP_GENERATOR=f32 / VSP_FRAME=f32 select the f32 matrix kernels, which have no such limit")

    def decode_buffers(self, B: int, Tp: int, Tf: int, max_len: Optional[int] = None):
        """Output tensors and workspace of ``decode`` for a padded frame count ``Tf``.  A caller that knows ``Tf`` before
        the frame counts are read back (a sharded batch's global padding, ``t_f``) allocates them while the GPU is still
        busy with ``encode``: the host work between the two halves is then the read itself."""
        d = self.dims
        inter = d.inter_channels
        if max_len is not None and int(max_len) < 0:
            raise ValueError("max_len must be >= 0 (None = no truncation)")   # the C side reads < 0 as "no limit"
        Tdec = Tf if max_len is None else min(Tf, int(max_len))
        o_buf = self._f(B, 1, max(Tdec * d.total_upsample, 1))      # (never a null pointer: max_len = 0 skips the vocoder)
        out = dict(o=o_buf[:, :, :Tdec * d.total_upsample] if Tdec * d.total_upsample != o_buf.shape[2] else o_buf,
                   x_mask=torch.empty(B, 1, Tf, dtype=torch.uint8, device=self.device),
                   z=self._f(B, inter, Tf), z_p=self._f(B, inter, Tf), m_p=self._f(B, inter, Tf),
                   logs_p=self._f(B, inter, Tf))
        ws = self._workspace("decode", self.lib.vsp_decode_workspace_bytes(self.ctx, B, Tp, Tf))
        return dict(Tf=int(Tf), Tdec=Tdec, o_buf=o_buf, out=out, ws=ws, max_len=max_len)

    def decode(self, enc: Mapping[str, torch.Tensor], Tf: int, noise: Optional[torch.Tensor], noise_scale: float,
               max_len: Optional[int] = None, noise_seed: Optional[int] = None, bufs=None,
               noise_offset: int = 0) -> Dict[str, torch.Tensor]:
        """``noise`` None with ``noise_scale`` != 0: the library draws it on the device -- elements ``noise_offset`` ..
        of ``vsp_randn(noise_seed)``; the caller must then name the seed (a silent default would hand out the same
        "random" sample on every call).  ``noise_offset``: a shard [lo, hi) of a global batch passes lo * inter * Tf so
        that its utterances get the noise they would get unsharded.
        ``bufs``: a ``decode_buffers`` result for the same ``Tf`` / ``max_len`` (else allocated here)."""
        if noise is None and float(noise_scale) != 0.0 and noise_seed is None:
            raise ValueError("pass noise or an explicit noise_seed (noise_scale != 0)")
        noise_seed = 0 if noise_seed is None else noise_seed
        d = self.dims
        B, _, Tp = enc["x_var"].shape
        inter = d.inter_channels
        if bufs is None or bufs["Tf"] != int(Tf) or bufs["max_len"] != max_len:
            bufs = self.decode_buffers(B, Tp, Tf, max_len)
        Tdec, o_buf, out, ws = bufs["Tdec"], bufs["o_buf"], bufs["out"], bufs["ws"]
        if noise is not None:
            noise = _dev_f32(noise, self.device)
            if tuple(noise.shape) != (B, inter, Tf):
                raise ValueError(f"noise must be [{B},{inter},{Tf}], got {tuple(noise.shape)}")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vsp_set_noise_offset(self.ctx, int(noise_offset)), self.ctx, "vsp_set_noise_offset")
            rc = self.lib.vsp_decode(self.ctx, self._stream(), B, Tp, Tf, -1 if max_len is None else Tdec,
                                     _ptr(enc["x_var"]), _ptr(enc["g"]), _ptr(enc["cum_dur"]),
                                     _ptr(enc["frame_lengths"]), _ptr(noise), int(noise_seed) & (2**64 - 1),
                                     float(noise_scale), _ptr(o_buf),
                                     _ptr(out["x_mask"]), _ptr(out["z"]), _ptr(out["z_p"]), _ptr(out["m_p"]),
                                     _ptr(out["logs_p"]), _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_decode")
        out["x_mask"] = out["x_mask"].view(torch.bool) if hasattr(torch, "bool") else out["x_mask"]
        return out

    # ------------------------------------------------------------------ per-stage entry points
    def infer_padded(self, phonemes, lengths, sid, tf_pad: int, noise, noise_scale: float = 1.0, max_len=None,
                     duration_ctl=None, pitch_ctl=None, energy_ctl=None, duration_scale: float = 1.0,
                     pitch_scale: float = 1.0, energy_scale: float = 1.0, noise_seed: Optional[int] = None,
                     noise_offset: int = 0) -> Dict[str, torch.Tensor]:
        """``vsp_infer``: the whole path in ONE call and without the host read of the frame counts, for
        callers that know an upper bound ``tf_pad`` of the frame count (supplied durations / fixed max_len).
        ``noise`` None with ``noise_scale`` != 0 needs an explicit ``noise_seed`` (see ``decode``)."""
        if noise is None and float(noise_scale) != 0.0 and noise_seed is None:
            raise ValueError("pass noise or an explicit noise_seed (noise_scale != 0)")
        noise_seed = 0 if noise_seed is None else noise_seed
        ph = _dev_i64(phonemes, self.device)
        B, Tp = ph.shape
        ln, sd = _dev_i64(lengths, self.device), _dev_i64(sid, self.device)
        ctl = [None if t is None else _dev_f32(t, self.device).reshape(B, Tp) for t in (duration_ctl, pitch_ctl, energy_ctl)]
        Tf = int(tf_pad)
        if max_len is not None and int(max_len) < 0:
            raise ValueError("max_len must be >= 0 (None = no truncation)")
        Tdec = Tf if max_len is None else min(Tf, int(max_len))
        inter = self.dims.inter_channels
        ns = float(noise_scale)
        nz = None if noise is None else _dev_f32(noise, self.device)
        if nz is not None and tuple(nz.shape) != (B, inter, Tf):
            raise ValueError("noise must be [B, inter_channels, tf_pad]")
        o = self._f(B, 1, max(Tdec, 0) * self.dims.total_upsample)
        z, z_p, m_p, logs_p = (self._f(B, inter, Tf) for _ in range(4))
        x_mask = torch.empty((B, 1, Tf), dtype=torch.uint8, device=self.device)
        dur, f0, en = (self._f(B, Tp) for _ in range(3))
        fl = torch.empty(B, dtype=torch.int64, device=self.device)
        ws = self._workspace("infer", self.lib.vsp_infer_workspace_bytes(self.ctx, B, Tp, Tf))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vsp_set_noise_offset(self.ctx, int(noise_offset)), self.ctx, "vsp_set_noise_offset")
            rc = self.lib.vsp_infer(self.ctx, self._stream(), B, Tp, Tf, -1 if max_len is None else Tdec,
                                    _ptr(ph), _ptr(ln), _ptr(sd), _ptr(ctl[0]), _ptr(ctl[1]), _ptr(ctl[2]),
                                    float(duration_scale), float(pitch_scale), float(energy_scale), _ptr(nz),
                                    int(noise_seed) & (2**64 - 1), ns,
                                    _ptr(o), _ptr(x_mask), _ptr(z), _ptr(z_p), _ptr(m_p), _ptr(logs_p), _ptr(dur),
                                    _ptr(f0), _ptr(en), _ptr(fl), _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_infer")
        return dict(o=o, x_mask=x_mask.view(torch.bool), z=z, z_p=z_p, m_p=m_p, logs_p=logs_p, duration=dur, F0=f0,
                    energy=en, frame_lengths=fl)

    def attention(self, which: int, layer: int, qkv, lengths) -> torch.Tensor:
        """``vsp_attention``: relative-position attention of one encoder layer on ready q|k|v [B,3H,T]."""
        qkv = _dev_f32(qkv, self.device)
        ln = _dev_i64(lengths, self.device)
        B, C3, T = qkv.shape
        out = self._f(B, C3 // 3, T)
        ws = self._workspace("attention", self.lib.vsp_attention_workspace_bytes(self.ctx, B, T))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_attention(self.ctx, self._stream(), which, layer, B, T, _ptr(qkv), _ptr(ln), _ptr(out),
                                        _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_attention")
        return out

    def wn_layer(self, which: int, layer: int, x, g, lengths, skip=None):
        """``vsp_wn_layer``: one layer of modules.WN.forward (reference modules.py:148-176) of flow ``which``
        (-1: the posterior encoder's WN).  Returns (x_new, skip_new); ``skip`` None starts the skip sum."""
        x = _dev_f32(x, self.device).clone()
        B, h, T = x.shape
        g = _dev_f32(g, self.device).reshape(B, -1)
        ln = _dev_i64(lengths, self.device)
        acc = skip is not None
        sk = _dev_f32(skip, self.device).clone() if acc else self._f(B, h, T)
        ws = self._workspace("wn_layer", self.lib.vsp_wn_layer_workspace_bytes(self.ctx, which, B, T))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_wn_layer(self.ctx, self._stream(), which, layer, B, T, _ptr(x), _ptr(g), _ptr(ln), _ptr(sk),
                                       int(acc), _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_wn_layer")
        return x, sk

    def randn(self, seed: int, *shape, first: int = 0) -> torch.Tensor:
        """``vsp_randn_at``: elements ``first`` .. of the library's own standard-normal stream (Philox4x32-10 keyed by
        ``seed``)."""
        out = self._f(*shape)
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_randn_at(self._stream(), int(seed) & (2**64 - 1), int(first), out.numel(), _ptr(out))
        _lib.check(rc, self.ctx, "vsp_randn_at")
        return out

    def encoder(self, which: int, x, lengths) -> torch.Tensor:
        x = _dev_f32(x, self.device)
        ln = _dev_i64(lengths, self.device)
        B, h, T = x.shape
        y = self._f(B, h, T)
        ws = self._workspace("encoder", self.lib.vsp_encoder_workspace_bytes(self.ctx, B, T))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_encoder(self.ctx, self._stream(), which, B, T, _ptr(x), _ptr(ln), _ptr(y), _ptr(ws),
                                      ws.numel())
        _lib.check(rc, self.ctx, "vsp_encoder")
        return y

    def length_regulate(self, x, cum_dur, Tf: int) -> torch.Tensor:
        x = _dev_f32(x, self.device)
        cum = torch.as_tensor(cum_dur).to(device=self.device, dtype=torch.int32).contiguous()
        B, Cc, Tp = x.shape
        y = self._f(B, Cc, Tf)
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_length_regulate(self.ctx, self._stream(), B, Cc, Tp, Tf, _ptr(x), _ptr(cum), _ptr(y))
        _lib.check(rc, self.ctx, "vsp_length_regulate")
        return y

    def flow_reverse(self, z_p, g, frame_lengths) -> torch.Tensor:
        z_p = _dev_f32(z_p, self.device)
        g = _dev_f32(g, self.device).reshape(z_p.shape[0], -1)
        fl = _dev_i64(frame_lengths, self.device)
        B, _, Tf = z_p.shape
        z = torch.empty_like(z_p)
        ws = self._workspace("flow", self.lib.vsp_flow_workspace_bytes(self.ctx, B, Tf))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_flow_reverse(self.ctx, self._stream(), B, Tf, _ptr(z_p), _ptr(g), _ptr(fl), _ptr(z),
                                           _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_flow_reverse")
        return z

    def flow_forward(self, z, g, frame_lengths) -> torch.Tensor:
        """ResidualCouplingBlock.forward(reverse=False) (reference models.py:202-206)."""
        z = _dev_f32(z, self.device)
        g = _dev_f32(g, self.device).reshape(z.shape[0], -1)
        fl = _dev_i64(frame_lengths, self.device)
        B, _, Tf = z.shape
        z_p = torch.empty_like(z)
        ws = self._workspace("flow", self.lib.vsp_flow_workspace_bytes(self.ctx, B, Tf))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_flow_forward(self.ctx, self._stream(), B, Tf, _ptr(z), _ptr(g), _ptr(fl), _ptr(z_p),
                                           _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_flow_forward")
        return z_p

    # ------------------------------------------------------------------ voice conversion
    @property
    def has_voice_conversion(self) -> bool:
        return bool(self.lib.vsp_has_voice_conversion(self.ctx))

    def spectrogram(self, audio, hop_length: Optional[int] = None) -> torch.Tensor:
        """``mel_processing.spectrogram_torch`` (reference mel_processing.py:50-69): audio [B, L] in [-1, 1] ->
        linear magnitude spectrogram [B, spec_channels, L // hop] (n_fft = win = 2 * (spec_channels - 1))."""
        a = _dev_f32(audio, self.device)
        if a.dim() != 2:
            raise ValueError("audio must be [B, L]")
        hop = int(self.dims.hop_length if hop_length is None else hop_length)
        B, L = a.shape
        T = int(self.lib.vsp_spectrogram_frames(self.ctx, L, hop))
        if T <= 0:
            raise ValueError("signal too short for the reflect padding of the spectrogram")
        spec = self._f(B, self.dims.spec_channels, T)
        ws = self._workspace("spectrogram", self.lib.vsp_spectrogram_workspace_bytes(self.ctx, B, L, hop))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_spectrogram(self.ctx, self._stream(), B, L, hop, _ptr(a), _ptr(spec), _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_spectrogram")
        return spec

    def spec_to_mel(self, spec, n_mels: int, sampling_rate: int, fmin: float = 0.0, fmax: Optional[float] = None) -> torch.Tensor:
        """``mel_processing.spec_to_mel_torch`` (reference mel_processing.py:73-82): linear magnitude spectrogram
        [B, n_fft // 2 + 1, T] -> log-mel [B, n_mels, T] (Slaney basis as ``librosa.filters.mel``, log(clamp(x, 1e-5)))."""
        sp = _dev_f32(spec, self.device)
        if sp.dim() != 3 or sp.shape[1] < 2:
            raise ValueError("spec must be [B, n_fft // 2 + 1, T]")
        B, nf, T = sp.shape
        mel = self._f(B, int(n_mels), T)
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_spec_to_mel(self._stream(), B, T, 2 * (nf - 1), int(n_mels), int(sampling_rate), float(fmin),
                                          0.0 if fmax is None else float(fmax), _ptr(sp), _ptr(mel))
        _lib.check(rc, self.ctx, "vsp_spec_to_mel")
        return mel

    def mel_spectrogram(self, audio, n_mels: int, sampling_rate: int, fmin: float = 0.0, fmax: Optional[float] = None,
                        hop_length: Optional[int] = None) -> torch.Tensor:
        """``mel_processing.mel_spectrogram_torch`` (reference mel_processing.py:85-112) on the GPU: spectrogram, then
        the mel projection and dynamic range compression."""
        return self.spec_to_mel(self.spectrogram(audio, hop_length), n_mels, sampling_rate, fmin, fmax)

    def posterior_encoder(self, y, y_lengths, g, noise):
        """PosteriorEncoder.forward (reference models.py:233-241) -> (z, m, logs)."""
        y = _dev_f32(y, self.device)
        g = _dev_f32(g, self.device).reshape(y.shape[0], -1)
        yl = _dev_i64(y_lengths, self.device)
        B, _, T = y.shape
        noise = _dev_f32(noise, self.device)
        z, m, logs = (self._f(B, self.dims.inter_channels, T) for _ in range(3))
        ws = self._workspace("posterior", self.lib.vsp_posterior_workspace_bytes(self.ctx, B, T))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_posterior_encoder(self.ctx, self._stream(), B, T, _ptr(y), _ptr(yl), _ptr(g), _ptr(noise),
                                                _ptr(z), _ptr(m), _ptr(logs), _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_posterior_encoder")
        return z, m, logs

    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt, noise) -> Dict[str, torch.Tensor]:
        """SynthesizerTrn.voice_conversion (reference models.py:724-732); ``noise`` [B,inter,T] is the
        ``torch.randn_like`` of the posterior encoder (models.py:240)."""
        y = _dev_f32(y, self.device)
        B, S, T = y.shape
        if S != self.dims.spec_channels:
            raise ValueError(f"y has {S} channels, the model's spec_channels is {self.dims.spec_channels}")
        yl, ss, st = (_dev_i64(t, self.device) for t in (y_lengths, sid_src, sid_tgt))
        noise = _dev_f32(noise, self.device)
        if tuple(noise.shape) != (B, self.dims.inter_channels, T):
            raise ValueError("noise must be [B, inter_channels, T]")
        inter = self.dims.inter_channels
        o = self._f(B, 1, T * self.dims.total_upsample)
        z, z_p, z_hat, m_q, logs_q = (self._f(B, inter, T) for _ in range(5))
        y_mask = torch.empty((B, 1, T), dtype=torch.uint8, device=self.device)
        ws = self._workspace("vc", self.lib.vsp_voice_conversion_workspace_bytes(self.ctx, B, T))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_voice_conversion(self.ctx, self._stream(), B, T, _ptr(y), _ptr(yl), _ptr(ss), _ptr(st),
                                               _ptr(noise), _ptr(o), _ptr(y_mask), _ptr(z), _ptr(z_p), _ptr(z_hat),
                                               _ptr(m_q), _ptr(logs_q), _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_voice_conversion")
        return dict(o_hat=o, y_mask=y_mask, z=z, z_p=z_p, z_hat=z_hat, m_q=m_q, logs_q=logs_q)

    def generator(self, z, g) -> torch.Tensor:
        z = _dev_f32(z, self.device)
        g = _dev_f32(g, self.device).reshape(z.shape[0], -1)
        B, _, T = z.shape
        o = self._f(B, 1, T * self.dims.total_upsample)
        ws = self._workspace("generator", self.lib.vsp_generator_workspace_bytes(self.ctx, B, T))
        with torch.cuda.device(self.device):
            rc = self.lib.vsp_generator(self.ctx, self._stream(), B, T, _ptr(z), _ptr(g), _ptr(o), _ptr(ws), ws.numel())
        _lib.check(rc, self.ctx, "vsp_generator")
        return o

    @property
    def generator_halo(self) -> int:
        """Frames of context the streamed vocoder adds on each side (``vsp_generator_halo_frames``)."""
        return int(self.lib.vsp_generator_halo_frames(self.ctx))

    def generator_frame_dependence(self):
        """(back, fwd): output frame F of the vocoder depends on input frames [F - back, F + fwd], sample-exact from the
        configuration (``vsp_generator_frame_dependence``; 13 / 13 for configs/config.json) -- what the trimmed tails of a
        ragged batch rest on: an utterance of L frames is computed to min(T, L + back + 1 + fwd) frames."""
        b, f = C.c_int(), C.c_int()
        _lib.check(self.lib.vsp_generator_frame_dependence(self.ctx, C.byref(b), C.byref(f)), self.ctx,
                   "vsp_generator_frame_dependence")
        return int(b.value), int(f.value)

    def generator_stream(self, z, g, chunk_frames: int = 256):
        """Streamed vocoder (BASELINE config 5): yields the waveform of ``z`` [B][C][T] chunk by chunk
        ([B,1,512*n] tensors) so that the first audio is available after one chunk instead of after
        the whole utterance.  Each chunk is one ``vsp_generator_stream_chunk`` call: the generator on its
        frames plus the halo on both sides (zero padding only at the true ends), so the concatenation is
        bit-identical to one ``generator(z, g)`` call."""
        z = _dev_f32(z, self.device)
        g = _dev_f32(g, self.device).reshape(z.shape[0], -1)
        B, _, T = z.shape
        up = self.dims.total_upsample
        ws = self._workspace("generator_stream", self.lib.vsp_generator_stream_workspace_bytes(self.ctx, B, chunk_frames))
        for f0 in range(0, T, chunk_frames):
            f1 = min(T, f0 + chunk_frames)
            o = self._f(B, 1, (f1 - f0) * up)
            with torch.cuda.device(self.device):
                rc = self.lib.vsp_generator_stream_chunk(self.ctx, self._stream(), B, T, _ptr(z), _ptr(g), f0, f1,
                                                         _ptr(o), _ptr(ws), ws.numel())
            _lib.check(rc, self.ctx, "vsp_generator_stream_chunk")
            yield o

    def profile(self, on: bool) -> None:
        _lib.check(self.lib.vsp_profile_enable(self.ctx, int(on)), self.ctx, "vsp_profile_enable")

    def profile_read(self, reset: bool = True, cls: int = _lib.PROF_GENERATOR):
        """(launches, ms, algorithmic FLOPs, SURVEY-8d bytes, bytes incl. residual / accumulate reads, bytes the launches
        move as fused) of one class."""
        n, ms, fl, by, bx, bm = C.c_int64(), C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.lib.vsp_profile_read_class(self.ctx, int(cls), C.byref(n), C.byref(ms), C.byref(fl), C.byref(by),
                                                   C.byref(bx), C.byref(bm), int(reset)), self.ctx, "vsp_profile_read_class")
        return int(n.value), float(ms.value), float(fl.value), float(by.value), float(bx.value), float(bm.value)


    _FAMILY_KINDS = {0: "other", 1: "conv", 2: "ups", 3: "pair", 4: "chain", 5: "pre"}

    def profile_read_families(self, cls: int = _lib.PROF_GENERATOR, max_families: int = 64):
        """Per kernel family of one class since the last reset (call BEFORE profile_read(reset=True)):
        [{kind, channels, launches, ms, flops, bytes, moved}], largest total time first."""
        m = int(max_families)
        fam, n = (C.c_int * m)(), (C.c_int64 * m)()
        ms, fl, by, bm = (C.c_double * m)(), (C.c_double * m)(), (C.c_double * m)(), (C.c_double * m)()
        k = self.lib.vsp_profile_read_families(self.ctx, int(cls), m, fam, n, ms, fl, by, bm)
        _lib.check(min(k, 0), self.ctx, "vsp_profile_read_families")
        out = [dict(kind=self._FAMILY_KINDS.get(fam[i] & 7, "other"), channels=32 << (fam[i] >> 3), launches=int(n[i]),
                    ms=float(ms[i]), flops=float(fl[i]), bytes=float(by[i]), moved=float(bm[i])) for i in range(k)]
        return sorted(out, key=lambda d: -d["ms"])


def rq_spline(x, uw, uh, ud, inverse: bool = False, tail_bound: float = 5.0):
    """piecewise_rational_quadratic_transform(..., tails='linear') on the GPU (reference
    transforms.py:12-193).  x [...]; uw, uh [..., nb]; ud [..., nb-1]."""
    l = _lib.lib()
    dev = torch.device("cuda", torch.cuda.current_device()) if not (torch.is_tensor(x) and x.is_cuda) else x.device
    x = _dev_f32(x, dev)
    uw, uh, ud = _dev_f32(uw, dev), _dev_f32(uh, dev), _dev_f32(ud, dev)
    nb = uw.shape[-1]
    n = x.numel()
    y, lad = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = l.vsp_rq_spline(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), n, nb, _ptr(x), _ptr(uw), _ptr(uh),
                             _ptr(ud), int(inverse), float(tail_bound), _ptr(y), _ptr(lad))
    _lib.check(rc, None, "vsp_rq_spline")
    return y, lad
