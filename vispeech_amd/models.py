"""Drop-in ``SynthesizerTrn`` for the inference path.

Mirrors the reference's class surface for ``infer`` and nothing else:
the constructor signature (reference models.py:537-561), ``infer`` with the same arguments and the
same 6-tuple result (reference models.py:672-722), ``load_state_dict`` accepting reference
checkpoints unchanged (753-tensor schema, weight_g/weight_v pairs included), ``eval()``, ``to()``.
``voice_conversion`` (reference models.py:724-732) is served as well when the checkpoint carries the
``enc_q.*`` tensors.  Training-time members (``forward``, discriminators) are out of scope and raise.  All arithmetic runs in libvispeech_hip on the MI355X; if the extension is not
built, constructing the model raises ImportError.
"""
from __future__ import annotations

from typing import Mapping, Optional

import numpy as np
import torch

from .engine import Engine
from .schema import ModelDims, dims_from_ctor, state_dict_schema, used_by_infer


class SynthesizerTrn:
    """Synthesizer for inference (reference models.py:532-722)."""

    def __init__(self, n_vocab, spec_channels, hop_length, sampling_rate, segment_size, inter_channels,
                 hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, resblock,
                 resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates, upsample_initial_channel,
                 upsample_kernel_sizes, n_speakers=0, gin_channels=0, use_sdp=False, freeze_textencoder=False,
                 freeze_decoder=False, device="cuda:0", **kwargs):
        if str(resblock) != "1":
            raise NotImplementedError("only resblock '1' (ResBlock1) is on the reference's configured path")
        self.dims: ModelDims = dims_from_ctor(
            n_vocab, spec_channels, hop_length, sampling_rate, segment_size, inter_channels, hidden_channels,
            filter_channels, n_heads, n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes,
            resblock_dilation_sizes, upsample_rates, upsample_initial_channel, upsample_kernel_sizes,
            n_speakers=n_speakers, gin_channels=gin_channels)
        self.n_speakers = n_speakers
        self.gin_channels = gin_channels
        self.use_sdp = use_sdp            # stored and ignored, as in the reference (models.py:583)
        self.hop_length = hop_length
        self.sampling_rate = sampling_rate
        self.training = False
        self._engine = Engine(self.dims, device)
        self._state: "dict[str, np.ndarray]" = {}

    # ---------------------------------------------------------------- nn.Module-like surface
    @property
    def device(self) -> torch.device:
        return self._engine.device

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("training is out of scope of the MI355X synthesis path")
        return self

    def to(self, device):
        if torch.device(device) != self._engine.device:
            eng = Engine(self.dims, device)
            if self._state:
                eng.set_weights(self._state, strict=False)
                eng.finalize()
            self._engine = eng
        return self

    def cuda(self, device=None):
        return self.to(f"cuda:{0 if device is None else device}")

    def state_dict(self):
        """Every tensor of the reference's 753-key schema, in the reference's order: the loaded value where one has been
        loaded, zeros of the schema's shape otherwise -- so that the reference's own loader (utils.py:29-46), which
        iterates ``model.state_dict()`` BEFORE the first load and keeps the model's value for keys the checkpoint
        lacks, runs against this class as written (inference.py:36).  (The reference keeps its random init there; this
        class has no init: a key the checkpoint lacks loads as zeros.)"""
        from collections import OrderedDict
        out = OrderedDict()
        for k, shape in state_dict_schema(self.dims).items():
            v = self._state.get(k)
            out[k] = torch.from_numpy(v.copy()) if v is not None else torch.zeros(shape, dtype=torch.float32)
        for k, v in self._state.items():
            if k not in out:
                out[k] = torch.from_numpy(v.copy())
        return out

    def load_state_dict(self, state_dict: Mapping[str, "torch.Tensor | np.ndarray"], strict: bool = True):
        """Accepts a reference ``net_g.state_dict()`` / ``checkpoint['model']`` unchanged."""
        schema = state_dict_schema(self.dims)
        host = {}
        for k, v in state_dict.items():
            # float32 host copy (what state_dict() / to() hand back); the engine itself receives the tensors as
            # they are, so a float16 / bfloat16 / float64 or device-resident checkpoint goes through the typed entry
            a = v.detach().to(torch.float32).cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            host[k] = np.ascontiguousarray(a, dtype=np.float32)
        missing = [k for k in schema if used_by_infer(k) and k not in host]
        if strict and missing:
            raise RuntimeError(f"load_state_dict: {len(missing)} missing keys, e.g. {missing[:3]}")
        _, unexpected = self._engine.set_weights(state_dict, strict=strict)
        self._state = host
        self._engine.finalize()
        return missing, unexpected

    def forward(self, *a, **k):
        raise NotImplementedError("SynthesizerTrn.forward is the training path (reference models.py:624-670): out of scope")

    @torch.no_grad()
    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt, *, noise: Optional[torch.Tensor] = None):
        """Reference models.py:724-732: posterior encoder on the linear spectrogram ``y``
        [B, spec_channels, T] with the source speaker, flow forward (source), flow reverse (target),
        generator (target).  ``noise`` (keyword-only, optional) replaces the ``torch.randn_like`` of
        the posterior encoder (models.py:240).  Returns ``(o_hat, y_mask, (z, z_p, z_hat))``; needs the
        ``enc_q.*`` tensors in the loaded state_dict."""
        eng = self._engine
        if not eng.ready:
            raise RuntimeError("weights not loaded: call load_state_dict first")
        assert self.n_speakers > 0, "n_speakers have to be larger than 0."      # models.py:725
        if not eng.has_voice_conversion:
            raise RuntimeError("voice_conversion needs the enc_q.* tensors: the loaded state_dict had none")
        B, _, T = y.shape
        if noise is None:
            noise = torch.randn(B, self.dims.inter_channels, T, dtype=torch.float32, device=eng.device)
        r = eng.voice_conversion(y, y_lengths, sid_src, sid_tgt, noise)
        return r["o_hat"], r["y_mask"].to(torch.float32), (r["z"], r["z_p"], r["z_hat"])

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    # ---------------------------------------------------------------- the hot path
    @torch.no_grad()
    def infer(self, phonemes, phonemes_lengths, sid=None, noise_scale=1, max_len=None, energy_control=None,
              pitch_control=None, duration_control=None, *, noise: Optional[torch.Tensor] = None,
              t_f: Optional[int] = None, noise_seed: Optional[int] = None, noise_offset: int = 0):
        """Reference models.py:672-722.  ``noise`` (keyword-only, optional) replaces the
        ``torch.randn_like`` draw of models.py:718 so runs can be reproduced; without it the draw is
        ``torch.randn`` on the GPU (torch's generator, as in the reference) unless ``noise_seed`` is given: then the
        library draws it itself (``vsp_randn``, what a C caller gets) from stream element ``noise_offset`` on (a shard
        [lo, hi) of a global batch passes lo * inter_channels * t_f: same noise as unsharded).  ``t_f`` pads the frame
        axis to a global maximum for sharded batches (SURVEY gotcha G6).  Returns
        ``(o, x_mask, (z, z_p, m_p, logs_p), duration, F0, energy)``."""
        eng = self._engine
        if not eng.ready:
            raise RuntimeError("weights not loaded: call load_state_dict first")
        if sid is None:
            raise ValueError("sid is required (the reference's EnergyPredictor needs g; n_speakers > 0)")
        B, Tp = phonemes.shape

        def split(ctl):
            if isinstance(ctl, torch.Tensor):      # the reference's isinstance(.., torch.Tensor) branches
                return ctl, 1.0
            return None, 1.0 if ctl is None else float(ctl)

        d_t, d_s = split(duration_control)
        p_t, p_s = split(pitch_control)
        e_t, e_s = split(energy_control)
        enc = eng.encode(phonemes, phonemes_lengths, sid, d_t, p_t, e_t, d_s, p_s, e_s)
        # (a known padding: the second half's tensors are allocated while the GPU is still busy with the first)
        bufs = eng.decode_buffers(B, Tp, int(t_f), max_len) if t_f is not None and int(t_f) > 0 else None
        _, tf_local = eng.frame_lengths_host(enc["frame_lengths"])
        Tf = tf_local if t_f is None else max(int(t_f), tf_local)
        if Tf <= 0:
            raise ValueError("all durations are zero: nothing to synthesise")
        ns = float(noise_scale)
        if noise is None and ns != 0.0 and noise_seed is None:
            noise = torch.randn(B, self.dims.inter_channels, Tf, dtype=torch.float32, device=eng.device)
        dec = eng.decode(enc, Tf, noise, ns, max_len, noise_seed=0 if noise_seed is None else int(noise_seed), bufs=bufs,
                         noise_offset=int(noise_offset))
        duration = duration_control if d_t is not None else enc["duration"].view(B, 1, Tp)
        return (dec["o"], dec["x_mask"], (dec["z"], dec["z_p"], dec["m_p"], dec["logs_p"]), duration, enc["F0"],
                enc["energy"])
