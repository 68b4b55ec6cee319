"""ORACLE -- test infrastructure, not product code.

CPU (torch fp32/fp64, functional) restatement of the reference's
``SynthesizerTrn.infer`` hot path, of ``voice_conversion`` (posterior encoder +
forward flow), of its rational-quadratic spline and of the linear spectrogram.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``vispeech_amd`` never does.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the pin is ``tests/golden/*.npz`` -- outputs of the
real reference (imported from /root/reference in the build container by
``tests/golden/make_golden.py``) on seeded synthetic weights and inputs;
``tests/test_oracle_golden.py`` checks this restatement against every stored
stage boundary (infer: 5 cases, voice_conversion: 1, spline: 1).  ``spectrogram``
is the exception: the reference's ``torch.stft`` call form is rejected by the
installed torch, so that function is pinned against ``torch.stft(..., return_complex=True)`` called
argument for argument and a float64 DFT, not against a reference run; ``mel_filterbank`` (librosa absent) is pinned
against transformers' implementation of librosa's routine.

It is written independently of the reference's module classes: weight-norm is
folded once, relative-position attention uses the closed banded form instead of
the pad/reshape skewing, the length regulator is a prefix sum + searchsorted
gather, and the reparameterisation noise is an explicit argument.  Each
function cites the reference lines whose arithmetic it restates.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # reference modules.py:17


# --------------------------------------------------------------------------- weights
def fold_weight_norm(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``w = g * v / ||v||`` with the norm over every axis but 0
    (torch.nn.utils.weight_norm default dim=0; call sites reference modules.py:128,135,145,
    191-206 and models.py:255 -- for ConvTranspose1d's [in,out,k] weight that is per INPUT
    channel).  Returns a dict with ``*.weight`` in place of ``*.weight_g``/``*.weight_v``."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            g = sd[k[:-1] + "g"]
            vv = v.double()
            nrm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape(g.shape)
            out[k[:-2]] = (vv * (g.double() / nrm)).to(v.dtype)
        else:
            out[k] = v
    return out


def _as_tensors(sd, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: (torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v).to(dtype)
            for k, v in sd.items()}


# --------------------------------------------------------------------------- primitives
def sequence_mask(lengths: torch.Tensor, max_len: int) -> torch.Tensor:
    """reference commons.py:121-125"""
    return torch.arange(max_len, dtype=lengths.dtype)[None, :] < lengths[:, None]


def layer_norm_ct(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """LayerNorm over the channel axis of [B,C,T] (reference modules.py:29-32)."""
    mean = x.mean(dim=1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma[None, :, None] + beta[None, :, None]


def rel_attention(q, k, v, emb_k, emb_v, mask_q, n_heads: int, window: int):
    """Windowed relative-position self-attention, closed form of reference
    attentions.py:148-179 (+ helpers :181-243).

    q,k,v [B,C,T]; emb_k/emb_v [1,2w+1,dk] shared over heads; mask_q [B,T] (1 = valid).
    S_ij = (q_i/sqrt(dk)).k_j + [|j-i|<=w] (q_i/sqrt(dk)).Ek[j-i+w]; masked (mask_i*mask_j==0)
    -> -1e4; P = softmax_j; O_i = sum_j P_ij v_j + sum_{|j-i|<=w} P_ij Ev[j-i+w].
    """
    b, c, t = q.shape
    dk = c // n_heads
    qh = q.view(b, n_heads, dk, t).transpose(2, 3) / math.sqrt(dk)      # [B,H,T,dk]
    kh = k.view(b, n_heads, dk, t).transpose(2, 3)
    vh = v.view(b, n_heads, dk, t).transpose(2, 3)
    scores = qh @ kh.transpose(-2, -1)                                   # [B,H,T,T]
    rel = qh @ emb_k[0].t()                                              # [B,H,T,2w+1]
    idx = torch.arange(t)
    off = idx[None, :] - idx[:, None]                                    # j - i
    band = off.abs() <= window
    gather = (off + window).clamp(0, 2 * window)
    band_logits = torch.gather(rel, 3, gather.expand(b, n_heads, t, t))
    scores = scores + torch.where(band, band_logits, torch.zeros((), dtype=q.dtype))
    m = mask_q.to(q.dtype)
    pair = m[:, None, :, None] * m[:, None, None, :]
    scores = scores.masked_fill(pair == 0, -1e4)
    p = torch.softmax(scores, dim=-1)
    out = p @ vh                                                         # [B,H,T,dk]
    # relative-value term: weights of the band, one column per offset
    pw = torch.zeros(b, n_heads, t, 2 * window + 1, dtype=q.dtype)
    for r in range(-window, window + 1):
        lo, hi = max(0, -r), min(t, t - r)
        if hi > lo:
            ii = torch.arange(lo, hi)
            pw[:, :, ii, r + window] = p[:, :, ii, ii + r]
    out = out + pw @ emb_v[0]
    return out.transpose(2, 3).reshape(b, c, t)


def encoder(w: Mapping[str, torch.Tensor], prefix: str, n_layers: int, x, mask, n_heads: int,
            window: int, kernel_size: int):
    """attentions.Encoder.forward (reference attentions.py:35-47), FFN (:277-285)."""
    m = mask.to(x.dtype)[:, None, :]
    x = x * m
    pl, pr = (kernel_size - 1) // 2, kernel_size // 2
    for i in range(n_layers):
        a = f"{prefix}.attn_layers.{i}"
        q = F.conv1d(x, w[f"{a}.conv_q.weight"], w[f"{a}.conv_q.bias"])
        k = F.conv1d(x, w[f"{a}.conv_k.weight"], w[f"{a}.conv_k.bias"])
        v = F.conv1d(x, w[f"{a}.conv_v.weight"], w[f"{a}.conv_v.bias"])
        y = rel_attention(q, k, v, w[f"{a}.emb_rel_k"], w[f"{a}.emb_rel_v"], mask, n_heads, window)
        y = F.conv1d(y, w[f"{a}.conv_o.weight"], w[f"{a}.conv_o.bias"])
        x = layer_norm_ct(x + y, w[f"{prefix}.norm_layers_1.{i}.gamma"], w[f"{prefix}.norm_layers_1.{i}.beta"])
        f = f"{prefix}.ffn_layers.{i}"
        y = F.conv1d(F.pad(x * m, (pl, pr)), w[f"{f}.conv_1.weight"], w[f"{f}.conv_1.bias"])
        y = torch.relu(y)
        y = F.conv1d(F.pad(y * m, (pl, pr)), w[f"{f}.conv_2.weight"], w[f"{f}.conv_2.bias"]) * m
        x = layer_norm_ct(x + y, w[f"{prefix}.norm_layers_2.{i}.gamma"], w[f"{prefix}.norm_layers_2.{i}.beta"])
    return x * m


def duration_predictor(w, x, mask, g):
    """reference models.py:119-133"""
    m = mask.to(x.dtype)[:, None, :]
    p = "duration_predictor"
    x = x + F.conv1d(g, w[f"{p}.cond.weight"], w[f"{p}.cond.bias"])
    x = F.conv1d(x * m, w[f"{p}.conv_1.weight"], w[f"{p}.conv_1.bias"], padding=1)
    x = layer_norm_ct(torch.relu(x), w[f"{p}.norm_1.gamma"], w[f"{p}.norm_1.beta"])
    x = F.conv1d(x * m, w[f"{p}.conv_2.weight"], w[f"{p}.conv_2.bias"], padding=1)
    x = layer_norm_ct(torch.relu(x), w[f"{p}.norm_2.gamma"], w[f"{p}.norm_2.beta"])
    x = F.conv1d(x * m, w[f"{p}.proj.weight"], w[f"{p}.proj.bias"])
    return x * m                                                        # [B,1,Tp]


def pitch_predictor(w, x, mask, g, dims):
    """reference models.py:505-514"""
    m = mask.to(x.dtype)[:, None, :]
    p = "pitch_predictor"
    x = x + F.conv1d(g, w[f"{p}.cond.weight"], w[f"{p}.cond.bias"])
    x = encoder(w, f"{p}.pitch_net", dims.pitch_layers, x * m, mask, dims.n_heads,
                dims.window_size, dims.kernel_size)
    x = x * m
    return F.conv1d(x, w[f"{p}.proj_f0.weight"], w[f"{p}.proj_f0.bias"]).squeeze(1)


def energy_predictor(w, x, g):
    """reference frame_prior_network.py:104-124 -- note: no mask anywhere."""
    p = "energy_predictor"
    q = f"{p}.predictor"
    x = x + F.conv1d(g, w[f"{p}.cond.weight"], w[f"{p}.cond.bias"])
    x = F.conv1d(x, w[f"{q}.conv_layer.conv_1.conv.weight"], w[f"{q}.conv_layer.conv_1.conv.bias"], padding=1)
    x = layer_norm_ct(torch.relu(x), w[f"{q}.conv_layer.layer_norm_1.weight"], w[f"{q}.conv_layer.layer_norm_1.bias"])
    x = F.conv1d(x, w[f"{q}.conv_layer.conv_2.conv.weight"], w[f"{q}.conv_layer.conv_2.conv.bias"], padding=1)
    x = layer_norm_ct(torch.relu(x), w[f"{q}.conv_layer.layer_norm_2.weight"], w[f"{q}.conv_layer.layer_norm_2.bias"])
    wl = w[f"{q}.linear_layer.weight"]                                   # [1,E]
    return (x * wl[0][None, :, None]).sum(dim=1) + w[f"{q}.linear_layer.bias"]


def length_regulate(x, duration):
    """reference models.py:398-427: repeat phoneme i ``max(int(d_i),0)`` times, zero-pad to the
    longest utterance.  Restated as cumsum + searchsorted gather.  Returns ([B,C,Tf], lengths)."""
    b, c, tp = x.shape
    d = duration.reshape(b, -1)[:, :tp]
    reps = d.to(torch.float64).trunc().clamp(min=0).to(torch.int64)     # int() truncation, clamp
    cum = reps.cumsum(dim=1)
    lens = cum[:, -1]
    tf = int(lens.max().item()) if b > 0 else 0
    frames = torch.arange(tf)
    idx = torch.searchsorted(cum, frames[None, :].expand(b, tf).contiguous(), right=True)
    valid = frames[None, :] < lens[:, None]
    idx = idx.clamp(max=tp - 1)
    out = torch.gather(x, 2, idx[:, None, :].expand(b, c, tf))
    out = out * valid[:, None, :].to(x.dtype)
    return out, lens


def wn_layer(w, prefix: str, i: int, x, out, mask_f, gc, hidden: int, n_layers: int, kernel: int):
    """One iteration of the loop of modules.WN.forward (reference modules.py:157-175): returns (x, out) after layer
    ``i``; ``gc`` = cond_layer(g) (all layers' rows).  The final ``output * x_mask`` (modules.py:176) is NOT applied."""
    a = F.conv1d(x, w[f"{prefix}.in_layers.{i}.weight"], w[f"{prefix}.in_layers.{i}.bias"], padding=(kernel - 1) // 2)
    a = a + gc[:, i * 2 * hidden:(i + 1) * 2 * hidden]
    acts = torch.tanh(a[:, :hidden]) * torch.sigmoid(a[:, hidden:])
    rs = F.conv1d(acts, w[f"{prefix}.res_skip_layers.{i}.weight"], w[f"{prefix}.res_skip_layers.{i}.bias"])
    if i < n_layers - 1:
        return (x + rs[:, :hidden]) * mask_f, out + rs[:, hidden:]
    return x, out + rs


def wn(w, prefix: str, x, mask_f, g, hidden: int, n_layers: int, kernel: int):
    """modules.WN.forward with dilation_rate 1 (reference modules.py:148-176) and the fused gate
    (reference commons.py:100-107)."""
    out = torch.zeros_like(x)
    gc = F.conv1d(g, w[f"{prefix}.cond_layer.weight"], w[f"{prefix}.cond_layer.bias"])
    for i in range(n_layers):
        x, out = wn_layer(w, prefix, i, x, out, mask_f, gc, hidden, n_layers, kernel)
    return out * mask_f


def flow_reverse(w, z, mask_f, g, dims):
    """ResidualCouplingBlock.forward(reverse=True) (reference models.py:202-209) over
    ResidualCouplingLayer (modules.py:324-343, mean_only) and Flip (modules.py:270-277)."""
    half = dims.inter_channels // 2
    x = z
    for i in reversed(range(dims.n_flows)):
        x = torch.flip(x, [1])
        p = f"flow.flows.{2 * i}"
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, w[f"{p}.pre.weight"], w[f"{p}.pre.bias"]) * mask_f
        h = wn(w, f"{p}.enc", h, mask_f, g, dims.hidden_channels, dims.flow_layers, dims.flow_kernel)
        m = F.conv1d(h, w[f"{p}.post.weight"], w[f"{p}.post.bias"]) * mask_f
        x1 = (x1 - m) * mask_f
        x = torch.cat([x0, x1], dim=1)
    return x


def flow_forward(w, z, mask_f, g, dims):
    """ResidualCouplingBlock.forward(reverse=False) (reference models.py:202-206): coupling
    (modules.py:324-340, mean_only so logs = 0) then Flip (modules.py:270-274), layers 0 .. n-1."""
    half = dims.inter_channels // 2
    x = z
    for i in range(dims.n_flows):
        p = f"flow.flows.{2 * i}"
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, w[f"{p}.pre.weight"], w[f"{p}.pre.bias"]) * mask_f
        h = wn(w, f"{p}.enc", h, mask_f, g, dims.hidden_channels, dims.flow_layers, dims.flow_kernel)
        m = F.conv1d(h, w[f"{p}.post.weight"], w[f"{p}.post.bias"]) * mask_f
        x1 = m + x1 * mask_f
        x = torch.flip(torch.cat([x0, x1], dim=1), [1])
    return x


def posterior_encoder(w, y, mask_f, g, noise, dims):
    """PosteriorEncoder.forward (reference models.py:233-241): returns z, m, logs."""
    x = F.conv1d(y, w["enc_q.pre.weight"], w["enc_q.pre.bias"]) * mask_f
    x = wn(w, "enc_q.enc", x, mask_f, g, dims.hidden_channels, dims.posterior_layers, dims.flow_kernel)
    stats = F.conv1d(x, w["enc_q.proj.weight"], w["enc_q.proj.bias"]) * mask_f
    m, logs = stats[:, :dims.inter_channels], stats[:, dims.inter_channels:]
    z = (m + noise * torch.exp(logs)) * mask_f
    return z, m, logs


def generator(w, x, g, dims):
    """Generator.forward (reference models.py:271-290) with ResBlock1 (modules.py:210-223)."""
    x = F.conv1d(x, w["dec.conv_pre.weight"], w["dec.conv_pre.bias"], padding=3)
    x = x + F.conv1d(g, w["dec.cond.weight"], w["dec.cond.bias"])
    nk = len(dims.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(dims.upsample_rates, dims.upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, w[f"dec.ups.{i}.weight"], w[f"dec.ups.{i}.bias"], stride=u,
                               padding=(k - u) // 2)
        xs = None
        for j, (rk, dil) in enumerate(zip(dims.resblock_kernel_sizes, dims.resblock_dilation_sizes)):
            p = f"dec.resblocks.{i * nk + j}"
            y = x
            for mth, dd in enumerate(dil):
                t = F.leaky_relu(y, LRELU_SLOPE)
                t = F.conv1d(t, w[f"{p}.convs1.{mth}.weight"], w[f"{p}.convs1.{mth}.bias"],
                             dilation=dd, padding=(rk * dd - dd) // 2)
                t = F.leaky_relu(t, LRELU_SLOPE)
                t = F.conv1d(t, w[f"{p}.convs2.{mth}.weight"], w[f"{p}.convs2.{mth}.bias"],
                             padding=(rk - 1) // 2)
                y = t + y
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x, 0.01)                       # F.leaky_relu default slope (models.py:286)
    x = F.conv1d(x, w["dec.conv_post.weight"], None, padding=3)
    return torch.tanh(x)


# --------------------------------------------------------------------------- the path
class Oracle:
    """Holds folded weights; ``infer`` mirrors reference models.py:672-722 and returns every
    stage boundary in a dict."""

    def __init__(self, state_dict: Mapping[str, "np.ndarray | torch.Tensor"], dims, dtype=torch.float32):
        self.dims = dims
        self.dtype = dtype
        self.w = fold_weight_norm(_as_tensors(state_dict, dtype))

    @torch.no_grad()
    def encode(self, phonemes, lengths, sid, duration_control=None, pitch_control=None,
               energy_control=None) -> Dict[str, torch.Tensor]:
        w, d = self.w, self.dims
        phonemes = torch.as_tensor(phonemes, dtype=torch.int64)
        lengths = torch.as_tensor(lengths, dtype=torch.int64)
        sid = torch.as_tensor(sid, dtype=torch.int64)
        b, tp = phonemes.shape
        g = w["emb_g.weight"][sid][:, :, None]                               # [B,gin,1]
        mask = sequence_mask(lengths, tp)
        m = mask.to(self.dtype)[:, None, :]
        x = (w["enc_p.symbol_emb.weight"][phonemes] * math.sqrt(d.hidden_channels)).transpose(1, 2)
        x = encoder(w, "enc_p.encoder", d.n_layers, x * m, mask, d.n_heads, d.window_size, d.kernel_size)
        res = {"g": g, "x_enc": x.clone()}

        if torch.is_tensor(duration_control) or isinstance(duration_control, np.ndarray):
            duration = torch.as_tensor(duration_control, dtype=self.dtype)
        else:
            dc = 1 if duration_control is None else duration_control
            logw = duration_predictor(w, x, mask, g)
            res["logw"] = logw
            duration = torch.ceil((torch.exp(logw) * m - 1) * dc)

        if torch.is_tensor(pitch_control) or isinstance(pitch_control, np.ndarray):
            pc = torch.as_tensor(pitch_control, dtype=self.dtype)
            lf0 = (2595.0 * torch.log10(1.0 + pc / 700.0)) / 500
        else:
            pc = 1 if pitch_control is None else pitch_control
            lf0 = pitch_predictor(w, x, mask, g, d) * pc
        x = x + F.conv1d(lf0[:, None, :], w["pitch_prenet.weight"], w["pitch_prenet.bias"], padding=1)
        f0 = (torch.pow(torch.tensor(10.0, dtype=self.dtype), lf0 * 500 / 2590) - 1) * 700

        if torch.is_tensor(energy_control) or isinstance(energy_control, np.ndarray):
            ec = torch.as_tensor(energy_control, dtype=self.dtype)
            norm_energy = (ec - 60) / 36
        else:
            ec = 1 if energy_control is None else energy_control
            norm_energy = (((energy_predictor(w, x, g) * 36 + 60) * ec) - 60) / 36
        x = x + F.conv1d(norm_energy[:, None, :], w["energy_prenet.weight"], w["energy_prenet.bias"], padding=1)
        energy = norm_energy * 36 + 60

        x_frame, frame_lengths = length_regulate(x, duration)
        res.update(x_var=x, lf0=lf0, duration=duration, F0=f0, energy=energy,
                   x_frame=x_frame, frame_lengths=frame_lengths)
        return res

    @torch.no_grad()
    def decode(self, enc: Mapping[str, torch.Tensor], noise, noise_scale: float = 1.0,
               max_len: Optional[int] = None, t_f: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """``t_f`` pads the frame axis beyond the local maximum (global T_f of a sharded run,
        SURVEY.md gotcha G6)."""
        w, d = self.w, self.dims
        x_frame, lens, g = enc["x_frame"], enc["frame_lengths"], enc["g"]
        if t_f is not None and t_f > x_frame.shape[2]:
            x_frame = F.pad(x_frame, (0, t_f - x_frame.shape[2]))
        tf = x_frame.shape[2]
        mask = sequence_mask(lens, tf)
        m = mask.to(self.dtype)[:, None, :]
        h = encoder(w, "frame_prior_net.fft_block", d.n_layers, x_frame * m, mask, d.n_heads,
                    d.window_size, d.kernel_size)
        stats = F.conv1d(h, w["project.proj.weight"], w["project.proj.bias"]) * m
        m_p, logs_p = stats[:, :d.inter_channels], stats[:, d.inter_channels:]
        noise = torch.as_tensor(noise, dtype=self.dtype)
        z_p = m_p + noise * torch.exp(logs_p) * noise_scale
        z = flow_reverse(w, z_p, m, g, d)
        o = generator(w, (z * m)[:, :, :max_len], g, d)
        return dict(h_frame=h, m_p=m_p, logs_p=logs_p, z_p=z_p, z=z, o=o, x_mask=mask[:, None, :])

    @torch.no_grad()
    def infer(self, phonemes, lengths, sid, noise=None, noise_scale=1.0, max_len=None,
              energy_control=None, pitch_control=None, duration_control=None, t_f=None):
        enc = self.encode(phonemes, lengths, sid, duration_control, pitch_control, energy_control)
        tf = int(enc["frame_lengths"].max().item())
        if t_f is not None:
            tf = max(tf, t_f)
        if noise is None:
            noise = torch.randn(phonemes.shape[0], self.dims.inter_channels, tf, dtype=self.dtype)
        out = self.decode(enc, noise, noise_scale, max_len, t_f)
        out.update(enc)
        return out


# --------------------------------------------------------------------------- spline
def _vc(self, y, y_lengths, sid_src, sid_tgt, noise):
    """SynthesizerTrn.voice_conversion (reference models.py:724-732)."""
    w, d = self.w, self.dims
    y = torch.as_tensor(y, dtype=self.dtype)
    y_lengths = torch.as_tensor(y_lengths, dtype=torch.int64)
    g_src = w["emb_g.weight"][torch.as_tensor(sid_src, dtype=torch.int64)][:, :, None]
    g_tgt = w["emb_g.weight"][torch.as_tensor(sid_tgt, dtype=torch.int64)][:, :, None]
    mask = sequence_mask(y_lengths, y.shape[2])
    m = mask.to(self.dtype)[:, None, :]
    noise = torch.as_tensor(noise, dtype=self.dtype)
    z, m_q, logs_q = posterior_encoder(w, y, m, g_src, noise, d)
    z_p = flow_forward(w, z, m, g_src, d)
    z_hat = flow_reverse(w, z_p, m, g_tgt, d)
    o_hat = generator(w, z_hat * m, g_tgt, d)
    return dict(o_hat=o_hat, y_mask=m, z=z, z_p=z_p, z_hat=z_hat, m_q=m_q, logs_q=logs_q)


Oracle.voice_conversion = torch.no_grad()(_vc)


def spectrogram(y, n_fft: int, hop: int):
    """mel_processing.spectrogram_torch (reference mel_processing.py:50-69; win_size = n_fft, center=False):
    reflect padding of (n_fft - hop)/2, periodic Hann window, one-sided STFT, sqrt(|X|^2 + 1e-6).  The
    reference calls torch.stft without ``return_complex`` (rejected by torch >= 2.0), so this restatement
    uses the complex form of the same routine: same arithmetic, parity unpinned by a reference RUN."""
    y = torch.as_tensor(y, dtype=torch.float32)
    pad = int((n_fft - hop) / 2)
    yp = F.pad(y[:, None, :], (pad, pad), mode="reflect")[:, 0]
    spec = torch.stft(yp, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft), center=False,
                      normalized=False, onesided=True, return_complex=True)
    return torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-6)


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: Optional[float] = None) -> np.ndarray:
    """The Slaney-style mel filterbank that ``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`` returns with
    its defaults (htk=False, norm='slaney'), which reference mel_processing.py:79 calls.  librosa is not in this
    image and the reference pins no version: this restates the published algorithm (linear below 1 kHz at
    200/3 Hz per mel, log above with step ln(6.4)/27; triangles between successive mel points; area
    normalisation 2 / (f[m+2] - f[m])).  Pinned (round 5) to transformers.audio_utils.mel_filter_bank(norm='slaney',
    mel_scale='slaney') -- a third party's implementation adapted from librosa -- to fp32 rounding
    (tests/test_oracle_golden.py); used only as an audio-domain metric in tests."""
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.maximum(0.0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_spectrogram(y, sr: int, n_fft: int, hop: int, n_mels: int, fmin: float = 0.0, fmax: Optional[float] = None):
    """mel_processing.mel_spectrogram_torch (reference mel_processing.py:85-112): spectrogram -> mel basis ->
    log(clamp(x, 1e-5)) (dynamic range compression, mel_processing.py:16-22, 35-37)."""
    spec = spectrogram(y, n_fft, hop)
    mel = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax)) @ spec
    return torch.log(torch.clamp(mel, min=1e-5))


def rq_spline(inputs, uw, uh, ud, inverse=False, tail_bound=5.0, min_bin_width=1e-3,
              min_bin_height=1e-3, min_derivative=1e-3):
    """Unconstrained (linear-tail) monotone rational-quadratic spline of reference
    transforms.py:55-193 as called by ConvFlow (modules.py:380-386: tails='linear').
    Per element: ``uw``,``uh`` [...,nb]; ``ud`` [...,nb-1].  Returns (outputs, logabsdet).
    Elementwise restatement (no boolean-mask scatter): compute the spline for every element
    on inputs clamped into the interval and select the identity outside."""
    x = torch.as_tensor(inputs)
    uw, uh, ud = torch.as_tensor(uw), torch.as_tensor(uh), torch.as_tensor(ud)
    nb = uw.shape[-1]
    inside = (x >= -tail_bound) & (x <= tail_bound)
    xc = x.clamp(-tail_bound, tail_bound)
    const = math.log(math.exp(1 - min_derivative) - 1)                  # transforms.py:73
    ud = F.pad(ud, (1, 1), value=const)
    left = bottom = -tail_bound
    right = top = tail_bound

    def knots(u, min_sz, lo, hi):
        s = min_sz + (1 - min_sz * nb) * torch.softmax(u, dim=-1)
        c = F.pad(torch.cumsum(s, dim=-1), (1, 0), value=0.0)
        c = (hi - lo) * c + lo
        c[..., 0] = lo
        c[..., -1] = hi
        return c, c[..., 1:] - c[..., :-1]

    cw, widths = knots(uw, min_bin_width, left, right)
    ch, heights = knots(uh, min_bin_height, bottom, top)
    deriv = min_derivative + F.softplus(ud)
    loc = (ch if inverse else cw).clone()
    loc[..., -1] += 1e-6                                                 # transforms.py:48
    bin_idx = ((xc[..., None] >= loc).sum(dim=-1) - 1).clamp(0, nb - 1)[..., None]
    take = lambda t: t.gather(-1, bin_idx)[..., 0]
    in_cw, in_w, in_ch, in_h = take(cw), take(widths), take(ch), take(heights)
    delta = heights / widths
    in_delta, d0, d1 = take(delta), take(deriv), take(deriv[..., 1:])
    if inverse:
        dy = xc - in_ch
        s = d0 + d1 - 2 * in_delta
        a = dy * s + in_h * (in_delta - d0)
        bq = in_h * d0 - dy * s
        c = -in_delta * dy
        disc = bq * bq - 4 * a * c
        root = (2 * c) / (-bq - torch.sqrt(disc))
        out = root * in_w + in_cw
        tt = root * (1 - root)
        den = in_delta + s * tt
        num = in_delta ** 2 * (d1 * root ** 2 + 2 * in_delta * tt + d0 * (1 - root) ** 2)
        lad = -(torch.log(num) - 2 * torch.log(den))
    else:
        theta = (xc - in_cw) / in_w
        tt = theta * (1 - theta)
        s = d0 + d1 - 2 * in_delta
        den = in_delta + s * tt
        out = in_ch + in_h * (in_delta * theta ** 2 + d0 * tt) / den
        num = in_delta ** 2 * (d1 * theta ** 2 + 2 * in_delta * tt + d0 * (1 - theta) ** 2)
        lad = torch.log(num) - 2 * torch.log(den)
    out = torch.where(inside, out, x)
    lad = torch.where(inside, lad, torch.zeros_like(lad))
    return out, lad


# --------------------------------------------------------------------------- the library's own normal draws
def philox4x32_10(counter: np.ndarray, key0: int, key1: int) -> np.ndarray:
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
    reference implementation's constants).  counter [n, 4] uint32 -> [n, 4] uint32.  Known answer (Random123 kat_vectors):
    counter 0, key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8."""
    c = [counter[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key0 & 0xFFFFFFFF), np.uint64(key1 & 0xFFFFFFFF)
    m0, m1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = m0 * c[0], m1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack(c, axis=1).astype(np.uint32)


def philox_randn(seed: int, n: int, first: int = 0) -> np.ndarray:
    """What ``vsp_randn_at(seed, first)`` computes (vispeech_amd/csrc/misc.hip randn_kernel): stream element i =
    Box-Muller word i % 4 of counter (i // 4, 0, 0, 0), key = (seed lo, seed hi); uniforms from the top 23 bits,
    (x + 0.5) / 2^23 (exact in fp32, strictly inside (0, 1)); returns elements first .. first + n - 1."""
    g0 = first // 4
    groups = (first + n + 3) // 4 - g0
    ctr = np.zeros((groups, 4), dtype=np.uint32)
    q = np.arange(groups, dtype=np.uint64) + np.uint64(g0)
    ctr[:, 0] = (q & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (q >> np.uint64(32)).astype(np.uint32)
    w = philox4x32_10(ctr, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = ((w >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    out = np.empty((groups, 4), dtype=np.float32)
    for p in range(2):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[:, 2 * p]))
        ang = np.float32(6.283185307179586) * u[:, 2 * p + 1]
        out[:, 2 * p] = rad * np.cos(ang)
        out[:, 2 * p + 1] = rad * np.sin(ang)
    return out.reshape(-1)[first - 4 * g0: first - 4 * g0 + n]
