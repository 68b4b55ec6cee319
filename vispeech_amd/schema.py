"""state_dict wire format of the synthesizer.

The reference has no plugin/FFI layer; the checkpoint key schema *is* the weight
wire format.  This module derives every key and shape
from the constructor hyper-parameters so that (a) a reference ``G_*.pth``
loads unchanged and (b) synthetic weights can be generated without importing
the reference.  Each block cites the reference constructor it mirrors.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

Shape = Tuple[int, ...]


@dataclass
class ModelDims:
    """Hyper-parameters that determine tensor shapes (reference models.py:537-561)."""

    n_vocab: int = 519
    spec_channels: int = 1025
    hop_length: int = 512
    sampling_rate: int = 44100
    segment_size: int = 32
    inter_channels: int = 192
    hidden_channels: int = 192
    filter_channels: int = 768
    n_heads: int = 2
    n_layers: int = 4
    kernel_size: int = 3
    p_dropout: float = 0.1
    resblock: str = "1"
    resblock_kernel_sizes: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: List[List[int]] = field(
        default_factory=lambda: [[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    upsample_rates: List[int] = field(default_factory=lambda: [8, 8, 4, 2])
    upsample_initial_channel: int = 512
    upsample_kernel_sizes: List[int] = field(default_factory=lambda: [16, 16, 4, 4])
    n_speakers: int = 200
    gin_channels: int = 256
    # fixed by the reference code, not by the config file:
    window_size: int = 4            # attentions.py:14 (Encoder default)
    pitch_layers: int = 6           # models.py:498
    dur_filter: int = 256           # models.py:599
    energy_filter: int = 768        # frame_prior_network.py:65
    flow_kernel: int = 5            # models.py:597
    flow_layers: int = 4            # models.py:597 (WN n_layers)
    n_flows: int = 4                # models.py:184
    posterior_layers: int = 16      # models.py:595

    @property
    def total_upsample(self) -> int:
        p = 1
        for u in self.upsample_rates:
            p *= u
        return p


def _encoder(prefix: str, n_layers: int, d: ModelDims, out: "OrderedDict[str, Shape]") -> None:
    """attentions.Encoder (reference attentions.py:13-33, 101-127, 257-275)."""
    h, f, k = d.hidden_channels, d.filter_channels, d.kernel_size
    dk = h // d.n_heads
    for i in range(n_layers):
        a = f"{prefix}.attn_layers.{i}"
        out[f"{a}.emb_rel_k"] = (1, 2 * d.window_size + 1, dk)
        out[f"{a}.emb_rel_v"] = (1, 2 * d.window_size + 1, dk)
        for nm in ("conv_q", "conv_k", "conv_v", "conv_o"):
            out[f"{a}.{nm}.weight"] = (h, h, 1)
            out[f"{a}.{nm}.bias"] = (h,)
    for i in range(n_layers):
        out[f"{prefix}.norm_layers_1.{i}.gamma"] = (h,)
        out[f"{prefix}.norm_layers_1.{i}.beta"] = (h,)
    for i in range(n_layers):
        p = f"{prefix}.ffn_layers.{i}"
        out[f"{p}.conv_1.weight"] = (f, h, k)
        out[f"{p}.conv_1.bias"] = (f,)
        out[f"{p}.conv_2.weight"] = (h, f, k)
        out[f"{p}.conv_2.bias"] = (h,)
    for i in range(n_layers):
        out[f"{prefix}.norm_layers_2.{i}.gamma"] = (h,)
        out[f"{prefix}.norm_layers_2.{i}.beta"] = (h,)


def _wn(prefix: str, hidden: int, kernel: int, n_layers: int, gin: int,
        out: "OrderedDict[str, Shape]") -> None:
    """modules.WN with weight-norm wrappers (reference modules.py:111-146)."""
    for i in range(n_layers):
        out[f"{prefix}.in_layers.{i}.bias"] = (2 * hidden,)
        out[f"{prefix}.in_layers.{i}.weight_g"] = (2 * hidden, 1, 1)
        out[f"{prefix}.in_layers.{i}.weight_v"] = (2 * hidden, hidden, kernel)
    for i in range(n_layers):
        rs = 2 * hidden if i < n_layers - 1 else hidden
        out[f"{prefix}.res_skip_layers.{i}.bias"] = (rs,)
        out[f"{prefix}.res_skip_layers.{i}.weight_g"] = (rs, 1, 1)
        out[f"{prefix}.res_skip_layers.{i}.weight_v"] = (rs, hidden, 1)
    if gin:
        out[f"{prefix}.cond_layer.bias"] = (2 * hidden * n_layers,)
        out[f"{prefix}.cond_layer.weight_g"] = (2 * hidden * n_layers, 1, 1)
        out[f"{prefix}.cond_layer.weight_v"] = (2 * hidden * n_layers, gin, 1)


def state_dict_schema(d: ModelDims) -> "OrderedDict[str, Shape]":
    """All checkpoint tensors of ``SynthesizerTrn`` (reference models.py:537-622)."""
    out: "OrderedDict[str, Shape]" = OrderedDict()
    h, gin = d.hidden_channels, d.gin_channels

    # enc_p: TextEncoder (models.py:136-166)
    out["enc_p.symbol_emb.weight"] = (d.n_vocab, h)
    _encoder("enc_p.encoder", d.n_layers, d, out)
    out["enc_p.proj.weight"] = (2 * d.inter_channels, h, 1)
    out["enc_p.proj.bias"] = (2 * d.inter_channels,)

    # dec: Generator (models.py:244-269), ResBlock1 (modules.py:187-206)
    c0 = d.upsample_initial_channel
    out["dec.conv_pre.weight"] = (c0, d.inter_channels, 7)
    out["dec.conv_pre.bias"] = (c0,)
    for i, (u, k) in enumerate(zip(d.upsample_rates, d.upsample_kernel_sizes)):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        out[f"dec.ups.{i}.bias"] = (cout,)
        out[f"dec.ups.{i}.weight_g"] = (cin, 1, 1)          # ConvTranspose: norm per INPUT channel
        out[f"dec.ups.{i}.weight_v"] = (cin, cout, k)
    nk = len(d.resblock_kernel_sizes)
    ch = c0
    for i in range(len(d.upsample_rates)):
        ch = c0 // (2 ** (i + 1))
        for j, (k, dil) in enumerate(zip(d.resblock_kernel_sizes, d.resblock_dilation_sizes)):
            p = f"dec.resblocks.{i * nk + j}"
            for grp in ("convs1", "convs2"):
                for m in range(len(dil)):
                    out[f"{p}.{grp}.{m}.bias"] = (ch,)
                    out[f"{p}.{grp}.{m}.weight_g"] = (ch, 1, 1)
                    out[f"{p}.{grp}.{m}.weight_v"] = (ch, ch, k)
    out["dec.conv_post.weight"] = (1, ch, 7)
    if gin:
        out["dec.cond.weight"] = (c0, gin, 1)
        out["dec.cond.bias"] = (c0,)

    # enc_q: PosteriorEncoder (models.py:212-232) -- present in checkpoints, unused by infer
    out["enc_q.pre.weight"] = (h, d.spec_channels, 1)
    out["enc_q.pre.bias"] = (h,)
    _wn("enc_q.enc", h, 5, d.posterior_layers, gin, out)
    out["enc_q.proj.weight"] = (2 * d.inter_channels, h, 1)
    out["enc_q.proj.bias"] = (2 * d.inter_channels,)

    # flow: ResidualCouplingBlock (models.py:177-200), layers at even indices (odd = Flip)
    half = d.inter_channels // 2
    for i in range(d.n_flows):
        p = f"flow.flows.{2 * i}"
        out[f"{p}.pre.weight"] = (h, half, 1)
        out[f"{p}.pre.bias"] = (h,)
        _wn(f"{p}.enc", h, d.flow_kernel, d.flow_layers, gin, out)
        out[f"{p}.post.weight"] = (half, h, 1)               # mean_only=True
        out[f"{p}.post.bias"] = (half,)

    # duration_predictor (models.py:99-117)
    f = d.dur_filter
    out["duration_predictor.conv_1.weight"] = (f, h, 3)
    out["duration_predictor.conv_1.bias"] = (f,)
    out["duration_predictor.norm_1.gamma"] = (f,)
    out["duration_predictor.norm_1.beta"] = (f,)
    out["duration_predictor.conv_2.weight"] = (f, f, 3)
    out["duration_predictor.conv_2.bias"] = (f,)
    out["duration_predictor.norm_2.gamma"] = (f,)
    out["duration_predictor.norm_2.beta"] = (f,)
    out["duration_predictor.proj.weight"] = (1, f, 1)
    out["duration_predictor.proj.bias"] = (1,)
    if gin:
        out["duration_predictor.cond.weight"] = (h, gin, 1)
        out["duration_predictor.cond.bias"] = (h,)

    # frame_prior_net (models.py:435-464)
    out["frame_prior_net.emb.weight"] = (121, h)
    _encoder("frame_prior_net.fft_block", d.n_layers, d, out)

    # pitch_predictor (models.py:473-503)
    _encoder("pitch_predictor.pitch_net", d.pitch_layers, d, out)
    out["pitch_predictor.proj_f0.weight"] = (1, h, 1)
    out["pitch_predictor.proj_f0.bias"] = (1,)
    if gin:
        out["pitch_predictor.cond.weight"] = (h, gin, 1)
        out["pitch_predictor.cond.bias"] = (h,)

    # energy_predictor (frame_prior_network.py:58-118)
    e = d.energy_filter
    p = "energy_predictor.predictor"
    out[f"{p}.conv_layer.conv_1.conv.weight"] = (e, h, 3)
    out[f"{p}.conv_layer.conv_1.conv.bias"] = (e,)
    out[f"{p}.conv_layer.layer_norm_1.weight"] = (e,)
    out[f"{p}.conv_layer.layer_norm_1.bias"] = (e,)
    out[f"{p}.conv_layer.conv_2.conv.weight"] = (e, e, 3)
    out[f"{p}.conv_layer.conv_2.conv.bias"] = (e,)
    out[f"{p}.conv_layer.layer_norm_2.weight"] = (e,)
    out[f"{p}.conv_layer.layer_norm_2.bias"] = (e,)
    out[f"{p}.linear_layer.weight"] = (1, e)
    out[f"{p}.linear_layer.bias"] = (1,)
    out[f"{p}.proj.weight"] = (h, 1)
    out[f"{p}.proj.bias"] = (h,)
    if gin:
        out["energy_predictor.cond.weight"] = (h, gin, 1)
        out["energy_predictor.cond.bias"] = (h,)

    # project + prenets + speaker table (models.py:517-524, 612-616)
    out["project.proj.weight"] = (2 * d.inter_channels, h, 1)
    out["project.proj.bias"] = (2 * d.inter_channels,)
    out["pitch_prenet.weight"] = (h, 1, 3)
    out["pitch_prenet.bias"] = (h,)
    out["energy_prenet.weight"] = (h, 1, 3)
    out["energy_prenet.bias"] = (h,)
    if d.n_speakers > 1:
        out["emb_g.weight"] = (d.n_speakers, gin)
    return out


# Keys that exist in checkpoints but that ``infer`` never reads (SURVEY gotcha G11).
_UNUSED_PREFIXES = ("enc_q.", "enc_p.proj.", "frame_prior_net.emb.",
                    "energy_predictor.predictor.proj.")


def used_by_infer(key: str) -> bool:
    return not key.startswith(_UNUSED_PREFIXES)


def infer_schema(d: ModelDims) -> "OrderedDict[str, Shape]":
    return OrderedDict((k, s) for k, s in state_dict_schema(d).items() if used_by_infer(k))


def param_count(schema: Dict[str, Shape]) -> int:
    n = 0
    for s in schema.values():
        c = 1
        for x in s:
            c *= x
        n += c
    return n


def dims_from_ctor(n_vocab, spec_channels, hop_length, sampling_rate, segment_size,
                   inter_channels, hidden_channels, filter_channels, n_heads, n_layers,
                   kernel_size, p_dropout, resblock, resblock_kernel_sizes,
                   resblock_dilation_sizes, upsample_rates, upsample_initial_channel,
                   upsample_kernel_sizes, n_speakers=0, gin_channels=0, **_ignored) -> ModelDims:
    """Map the reference constructor signature (models.py:537-561) onto ModelDims."""
    return ModelDims(
        n_vocab=n_vocab, spec_channels=spec_channels, hop_length=hop_length,
        sampling_rate=sampling_rate, segment_size=segment_size,
        inter_channels=inter_channels, hidden_channels=hidden_channels,
        filter_channels=filter_channels, n_heads=n_heads, n_layers=n_layers,
        kernel_size=kernel_size, p_dropout=p_dropout, resblock=str(resblock),
        resblock_kernel_sizes=list(resblock_kernel_sizes),
        resblock_dilation_sizes=[list(x) for x in resblock_dilation_sizes],
        upsample_rates=list(upsample_rates),
        upsample_initial_channel=upsample_initial_channel,
        upsample_kernel_sizes=list(upsample_kernel_sizes),
        n_speakers=n_speakers, gin_channels=gin_channels)
