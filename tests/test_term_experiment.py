"""Gated experiment of VERDICT r1 item 9: can some vocoder convolutions run with TWO split-f16 products instead of
three and still meet the fp32 waveform gate (<= 1e-4 * max|ref|)?

CPU numerics only (no kernel involved): the generator of the oracle is restated here with the operand rounding each
scheme implies -- the f16 MFMA multiplies exactly and accumulates in fp32, so a scheme is fully described by what it
rounds:
  3 terms  xh*wh + xl*wh + xh*wl   nothing rounded (error 2^-22: the dropped xl*wl)      -> the product path
  2 terms  xh*wh + xl*wh           = x * f16(w): weights rounded to f16
  2 terms  xh*wh + xh*wl           = f16(x) * w: activations rounded to f16
  1 term   xh*wh                   both rounded                                            -> VSP_GENERATOR=f16
applied to all ResBlock / upsampling convolutions or to a subset (the k=11 ResBlocks carry 52 % of the generator's
FLOPs, the 32-channel stage 13 %).  Result (asserted below, synthetic weights of the golden cases): every 2-term variant lands ABOVE the gate -- 7e-4
everywhere, 1.7e-4 on the k=11 ResBlocks alone (17 % fewer MFMAs), 7e-5 when restricted further to the 128-channel
stage (6 % fewer MFMAs for a 1.4x margin on ONE set of weights), 3.5e-4 on the 32-channel stage -- while the product sits at 8e-7.  A mode that misses the gate, or would
clear it by a hair on one set of weights, is not adopted; VSP_GENERATOR=f16 stays the only (named, opt-in) reduced mode.
Reference call sites: models.py:271-290, modules.py:210-223."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.vispeech_oracle import LRELU_SLOPE, Oracle
from vispeech_amd.schema import ModelDims
from vispeech_amd.synth import synth_state_dict

WAVE_TOL = 1e-4


def f16(t):
    return t.to(torch.float16).to(torch.float32)


def generator_rounded(w, x, g, dims, policy):
    """oracle.generator with policy(stage, kernel) -> (round_x, round_w) for the upsampling (kernel = 0) and ResBlock
    convolutions; conv_pre / conv_post stay fp32 (they do in the product, too)."""
    def conv(fn, t, wt, b, st, k, **kw):
        rx, rw = policy(st, k)
        return fn(f16(t) if rx else t, f16(wt) if rw else wt, b, **kw)

    x = F.conv1d(x, w["dec.conv_pre.weight"], w["dec.conv_pre.bias"], padding=3)
    x = x + F.conv1d(g, w["dec.cond.weight"], w["dec.cond.bias"])
    nk = len(dims.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(dims.upsample_rates, dims.upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = conv(F.conv_transpose1d, x, w[f"dec.ups.{i}.weight"], w[f"dec.ups.{i}.bias"], i, 0, stride=u,
                 padding=(k - u) // 2)
        xs = None
        for j, (rk, dil) in enumerate(zip(dims.resblock_kernel_sizes, dims.resblock_dilation_sizes)):
            p = f"dec.resblocks.{i * nk + j}"
            y = x
            for mth, dd in enumerate(dil):
                t = F.leaky_relu(y, LRELU_SLOPE)
                t = conv(F.conv1d, t, w[f"{p}.convs1.{mth}.weight"], w[f"{p}.convs1.{mth}.bias"], i, rk,
                         dilation=dd, padding=(rk * dd - dd) // 2)
                t = F.leaky_relu(t, LRELU_SLOPE)
                t = conv(F.conv1d, t, w[f"{p}.convs2.{mth}.weight"], w[f"{p}.convs2.{mth}.bias"], i, rk,
                         padding=(rk - 1) // 2)
                y = t + y
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x, 0.01)
    x = F.conv1d(x, w["dec.conv_post.weight"], None, padding=3)
    return torch.tanh(x)


SCHEMES = {
    # name: (policy, fraction of the generator's MFMAs saved, must pass the gate?)
    "3_terms": (lambda st, k: (False, False), 0.0, True),
    "2_terms_weights_f16_everywhere": (lambda st, k: (False, True), 1 / 3, False),
    "2_terms_activations_f16_everywhere": (lambda st, k: (True, False), 1 / 3, False),
    "2_terms_weights_f16_on_k11_resblocks": (lambda st, k: (False, k == 11), 0.52 / 3, False),
    "2_terms_weights_f16_on_k11_resblocks_of_128_channel_stage": (lambda st, k: (False, k == 11 and st == 1), 0.52 / 9, None),
    "2_terms_weights_f16_on_32_channel_stage": (lambda st, k: (False, st == 3), 0.13 / 3, False),
    "1_term_everywhere": (lambda st, k: (True, True), 2 / 3, False),
}


@pytest.fixture(scope="module")
def setup(golden_dir):
    dims = ModelDims()
    orc = Oracle(synth_state_dict(dims, seed=1234, infer_only=True), dims)
    g = np.load(os.path.join(golden_dir, "ragged_controls.npz"))
    z = torch.from_numpy(g["z"])
    mask = torch.from_numpy(g["x_mask"]).to(torch.float32)
    gv = orc.w["emb_g.weight"][torch.from_numpy(g["in_sid"]).to(torch.int64)][:, :, None]
    return orc, dims, z * mask, gv, g["o"]


@pytest.mark.parametrize("name", list(SCHEMES))
def test_split_term_schemes_against_the_waveform_gate(setup, name):
    orc, dims, zin, gv, ref = setup
    policy, saved, must_pass = SCHEMES[name]
    with torch.no_grad():
        o = generator_rounded(orc.w, zin, gv, dims, policy).numpy().astype(np.float64)
    err = np.abs(o - ref).max() / np.abs(ref).max()
    print(f"{name}: max|o - ref| / max|ref| = {err:.2e}  (gate {WAVE_TOL:.0e}; saves {100 * saved:.0f} % of the generator's MFMAs)")
    if must_pass:
        assert err <= WAVE_TOL
    elif must_pass is None:
        # clears the gate by less than 2x on this one set of weights: no margin to adopt it on
        assert WAVE_TOL / 2 < err <= WAVE_TOL, (name, err)
    else:
        # recorded outcome of the experiment: NOT adoptable (the assertion fails loudly should that ever change)
        assert err > WAVE_TOL, (name, err)
