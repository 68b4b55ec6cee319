"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/vispeech_hip.h
declares, and its host logic (schema validation, arena planning, workspace sizing) behaves.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vispeech_amd import _lib
from vispeech_amd.schema import ModelDims, infer_schema, state_dict_schema, used_by_infer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vispeech_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vsp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes signature table and header disagree"
    assert lib.vsp_abi_version() == _lib.ABI_VERSION == 7


@pytest.fixture()
def ctx():
    lib = _lib.lib()
    cfg = _lib.make_config(ModelDims())
    h = C.c_void_p()
    assert lib.vsp_create(C.byref(cfg), 0, C.byref(h)) == 0
    yield lib, h
    lib.vsp_destroy(h)


def _set(lib, h, key, arr):
    a = np.ascontiguousarray(arr, dtype=np.float32)
    shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
    return lib.vsp_set_weight(h, key.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)


def test_schema_matches_python_schema(ctx):
    lib, h = ctx
    dims = ModelDims()
    schema = state_dict_schema(dims)
    n_used = sum(1 for k in schema if used_by_infer(k))
    assert lib.vsp_missing_weights(h) == n_used
    for k, shape in schema.items():
        assert _set(lib, h, k, np.zeros(shape, dtype=np.float32)) == 0, (k, lib.vsp_last_error(h))
    assert lib.vsp_missing_weights(h) == 0
    assert len(schema) == 753            # the reference's state_dict size


def test_posterior_encoder_tensors_are_optional_but_validated(ctx):
    """enc_q.* (voice conversion only) never counts as missing, but its shapes are checked when the
    config names spec_channels; a context without spec_channels accepts and ignores them."""
    lib, h = ctx
    dims = ModelDims()
    schema = state_dict_schema(dims)
    for k, shape in schema.items():
        if used_by_infer(k):
            assert _set(lib, h, k, np.zeros(shape, dtype=np.float32)) == 0
    assert lib.vsp_missing_weights(h) == 0                      # enc_q.* absent, nothing missing
    assert _set(lib, h, "enc_q.pre.weight", np.zeros(schema["enc_q.pre.weight"])) == 0
    assert _set(lib, h, "enc_q.pre.weight", np.zeros((192, 513, 1))) == -5
    assert _set(lib, h, "enc_q.nonexistent", np.zeros((1,))) == -4
    assert lib.vsp_has_voice_conversion(h) == 0                 # nothing finalised
    cfg = _lib.make_config(dims)
    cfg.spec_channels = 0
    h2 = C.c_void_p()
    assert lib.vsp_create(C.byref(cfg), 0, C.byref(h2)) == 0
    assert _set(lib, h2, "enc_q.pre.weight", np.zeros((192, 513, 1))) == 0     # ignored
    assert lib.vsp_voice_conversion_workspace_bytes(h2, 1, 16) > 0
    assert lib.vsp_weight_arena_bytes(h2) < lib.vsp_weight_arena_bytes(h)
    lib.vsp_destroy(h2)


def test_bad_key_and_shape_are_rejected(ctx):
    lib, h = ctx
    assert _set(lib, h, "dec.nonexistent.weight", np.zeros((1,))) == -4
    assert b"unknown" in lib.vsp_last_error(h)
    assert _set(lib, h, "dec.conv_pre.bias", np.zeros((7,))) == -5
    # pre-folded weight accepted where the schema has weight_v
    assert _set(lib, h, "dec.ups.0.weight", np.zeros((512, 256, 16))) == 0


def test_arena_and_workspace_sizes(ctx):
    lib, h = ctx
    arena = lib.vsp_weight_arena_bytes(h)
    n_params = sum(int(np.prod(s)) for k, s in state_dict_schema(ModelDims()).items()
                   if used_by_infer(k) or k.startswith("enc_q."))
    # packed arena holds every infer-path parameter and the posterior encoder (weight_g folded away, some zero padding)
    assert 0.9 * 4 * n_params < arena < 2.0 * 4 * n_params   # generator weights are held in several packings
    e1, e2 = lib.vsp_encode_workspace_bytes(h, 2, 40), lib.vsp_encode_workspace_bytes(h, 4, 40)
    assert 0 < e1 < e2
    d1, d2 = lib.vsp_decode_workspace_bytes(h, 1, 40, 100), lib.vsp_decode_workspace_bytes(h, 1, 40, 200)
    assert 0 < d1 < d2
    # generator dominates: 5 buffers of B * max_stage(C*T) floats
    g = lib.vsp_generator_workspace_bytes(h, 1, 100)
    assert g >= 5 * 4 * 32 * 100 * 512
    assert lib.vsp_decode_workspace_bytes(h, 0, 40, 100) < 0


def test_calls_before_finalize_fail_loudly(ctx):
    lib, h = ctx
    rc = lib.vsp_generator(h, None, 1, 8, C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), 1 << 20)
    assert rc == -2 and b"not finalised" in lib.vsp_last_error(h)
    assert lib.vsp_finalize_weights(h, None) == -2      # weights missing


def test_unsupported_configs_are_reported():
    lib = _lib.lib()
    d = ModelDims()
    d.n_heads = 3            # head dim 64 is covered; 192/3 = 64 -> ok; use 4 -> 48 unsupported
    d.n_heads = 4
    cfg = _lib.make_config(d)
    h = C.c_void_p()
    assert lib.vsp_create(C.byref(cfg), 0, C.byref(h)) == -7
    assert b"head dim" in lib.vsp_last_error(h)
    lib.vsp_destroy(h)
    d = ModelDims()
    d.n_speakers = 0
    cfg = _lib.make_config(d)
    assert lib.vsp_create(C.byref(cfg), 0, C.byref(h)) == -7
    lib.vsp_destroy(h)


def test_missing_extension_is_an_import_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.lib()


def test_model_refuses_cpu_device():
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    with pytest.raises(RuntimeError):
        SynthesizerTrn(*args, device="cpu", **kwargs)


def test_typed_weights_and_begin_weights_host_logic(ctx):
    """ABI 3 host logic (no GPU): vsp_set_weight_typed widens f16 / bf16 / f64 host data, rejects unknown dtypes;
    vsp_begin_weights forgets a load; a folded '<x>.weight' and a weight_g / weight_v pair replace each other."""
    import torch
    lib, h = ctx
    dims = ModelDims()
    schema = state_dict_schema(dims)
    n_used = sum(1 for k in schema if used_by_infer(k))
    key = "dec.conv_pre.weight"
    shape = schema[key]
    arr = (C.c_int64 * len(shape))(*shape)
    for name, dt in (("float16", torch.float16), ("bfloat16", torch.bfloat16), ("float64", torch.float64)):
        t = torch.randn(*shape).to(dt).contiguous()
        rc = lib.vsp_set_weight_typed(h, key.encode(), C.c_void_p(t.data_ptr()), arr, len(shape), _lib.DTYPES[name], 0)
        assert rc == 0, lib.vsp_last_error(h)
    assert lib.vsp_missing_weights(h) == n_used - 1
    t = torch.zeros(*shape)
    assert lib.vsp_set_weight_typed(h, key.encode(), C.c_void_p(t.data_ptr()), arr, len(shape), 9, 0) == -1
    # folded vs weight-norm forms of one layer: the later one wins, never both
    vk, gk = "dec.ups.0.weight_v", "dec.ups.0.weight_g"
    assert _set(lib, h, vk, np.zeros(schema[vk], np.float32)) == 0 and _set(lib, h, gk, np.zeros(schema[gk], np.float32)) == 0
    assert lib.vsp_missing_weights(h) == n_used - 3
    assert _set(lib, h, "dec.ups.0.weight", np.zeros(schema[vk], np.float32)) == 0      # folded form: replaces the pair
    assert lib.vsp_missing_weights(h) == n_used - 3                                      # (both still satisfied by it)
    assert _set(lib, h, vk, np.zeros(schema[vk], np.float32)) == 0                      # back: the folded tensor is dropped
    assert lib.vsp_missing_weights(h) == n_used - 2                                      # ... so weight_g is missing again
    assert lib.vsp_begin_weights(h) == 0
    assert lib.vsp_missing_weights(h) == n_used
    # an adopted arena cannot be committed without a device (and a context is not ready before)
    assert lib.vsp_commit_adopted_weights(h, None) == -2
    assert lib.vsp_generator_halo_frames(h) == 14
    # ABI 5: the sample-exact frame dependence the trimmed tails rest on, against an independent derivation: an impulse at
    # input position 0 reaches output samples [lo, hi]; conv (k, d): +- (k - 1) d / 2; transposed conv (k, s, p = (k - s) / 2):
    # [lo, hi] -> [lo s - p, hi s - p + k - 1] (reference models.py:255-257, 271-290; modules.py:187-223)
    back, fwd = C.c_int(-1), C.c_int(-1)
    assert lib.vsp_generator_frame_dependence(h, C.byref(back), C.byref(fwd)) == 0
    lo, hi, up = -3, 3, 1
    for k, s in zip((16, 16, 4, 4), (8, 8, 4, 2)):
        p = (k - s) // 2
        lo, hi, up = lo * s - p, hi * s - p + k - 1, up * s
        ext = max(sum((kk - 1) * d // 2 + (kk - 1) // 2 for d in (1, 3, 5)) for kk in (3, 7, 11))
        lo, hi = lo - ext, hi + ext
    lo, hi = lo - 3, hi + 3
    assert up == 512 and (back.value, fwd.value) == (hi // up, (up - 1 - lo) // up) == (13, 13)


def test_status_word_is_readable_without_a_device(ctx):
    """vsp_status (ABI 7) before anything has run: no flags, no error; null arguments are refused."""
    lib, h = ctx
    flags = C.c_uint(123)
    assert lib.vsp_status(h, C.byref(flags), 1) == 0 and flags.value == 0
    assert lib.vsp_status(h, None, 0) == -1 and lib.vsp_status(None, C.byref(flags), 0) == -1
