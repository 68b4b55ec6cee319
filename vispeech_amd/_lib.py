"""ctypes binding of libvispeech_hip.so (include/vispeech_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C vispeech_amd/csrc``.
There is NO fallback: if the shared object is missing or a symbol is absent, importing
callers get an ImportError -- the synthesis path never silently runs on anything else.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSP_LIB_PATH") or os.path.join(_HERE, "lib", "libvispeech_hip.so")

VSP_MAX_LIST = 8
ABI_VERSION = 7
DTYPES = {"float32": 0, "float16": 1, "bfloat16": 2, "float64": 3}   # VSP_DTYPE_*
PROF_GENERATOR, PROF_ATTENTION, PROF_FRAME = 0, 1, 2
FLAG_NONFINITE_LATENT, FLAG_NONFINITE_WAVE = 1, 2                    # VSP_FLAG_* (vsp_status)

ERRORS = {0: "VSP_OK", -1: "VSP_ERR_ARG", -2: "VSP_ERR_STATE", -3: "VSP_ERR_HIP", -4: "VSP_ERR_KEY",
          -5: "VSP_ERR_SHAPE", -6: "VSP_ERR_WORKSPACE", -7: "VSP_ERR_UNSUPPORTED"}


class VspConfig(C.Structure):
    """``vsp_config`` of include/vispeech_hip.h (field order is ABI)."""
    _fields_ = [
        ("n_vocab", C.c_int32), ("inter_channels", C.c_int32), ("hidden_channels", C.c_int32),
        ("filter_channels", C.c_int32), ("n_heads", C.c_int32), ("n_layers", C.c_int32),
        ("kernel_size", C.c_int32), ("n_resblock_kernels", C.c_int32),
        ("resblock_kernel_sizes", C.c_int32 * VSP_MAX_LIST), ("n_resblock_dilations", C.c_int32),
        ("resblock_dilation_sizes", (C.c_int32 * VSP_MAX_LIST) * VSP_MAX_LIST),
        ("n_upsamples", C.c_int32), ("upsample_rates", C.c_int32 * VSP_MAX_LIST),
        ("upsample_kernel_sizes", C.c_int32 * VSP_MAX_LIST), ("upsample_initial_channel", C.c_int32),
        ("n_speakers", C.c_int32), ("gin_channels", C.c_int32), ("window_size", C.c_int32),
        ("pitch_layers", C.c_int32), ("dur_filter", C.c_int32), ("energy_filter", C.c_int32),
        ("flow_kernel", C.c_int32), ("flow_layers", C.c_int32), ("n_flows", C.c_int32),
        ("spec_channels", C.c_int32), ("posterior_layers", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int
_I64 = C.c_int64
_F = C.c_float
_U64 = C.c_uint64

# name -> (restype, argtypes); every symbol include/vispeech_hip.h declares
SIGNATURES = {
    "vsp_abi_version": (_I, []),
    "vsp_create": (_I, [C.POINTER(VspConfig), _I, C.POINTER(_P)]),
    "vsp_destroy": (_I, [_P]),
    "vsp_last_error": (C.c_char_p, [_P]),
    "vsp_status": (_I, [_P, C.POINTER(C.c_uint), _I]),
    "vsp_set_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I]),
    "vsp_set_weight_typed": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I, _I, _I]),
    "vsp_begin_weights": (_I, [_P]),
    "vsp_missing_weights": (_I, [_P]),
    "vsp_weight_arena_bytes": (_I64, [_P]),
    "vsp_finalize_weights": (_I, [_P, _P]),
    "vsp_adopt_packed_weights": (_I, [_P, _P]),
    "vsp_commit_adopted_weights": (_I, [_P, _P]),
    "vsp_weight_arena": (_I, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "vsp_encode_workspace_bytes": (_I64, [_P, _I, _I]),
    "vsp_encode": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _I64]),
    "vsp_frame_lengths_host": (_I, [_P, _P, _I, _P, C.POINTER(_I64), C.POINTER(_I64)]),
    "vsp_decode_workspace_bytes": (_I64, [_P, _I, _I, _I]),
    "vsp_decode": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _U64, _F, _P, _P, _P, _P, _P, _P, _P, _I64]),
    "vsp_infer_workspace_bytes": (_I64, [_P, _I, _I, _I]),
    "vsp_infer": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P, _U64, _F,
                       _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64]),
    "vsp_attention_workspace_bytes": (_I64, [_P, _I, _I]),
    "vsp_attention": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I64]),
    "vsp_wn_layer_workspace_bytes": (_I64, [_P, _I, _I, _I]),
    "vsp_wn_layer": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I64]),
    "vsp_randn": (_I, [_P, _U64, _I64, _P]),
    "vsp_randn_at": (_I, [_P, _U64, _I64, _I64, _P]),
    "vsp_set_noise_offset": (_I, [_P, _I64]),
    "vsp_encoder_workspace_bytes": (_I64, [_P, _I, _I]),
    "vsp_encoder": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I64]),
    "vsp_length_regulate": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "vsp_flow_workspace_bytes": (_I64, [_P, _I, _I]),
    "vsp_flow_reverse": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _I64]),
    "vsp_generator_workspace_bytes": (_I64, [_P, _I, _I]),
    "vsp_generator": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _I64]),
    "vsp_generator_halo_frames": (_I, [_P]),
    "vsp_generator_frame_dependence": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "vsp_generator_stream_workspace_bytes": (_I64, [_P, _I, _I]),
    "vsp_generator_stream_chunk": (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _I64]),
    "vsp_flow_forward": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _I64]),
    "vsp_voice_conversion_workspace_bytes": (_I64, [_P, _I, _I]),
    "vsp_voice_conversion": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64]),
    "vsp_posterior_workspace_bytes": (_I64, [_P, _I, _I]),
    "vsp_posterior_encoder": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I64]),
    "vsp_has_voice_conversion": (_I, [_P]),
    "vsp_spectrogram_frames": (_I, [_P, _I, _I]),
    "vsp_spectrogram_workspace_bytes": (_I64, [_P, _I, _I, _I]),
    "vsp_spectrogram": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _I64]),
    "vsp_rq_spline": (_I, [_P, _I64, _I, _P, _P, _P, _P, _I, _F, _P, _P]),
    "vsp_mel_filterbank": (_I, [_I, _I, _I, _F, _F, _P]),
    "vsp_spec_to_mel": (_I, [_P, _I, _I, _I, _I, _I, _F, _F, _P, _P]),
    "vsp_cl_conv1d": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _F, _P, _I, _P]),
    "vsp_conv1d": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _F, _I, _P, _I, _I, _P]),
    "vsp_cl_resblock": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P]),
    "vsp_profile_enable": (_I, [_P, _I]),
    "vsp_profile_read": (_I, [_P, C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _I]),
    "vsp_profile_read_class": (_I, [_P, _I, C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _I]),
    "vsp_profile_read_families": (_I, [_P, _I, _I, C.POINTER(_I), C.POINTER(_I64), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and type the shared library.  Raises ImportError loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C vispeech_amd/csrc`). There is no CPU fallback for the synthesis path.")
    # One HIP runtime per process: torch ships its own libamdhip64; loaded AFTER a copy this library pulled in from
    # /opt/rocm, the two runtimes do not share a device context ("no ROCm-capable device" at the first hipMemcpy).
    import torch  # noqa: F401
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise ImportError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if l.vsp_abi_version() != ABI_VERSION:
        raise ImportError("libvispeech_hip ABI version mismatch")
    _lib = l
    return l


class VspError(RuntimeError):
    pass


def check(rc: int, ctx=None, what: str = "") -> None:
    if rc == 0:
        return
    msg = ""
    if ctx:
        m = lib().vsp_last_error(ctx)
        msg = m.decode("utf-8", "replace") if m else ""
    raise VspError(f"{what or 'libvispeech_hip'}: {ERRORS.get(rc, rc)}: {msg}")


def make_config(dims) -> VspConfig:
    """``vispeech_amd.schema.ModelDims`` -> ``vsp_config``."""
    c = VspConfig()
    c.n_vocab = dims.n_vocab
    c.inter_channels = dims.inter_channels
    c.hidden_channels = dims.hidden_channels
    c.filter_channels = dims.filter_channels
    c.n_heads = dims.n_heads
    c.n_layers = dims.n_layers
    c.kernel_size = dims.kernel_size
    ks = list(dims.resblock_kernel_sizes)
    ds = [list(x) for x in dims.resblock_dilation_sizes]
    if len(ks) > VSP_MAX_LIST or len(ks) != len(ds) or any(len(d) != len(ds[0]) or len(d) > VSP_MAX_LIST for d in ds):
        raise ValueError("resblock lists: at most 8 kernels, equal dilation counts")
    c.n_resblock_kernels = len(ks)
    c.n_resblock_dilations = len(ds[0])
    for i, k in enumerate(ks):
        c.resblock_kernel_sizes[i] = k
        for j, d in enumerate(ds[i]):
            c.resblock_dilation_sizes[i][j] = d
    ur, uk = list(dims.upsample_rates), list(dims.upsample_kernel_sizes)
    if len(ur) != len(uk) or len(ur) > VSP_MAX_LIST:
        raise ValueError("upsample lists: equal length, at most 8")
    c.n_upsamples = len(ur)
    for i, (u, k) in enumerate(zip(ur, uk)):
        c.upsample_rates[i] = u
        c.upsample_kernel_sizes[i] = k
    c.upsample_initial_channel = dims.upsample_initial_channel
    c.n_speakers = dims.n_speakers
    c.gin_channels = dims.gin_channels
    c.window_size = dims.window_size
    c.pitch_layers = dims.pitch_layers
    c.dur_filter = dims.dur_filter
    c.energy_filter = dims.energy_filter
    c.flow_kernel = dims.flow_kernel
    c.flow_layers = dims.flow_layers
    c.n_flows = dims.n_flows
    c.spec_channels = dims.spec_channels
    c.posterior_layers = dims.posterior_layers
    return c
