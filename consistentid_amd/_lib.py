"""ctypes binding of libcid.so (include/cid.h).  No fallback: if the HIP library is
missing or a symbol is absent this module raises -- the product path never runs on a
CPU/PyTorch substitute."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libcid.so"

c_half_p = C.c_void_p
c_stream = C.c_void_p


class GemmDesc(C.Structure):
    """struct cid_gemm_desc (include/cid.h)."""
    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p),
        ("c1", C.c_int32), ("c2", C.c_int32), ("ld1", C.c_int32), ("ld2", C.c_int32),
        ("w", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p), ("ld_rowbias", C.c_int32), ("rows_per_sample", C.c_int32),
        ("res", C.c_void_p), ("ldr", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32),
        ("taps", C.c_int32),
        ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("stride", C.c_int32), ("up", C.c_int32),
        ("mode", C.c_int32),
        ("vt", C.c_void_p), ("n_vt0", C.c_int32), ("heads", C.c_int32), ("dhead", C.c_int32),
        ("dvp", C.c_int32), ("ntok", C.c_int32),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("ln_s", C.c_void_p), ("ln_b", C.c_void_p), ("ln_eps", C.c_float),
        ("gn_stats", C.c_void_p),
        ("att_kp", C.c_void_p), ("att_vp", C.c_void_p), ("att_kvrow", C.c_void_p),
        ("att_n_txt", C.c_int32), ("att_n_ip", C.c_int32), ("att_ip_scale", C.c_float),
        ("out2", C.c_void_p),
    ]


class StepSeg(C.Structure):
    """struct cid_step_seg (include/cid.h)."""
    _fields_ = [("dst", C.c_void_p), ("offset", C.c_int64), ("nbytes", C.c_int64)]


# name -> (restype, argtypes); mirrors include/cid.h one to one
SIGNATURES = {
    "cid_version": (C.c_int, []),
    "cid_last_error": (C.c_char_p, []),
    "cid_gemm_f16": (C.c_int, [C.POINTER(GemmDesc), c_stream]),
    "cid_gemm_stats_rows": (C.c_int, [C.POINTER(GemmDesc)]),
    "cid_self_attn_f16": (C.c_int, [c_half_p] * 4 + [C.c_int32] * 8 + [c_stream]),
    "cid_self_attn_keys_f16": (C.c_int, [c_half_p] * 4 + [C.c_int32] * 9 + [c_stream]),
    "cid_id_xattn_f16": (C.c_int, [c_half_p] * 5 + [C.c_float] + [c_half_p] * 5 + [C.c_void_p]
                         + [C.c_int32] * 6 + [C.c_float, c_stream]),
    "cid_id_xattn_core_f16": (C.c_int, [c_half_p] * 4 + [C.c_void_p] + [C.c_int32] * 6 + [C.c_float, c_stream]),
    "cid_kv_pack_elems": (C.c_int64, [C.c_int32] * 3),
    "cid_kv_pack_f16": (C.c_int, [c_half_p] * 4 + [C.c_int32] * 5 + [c_stream]),
    "cid_pack_wfrag_f16": (C.c_int, [c_half_p] * 2 + [C.c_int32] * 2 + [c_stream]),
    "cid_kv_pack2_elems": (C.c_int64, [C.c_int32] * 3),
    "cid_id_xattn3_supported": (C.c_int, [C.c_int32] * 4),
    "cid_id_xattn3_f16": (C.c_int, [c_half_p] * 3 + [C.c_void_p] * 2 + [c_half_p] * 4 + [C.c_void_p]
                          + [C.c_int32] * 6 + [C.c_float, C.c_float, C.c_int32, c_stream]),
    "cid_gather_pack_f16": (C.c_int, [c_half_p, c_half_p, C.c_void_p, c_half_p, C.c_int32, C.c_int64, C.c_int64, c_stream]),
    "cid_layernorm_f16": (C.c_int, [c_half_p] * 4 + [C.c_int32] * 2 + [C.c_float, c_stream]),
    "cid_softmax_rows_f16": (C.c_int, [c_half_p, C.c_int32, C.c_int32, C.c_int64, c_stream]),
    "cid_groupnorm_ws_bytes": (C.c_int64, [C.c_int32] * 2),
    "cid_groupnorm_f16": (C.c_int, [c_half_p, c_half_p, C.c_int32, C.c_int32, c_half_p, c_half_p, c_half_p,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, c_stream]),
    "cid_groupnorm_stats_ok": (C.c_int, [C.c_int32] * 3),
    "cid_groupnorm_stats_f16": (C.c_int, [c_half_p, c_half_p, C.c_int32, C.c_int32, c_half_p, c_half_p, c_half_p,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_int32, c_stream]),
    "cid_gemm_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 12 + [c_stream]),
    "cid_groupnorm_f32_ws_bytes": (C.c_int64, [C.c_int32] * 3),
    "cid_groupnorm_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_float, C.c_int32, C.c_void_p, c_stream]),
    "cid_softmax_rows_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, c_stream]),
    "cid_conv_in_f16": (C.c_int, [c_half_p] * 4 + [C.c_int32] * 6 + [C.c_void_p, c_stream]),
    "cid_conv_in_cat_f16": (C.c_int, [c_half_p, C.c_int32, c_half_p, C.c_int32, c_half_p, c_half_p, c_half_p]
                            + [C.c_int32] * 5 + [C.c_void_p, c_stream]),
    "cid_conv3x3_small_f16": (C.c_int, [c_half_p] * 4 + [C.c_int32] * 7 + [c_stream]),
    "cid_gelu_f16": (C.c_int, [c_half_p, C.c_int64, c_stream]),
    "cid_small_attn_f16": (C.c_int, [c_half_p, C.c_int32, c_half_p, C.c_int32, c_half_p, C.c_int32, C.c_int32, c_half_p,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, c_stream]),
    "cid_conv_out_f16": (C.c_int, [c_half_p] * 4 + [C.c_int32] * 5 + [c_stream]),
    "cid_sincos_embed_f16": (C.c_int, [C.c_void_p, c_half_p, C.c_int32, C.c_int32, c_stream]),
    "cid_linear_small_f16": (C.c_int, [c_half_p, C.c_int32, c_half_p, c_half_p, c_half_p, C.c_int32, c_half_p,
                                       C.c_int32] + [C.c_int32] * 5 + [c_stream]),
    "cid_cfg_ddim_step_f16": (C.c_int, [c_half_p, c_half_p, C.c_void_p, C.c_float, c_half_p, c_half_p, c_half_p,
                                        C.c_int32, C.c_int32, c_stream]),
    "cid_add_inplace_f16": (C.c_int, [c_half_p, c_half_p, C.c_int64, C.c_int64, c_stream]),
    "cid_step_select": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(StepSeg), C.c_int32, c_stream]),
}

# entry points of experiment builds (build.py --variant ..., selected with CID_LIBRARY): bound when the library has them
OPTIONAL_SIGNATURES = {}

_lib = None


class CidError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libcid.so and bind every declared entry point (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("CID_LIBRARY") or LIB_PATH)   # CID_LIBRARY: an experiment build (build.py --variant)
    if not path.exists():
        raise CidError(f"{path} is missing: build it with `python -m consistentid_amd.build` "
                       f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: its wheel bundles the HIP runtime (libamdhip64) that owns the device context and the
    # streams we are handed; loading libcid.so before it would bind us to a second, uninitialised runtime
    import torch  # noqa: F401
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in OPTIONAL_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().cid_last_error().decode(errors="replace")
        raise CidError(f"{what or 'libcid'} failed (rc={rc}): {msg}")
