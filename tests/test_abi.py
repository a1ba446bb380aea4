"""The C-ABI library loads without a GPU, exports every symbol include/cid.h declares, and
rejects bad arguments with errno-style codes + a message (no compute is launched here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    src = (ROOT / "include" / "cid.h").read_text()
    return sorted(set(re.findall(r"\b(cid_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(lib):
    from consistentid_amd import _lib
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"libcid.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.cid_version() >= 100


def test_argument_validation_without_gpu(lib):
    from consistentid_amd._lib import GemmDesc
    d = GemmDesc()
    assert lib.cid_gemm_f16(C.byref(d), None) == -22
    assert b"null pointer" in lib.cid_last_error()
    assert lib.cid_layernorm_f16(None, None, None, None, 4, 320, 1e-5, None) == -22
    assert lib.cid_self_attn_f16(1, 1, 1, 1, 1, 100, 8, 40, 640, 640, 64, 320, None) == -22   # N % 64
    assert b"multiple of 64" in lib.cid_last_error()
    assert lib.cid_self_attn_f16(1, 1, 1, 1, 1, 128, 8, 48, 640, 640, 64, 320, None) == -22   # head dim
    assert lib.cid_id_xattn_f16(1, 1, None, None, None, 1e-5, 1, 1, None, 1, 1, 1, 2, 4096, 320, 7, 77, 4, 1.0, None) == -22
    assert lib.cid_conv_out_f16(1, 1, 1, 1, 1, 8, 8, 320, 5, None) == -22
    # mode 3 (query projection with the attention epilogue): refused without its operands / outside its geometry
    q = GemmDesc()
    q.x1, q.w, q.out = 64, 64, 64
    q.c1, q.ld1, q.ldo, q.M, q.N, q.taps, q.mode = 640, 640, 640, 1024, 640, 1, 3
    q.heads, q.dhead, q.ntok = 8, 80, 1024
    assert lib.cid_gemm_f16(C.byref(q), None) == -22 and b"att_kp" in lib.cid_last_error()
    q.att_kp, q.att_vp, q.att_kvrow, q.att_n_txt, q.att_n_ip = 64, 64, 64, 81, 0
    assert lib.cid_gemm_f16(C.byref(q), None) == -22 and b"77 + 4" in lib.cid_last_error()
    q.att_n_txt, q.att_n_ip, q.dhead, q.heads = 77, 4, 40, 16
    assert lib.cid_gemm_f16(C.byref(q), None) == -22 and b"dhead" in lib.cid_last_error()
    from consistentid_amd._lib import StepSeg
    seg = (StepSeg * 1)(StepSeg(16, 0, 8))
    assert lib.cid_step_select(None, 8, 1, 16, seg, 1, None) == -22            # null table
    assert lib.cid_step_select(16, 6, 1, 16, seg, 1, None) == -22              # row bytes % 4
    assert lib.cid_step_select(16, 8, 1, 16, seg, 9, None) == -22              # too many segments
    seg[0].nbytes = 12
    assert lib.cid_step_select(16, 8, 1, 16, seg, 1, None) == -22              # segment past the row
    assert lib.cid_kv_pack_elems(320, 8, 0) == 8 * 3 * 3 * 512
    assert lib.cid_kv_pack_elems(320, 8, 1) == 8 * 2 * 6 * 512
    assert lib.cid_groupnorm_ws_bytes(8, 2560) > 0


def test_python_front_end_refuses_cpu_tensors(lib):
    import torch
    from consistentid_amd import ops
    from consistentid_amd._lib import CidError
    x = torch.zeros(4, 320, dtype=torch.float16)
    with pytest.raises(CidError):
        ops.layernorm(x, x.clone(), x[0], x[0], M=4, C_=320)
