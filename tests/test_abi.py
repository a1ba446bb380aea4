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
    # plain launches: a misaligned residual pitch / a second destination outside mode 0 are refused before any launch
    r = GemmDesc()
    r.x1, r.w, r.out, r.res = 64, 64, 64, 64
    r.c1, r.ld1, r.ldo, r.ldr, r.M, r.N, r.taps = 320, 320, 320, 324, 256, 320, 1
    assert lib.cid_gemm_f16(C.byref(r), None) == -22 and b"ldr" in lib.cid_last_error()
    r.ldr, r.res = 320, 72
    assert lib.cid_gemm_f16(C.byref(r), None) == -22 and b"16-byte aligned" in lib.cid_last_error()
    r.res, r.out2, r.mode, r.N, r.ldo = 64, 64, 1, 640, 320
    assert lib.cid_gemm_f16(C.byref(r), None) == -22 and b"out2" in lib.cid_last_error()
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


def test_conv_tile_plan_without_gpu(lib):
    """cid_gemm_stats_rows is host code: it runs plan_gemm and reports the tile height a launch would use for GroupNorm
    statistics -- which doubles as a probe of the convolution tile rules (csrc/gemm.hip plan_gemm, csrc/conv3x3.hip):
    256-token tiles where they fill the chip, 128-token tiles for short K, no statistics from split-K launches."""
    from consistentid_amd._lib import GemmDesc

    def rows(B, side, cin, cout, up=0, stride=1, taps=9, ws=True):
        d = GemmDesc()
        d.x1, d.w, d.out = 64, 64, 64
        so = side << up if stride == 1 else side // 2
        d.c1, d.ld1, d.ldo, d.N, d.taps = cin, cin, cout, cout, taps
        d.M = B * so * so
        d.Hi, d.Wi, d.Ho, d.Wo, d.stride, d.up = side, side, so, so, stride, up
        if ws:
            d.ws, d.ws_bytes = 64, 64 << 20
        return lib.cid_gemm_stats_rows(C.byref(d))

    assert rows(8, 64, 320, 320) == 256          # level 0: 256 tiles of 256 tokens (conv3x3.hip)
    assert rows(8, 64, 960, 320) == 256
    assert rows(4, 64, 320, 320) == 128          # CFG-deduplicated level 0: 128-token tiles, five channel slabs
    assert rows(8, 32, 640, 640) == 128          # 32 x 32 level, ten channel slabs
    assert rows(8, 32, 1280, 640) == 0           # ... twenty: halo kernel + split-K, no statistics
    assert rows(8, 32, 1280, 1280) == 256        # (1280 output channels: 256 tiles again)
    assert rows(8, 16, 1280, 1280) == 0          # 16 x 16 level: split-K
    assert rows(8, 32, 640, 640, up=1) == 256    # Upsample2D conv 32 -> 64: 512 tiles of 256 output tokens
    assert rows(8, 16, 1280, 1280, up=1) == 256
    assert rows(8, 8, 1280, 1280, up=1) == 0     # 8 -> 16: 64 tiles, split-K
    assert rows(4, 128, 320, 320) == 128         # SDXL 128 x 128 level: a 256-token halo would be 520 rows (> 400)
    assert rows(8, 64, 320, 320, stride=2) in (0, 256, 128)     # (strided convolutions stay on the gather path; any tile)
