"""Thin torch-tensor front end of the C ABI (include/cid.h).  PyTorch is used for
device memory and the current HIP stream only; every op below is one (or a fixed few)
hand-written HIP kernel launch in libcid.so.  Nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import GemmDesc, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def _req(t: torch.Tensor, name: str, dtype=torch.float16):
    if not t.is_cuda:
        raise _lib.CidError(f"{name}: tensor must live on the GPU (got {t.device}); libcid has no CPU path")
    if t.dtype != dtype:
        raise _lib.CidError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.data_ptr() % 16:
        raise _lib.CidError(f"{name}: base pointer must be 16-byte aligned")


def dvp_of(d: int) -> int:
    """rows reserved per head in the transposed-V buffer (head dim rounded up to 32)."""
    return (d + 31) // 32 * 32


# --------------------------------------------------------------------------- GEMM / conv
def gemm(x1: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, M: int, N: int, c1: int, ld1: Optional[int] = None,
         x2: Optional[torch.Tensor] = None, c2: int = 0, ld2: Optional[int] = None, ldo: Optional[int] = None,
         bias: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None, ld_rowbias: int = 0,
         rows_per_sample: int = 1, res: Optional[torch.Tensor] = None, ldr: Optional[int] = None,
         taps: int = 1, Hi: int = 0, Wi: int = 0, Ho: int = 0, Wo: int = 0, stride: int = 1, up: int = 0,
         mode: int = 0, vt: Optional[torch.Tensor] = None, n_vt0: int = 0, heads: int = 0, dhead: int = 0,
         ntok: int = 0, ws: Optional[torch.Tensor] = None, ln=None, gn_hw: int = 0, att=None,
         out2: Optional[torch.Tensor] = None):
    """``att`` = (kp, vp, kvrow, n_txt, n_ip, ip_scale) with ``mode=3``: the query projection of the identity cross-attention
    with the two-stream attention as its epilogue (``heads``, ``dhead``, ``ntok`` describe the heads and the tokens per sample).
    ``ln`` = (s, b, eps): LayerNorm folded into the projection -- ``x1`` is the raw residual stream, ``w`` carries gamma,
    s / b are the fp32 [N] fold vectors (weights.fold_ln); no ``bias`` then (it is inside b).
    ``gn_hw`` > 0: the consumer of ``out`` is a GroupNorm over samples of ``gn_hw`` tokens -- if this launch can emit the
    statistics from its epilogue they are attached as ``out._gn_stats = (fp32 [M / rows, 32, 2], rows)`` for
    ``groupnorm`` to pick up (saves its statistics pass).
    ``out2``: a second destination for the same rows (same pitch as ``out``; mode 0): the CFG duplication of a tensor both
    halves of the batch share, written by the producer."""
    lib = _lib.load()
    for name, t in (("x1", x1), ("w", w), ("out", out)):
        _req(t, f"gemm.{name}")
    for name, t in (("x2", x2), ("bias", bias), ("rowbias", rowbias), ("res", res), ("vt", vt), ("out2", out2)):
        if t is not None:
            _req(t, f"gemm.{name}")
    d = GemmDesc()
    d.out2 = _p(out2)
    d.x1, d.x2 = _p(x1), _p(x2)
    d.c1, d.c2 = c1, c2
    d.ld1 = ld1 if ld1 is not None else c1
    d.ld2 = ld2 if ld2 is not None else c2
    d.w = _p(w)
    d.out = _p(out)
    n_out = N // 2 if mode == 1 else (n_vt0 if mode == 2 else N)
    d.ldo = ldo if ldo is not None else n_out
    d.bias = _p(bias)
    d.rowbias, d.ld_rowbias, d.rows_per_sample = _p(rowbias), ld_rowbias, rows_per_sample
    d.res = _p(res)
    d.ldr = ldr if ldr is not None else N
    d.M, d.N, d.taps = M, N, taps
    d.Hi, d.Wi, d.Ho, d.Wo, d.stride, d.up = Hi, Wi, Ho, Wo, stride, up
    d.mode = mode
    d.vt, d.n_vt0, d.heads, d.dhead, d.dvp, d.ntok = _p(vt), n_vt0, heads, dhead, dvp_of(dhead) if dhead else 0, ntok
    if ws is not None:
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    if att is not None:
        kp, vp, kvrow, n_txt, n_ip, ip_scale = att
        _req(kp, "gemm.att_kp")
        _req(vp, "gemm.att_vp")
        _req(kvrow, "gemm.att_kvrow", torch.int32)
        d.att_kp, d.att_vp, d.att_kvrow = kp.data_ptr(), vp.data_ptr(), kvrow.data_ptr()
        d.att_n_txt, d.att_n_ip, d.att_ip_scale = int(n_txt), int(n_ip), float(ip_scale)
    if ln is not None:
        ls, lb, eps = ln
        _req(ls, "gemm.ln_s", torch.float32)
        _req(lb, "gemm.ln_b", torch.float32)
        d.ln_s, d.ln_b, d.ln_eps = ls.data_ptr(), lb.data_ptr(), float(eps)
    stats = None
    if gn_hw > GN_SMALL_MAX_HW and GN_EPILOGUE_STATS:      # (smaller samples: cid_groupnorm_f16 is one launch anyway)
        rows = int(lib.cid_gemm_stats_rows(C.byref(d)))
        if rows > 0 and gn_hw % rows == 0:
            stats = torch.empty(M // rows, 32, 2, dtype=torch.float32, device=out.device)
            d.gn_stats = stats.data_ptr()
    check(lib.cid_gemm_f16(C.byref(d), _stream()), "cid_gemm_f16")
    if stats is not None:
        out._gn_stats = (stats, rows)
    elif hasattr(out, "_gn_stats"):
        del out._gn_stats            # a reused output tensor must not carry the statistics of what it held before
    return out


LN_EPS = 1e-5        # diffusers BasicTransformerBlock LayerNorms: shared by layernorm(), the folded projections and the fused kernels

def qattn_supported(C_: int, heads: int, N: int, n_txt: int, n_ip: int) -> bool:
    """can ``gemm(mode=3)`` (query projection + attention epilogue) serve this cross-attention level?"""
    if heads <= 0 or C_ % heads:
        return False
    d = C_ // heads
    return (d in (64, 80, 160) and n_txt == 77 and n_ip == 4 and N % 64 == 0 and (d != 64 or (C_ % 128 == 0 and N % 128 == 0))
            and C_ % (128 if d == 64 else 160) == 0)


# A/B switches (environment): the GEMM epilogues emit GroupNorm statistics / LayerNorm is folded into the projections
GN_EPILOGUE_STATS = os.environ.get("CID_GN_EPILOGUE_STATS", "1") != "0"
# LayerNorm fold: "auto" (default) folds where it measured faster than layernorm + GEMM -- up to 2048 tokens per launch (the
# in-loop row statistics cost the GEMMs of the larger levels more than the LayerNorm kernel they replace:
# profiles/r03_kbench.txt; SDXL folded everywhere 1.52 images/s, with this rule 1.59 -- profiles/r03_bench_sdxl_*.json);
# "1" always, "0" never
_LN_FOLD_MODE = os.environ.get("CID_LN_FOLD", "auto")


def ln_fold(M: int) -> bool:
    return _LN_FOLD_MODE == "1" or (_LN_FOLD_MODE == "auto" and M <= 2048)


_GEGLU_H32 = os.environ.get("CID_GEGLU_H32", "1") != "0"
_GEGLU_FOLD_MAX = int(os.environ.get("CID_GEGLU_FOLD_MAX", "8192"))      # A/B switch (2048 = the rule of rounds 3-5)


def ln_fold_geglu(M: int, C_: int) -> bool:
    """norm3 folded into the GEGLU projection?  Like :func:`ln_fold`, except where the launch runs on csrc/linear_h32.hip
    (deep K, >= 256 tiles of 256 x 160: plan_gemm's rule), which takes a plain LayerNorm-ed input: layernorm + that kernel
    measured 8 + 64 us against 84 us for the folded 16 x 16 x 32 form at SD1.5's 16 x 16 level (profiles/r06_kbench.txt)"""
    h32 = _GEGLU_H32 and C_ >= 1024 and M % 256 == 0 and (M // 256) * (8 * C_ // 160) >= 256 and (8 * C_) % 160 == 0
    if _LN_FOLD_MODE != "auto":
        return ln_fold(M)
    # (the GEGLU launch takes its row statistics during the first n-tile only: folded it stays ahead of layernorm + GEMM up to
    #  8192 tokens -- 69.7 vs 8.0 + 65.0 us at SD1.5's 32 x 32 level, profiles/r06_kbench.txt; 86.0 vs 10.2 + 73.9 at 64 x 64)
    return M <= _GEGLU_FOLD_MAX and not h32


# the query projection of the cross-attention with the attention epilogue (gemm mode 3): folded up to 8192 tokens per launch
# (SD1.5's 32 x 32 level at CFG batch 8: 43.1 -> 39.9 us with norm2 folded; SDXL's 4096-token level 58.4 vs 57.9; A/B switch)
_QATTN_FOLD_MAX = int(os.environ.get("CID_QATTN_LNFOLD_MAX", "8192"))


def ln_fold_q(M: int) -> bool:
    return _LN_FOLD_MODE == "1" or (_LN_FOLD_MODE == "auto" and M <= _QATTN_FOLD_MAX)


# --------------------------------------------------------------------------- attention
def self_attn(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, *, B: int, N: int,
              heads: int, d: int, ldq: int, ldk: int, ldo: int, n_keys: Optional[int] = None):
    """``n_keys``: only the first n_keys tokens of every sample are real keys (padding beyond)"""
    lib = _lib.load()
    for name, t in (("q", q), ("k", k), ("vt", vt), ("out", out)):
        _req(t, f"self_attn.{name}")
    if n_keys is not None and n_keys != N:
        check(lib.cid_self_attn_keys_f16(_p(q), _p(k), _p(vt), _p(out), B, N, heads, d, ldq, ldk, dvp_of(d), ldo, n_keys,
                                         _stream()), "cid_self_attn_keys_f16")
        return out
    check(lib.cid_self_attn_f16(_p(q), _p(k), _p(vt), _p(out), B, N, heads, d, ldq, ldk, dvp_of(d), ldo, _stream()),
          "cid_self_attn_f16")
    return out


def kv_pack_elems(C_: int, heads: int):
    lib = _lib.load()
    return int(lib.cid_kv_pack_elems(C_, heads, 0)), int(lib.cid_kv_pack_elems(C_, heads, 1))


def kv_pack(kv_txt: torch.Tensor, kv_ip: torch.Tensor, kp: torch.Tensor, vp: torch.Tensor, *, R: int, C_: int,
            heads: int, n_txt: int, n_ip: int):
    lib = _lib.load()
    for name, t in (("kv_txt", kv_txt), ("kv_ip", kv_ip), ("kp", kp), ("vp", vp)):
        _req(t, f"kv_pack.{name}")
    check(lib.cid_kv_pack_f16(_p(kv_txt), _p(kv_ip), _p(kp), _p(vp), R, C_, heads, n_txt, n_ip, _stream()),
          "cid_kv_pack_f16")


def pack_wfrag(w: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _req(w, "pack_wfrag.w")
    rows, K = w.shape
    out = torch.empty_like(w)
    check(lib.cid_pack_wfrag_f16(_p(w.contiguous()), _p(out), rows, K, _stream()), "cid_pack_wfrag_f16")
    return out


def id_xattn(x: torch.Tensor, out: torch.Tensor, *, wq: torch.Tensor, wo: torch.Tensor, bo: Optional[torch.Tensor],
             kp: torch.Tensor, vp: torch.Tensor, kvrow: torch.Tensor, B: int, N: int, C_: int, heads: int,
             n_txt: int, n_ip: int, ip_scale: float, residual: Optional[torch.Tensor] = None,
             ln_gamma: Optional[torch.Tensor] = None, ln_beta: Optional[torch.Tensor] = None, ln_eps: float = 1e-5):
    lib = _lib.load()
    for name, t in (("x", x), ("out", out), ("wq", wq), ("wo", wo), ("kp", kp), ("vp", vp)):
        _req(t, f"id_xattn.{name}")
    for name, t in (("bo", bo), ("residual", residual), ("ln_gamma", ln_gamma), ("ln_beta", ln_beta)):
        if t is not None:
            _req(t, f"id_xattn.{name}")
    _req(kvrow, "id_xattn.kvrow", torch.int32)
    check(lib.cid_id_xattn_f16(_p(x), _p(out), _p(residual), _p(ln_gamma), _p(ln_beta), ln_eps, _p(wq), _p(wo),
                               _p(bo), _p(kp), _p(vp), _p(kvrow), B, N, C_, heads, n_txt, n_ip, float(ip_scale),
                               _stream()), "cid_id_xattn_f16")
    return out


def kv_pack2_elems(C_: int, heads: int):
    lib = _lib.load()
    return int(lib.cid_kv_pack2_elems(C_, heads, 0)), int(lib.cid_kv_pack2_elems(C_, heads, 1))


_KV_IDX_CACHE = {}


def kv_pack2(kv_txt: torch.Tensor, kv_ip: torch.Tensor, kp: torch.Tensor, vp: torch.Tensor, *, R: int, L: int, C_: int,
             heads: int, n_txt: int, n_ip: int, order: str = "slot"):
    """projected [K | V] rows of R context rows ([R, L, 2C], text and ID projections) -> the fragment-ordered
    operands of cid_id_xattn3_f16 (order "reg": register-major keys, xattn_pack.slot_key; order "slot" is the
    key order of the retired second generation, kept for the layout tests); index tables from xattn_pack, cached on the device"""
    from . import xattn_pack
    lib = _lib.load()
    for name, t in (("kv_txt", kv_txt), ("kv_ip", kv_ip), ("kp", kp), ("vp", vp)):
        _req(t, f"kv_pack2.{name}")
    key = (C_, heads, n_txt, n_ip, order, kv_txt.device)
    if key not in _KV_IDX_CACHE:
        ki, vi = xattn_pack.kv_index_tables(C_, heads, n_txt, n_ip, order)
        _KV_IDX_CACHE[key] = (torch.from_numpy(ki).to(kv_txt.device), torch.from_numpy(vi).to(kv_txt.device))
    ki, vi = _KV_IDX_CACHE[key]
    assert L == n_txt + n_ip
    for idx, dst in ((ki, kp), (vi, vp)):
        check(lib.cid_gather_pack_f16(_p(kv_txt), _p(kv_ip), idx.data_ptr(), _p(dst), R, L * 2 * C_, idx.numel(), _stream()),
              "cid_gather_pack_f16")


XATTN_GEN_DEFAULT = 3


def xattn_generation() -> int:
    """which fused cross-attention kernel serves the SD1.5 level-0 geometry: 3 = csrc/xattn3.hip,
    1 = the first-generation csrc/xattn.hip (A/B switch: CID_XATTN_GEN; CID_XATTN_V2=0 is the older spelling of 1)"""
    import os
    if os.environ.get("CID_XATTN_V2", "1") == "0":
        return 1
    return int(os.environ.get("CID_XATTN_GEN", str(XATTN_GEN_DEFAULT)))


def id_xattn3_supported(C_: int, heads: int, n_txt: int, n_ip: int) -> bool:
    return bool(_lib.load().cid_id_xattn3_supported(C_, heads, n_txt, n_ip))


def id_xattn3(x: torch.Tensor, out: torch.Tensor, *, wq_p: torch.Tensor, q_rowsum: torch.Tensor, q_bias: torch.Tensor,
              wo_p: torch.Tensor, bo: Optional[torch.Tensor], kp: torch.Tensor, vp: torch.Tensor, kvrow: torch.Tensor,
              B: int, N: int, C_: int, heads: int, n_txt: int, n_ip: int, ip_scale: float, has_ln: bool,
              add_residual: bool, ln_eps: float = 1e-5):
    """third-generation fused identity cross-attention: ``wq_p`` / ``wo_p`` = xattn_pack.pack_w3 of the (LayerNorm-folded)
    query and the output projection; K / V operands from kv_pack2(order="reg")"""
    lib = _lib.load()
    for name, t in (("x", x), ("out", out), ("wq_p", wq_p), ("wo_p", wo_p), ("kp", kp), ("vp", vp)):
        _req(t, f"id_xattn3.{name}")
    if bo is not None:
        _req(bo, "id_xattn3.bo")
    _req(q_rowsum, "id_xattn3.q_rowsum", torch.float32)
    _req(q_bias, "id_xattn3.q_bias", torch.float32)
    _req(kvrow, "id_xattn3.kvrow", torch.int32)
    check(lib.cid_id_xattn3_f16(_p(x), _p(out), _p(wq_p), _p(q_rowsum), _p(q_bias), _p(wo_p), _p(bo), _p(kp), _p(vp),
                                _p(kvrow), B, N, C_, heads, n_txt, n_ip, float(ip_scale), float(ln_eps),
                                (1 if has_ln else 0) | (2 if add_residual else 0), _stream()), "cid_id_xattn3_f16")
    return out


def id_xattn_core(q: torch.Tensor, out: torch.Tensor, *, kp: torch.Tensor, vp: torch.Tensor, kvrow: torch.Tensor,
                  B: int, N: int, C_: int, heads: int, n_txt: int, n_ip: int, ip_scale: float):
    lib = _lib.load()
    for name, t in (("q", q), ("out", out), ("kp", kp), ("vp", vp)):
        _req(t, f"id_xattn_core.{name}")
    _req(kvrow, "id_xattn_core.kvrow", torch.int32)
    check(lib.cid_id_xattn_core_f16(_p(q), _p(out), _p(kp), _p(vp), _p(kvrow), B, N, C_, heads, n_txt, n_ip,
                                    float(ip_scale), _stream()), "cid_id_xattn_core_f16")
    return out


# --------------------------------------------------------------------------- norms
def layernorm(x: torch.Tensor, out: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, M: int, C_: int,
              eps: float = 1e-5):
    lib = _lib.load()
    for name, t in (("x", x), ("out", out), ("gamma", gamma), ("beta", beta)):
        _req(t, f"layernorm.{name}")
    check(lib.cid_layernorm_f16(_p(x), _p(out), _p(gamma), _p(beta), M, C_, eps, _stream()), "cid_layernorm_f16")
    return out


GN_SMALL_MAX_HW = 256     # up to 16 x 16 pixels cid_groupnorm_f16 is ONE launch anyway (gn_small_kernel, GNS_MAXHW)


def groupnorm_ws_bytes(B: int, C_: int) -> int:
    return int(_lib.load().cid_groupnorm_ws_bytes(B, C_))


def groupnorm(x1: torch.Tensor, out: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, ws: torch.Tensor, *,
              B: int, HW: int, c1: int, x2: Optional[torch.Tensor] = None, c2: int = 0, groups: int = 32,
              eps: float = 1e-5, silu: bool = True):
    lib = _lib.load()
    for name, t in (("x1", x1), ("out", out), ("gamma", gamma), ("beta", beta)):
        _req(t, f"groupnorm.{name}")
    if x2 is not None:
        _req(x2, "groupnorm.x2")
    st1, st2 = getattr(x1, "_gn_stats", None), (getattr(x2, "_gn_stats", None) if x2 is not None else None)
    if st1 is not None and (x2 is None or st2 is not None) and HW > GN_SMALL_MAX_HW and \
            lib.cid_groupnorm_stats_ok(c1, c2, groups) and st1[0].shape[0] * st1[1] == B * HW and \
            (st2 is None or st2[0].shape[0] * st2[1] == B * HW):
        # statistics already emitted by the GEMM epilogue(s) that wrote x1 / x2: one launch, one read of x
        check(lib.cid_groupnorm_stats_f16(_p(x1), _p(x2), c1, c2, _p(out), _p(gamma), _p(beta), B, HW, groups, eps,
                                          1 if silu else 0, st1[0].data_ptr(), st1[1],
                                          st2[0].data_ptr() if st2 is not None else None, st2[1] if st2 is not None else 0,
                                          _stream()), "cid_groupnorm_stats_f16")
        return out
    if ws.numel() * ws.element_size() < groupnorm_ws_bytes(B, c1 + c2):
        raise _lib.CidError("groupnorm: workspace too small")
    check(lib.cid_groupnorm_f16(_p(x1), _p(x2), c1, c2, _p(out), _p(gamma), _p(beta), B, HW, groups, eps,
                                1 if silu else 0, ws.data_ptr(), _stream()), "cid_groupnorm_f16")
    return out


# --------------------------------------------------------------------------- UNet ends, time path, loop glue
def conv_in(sample: torch.Tensor, out: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, *, B: int, Bin: int,
            cin: int, H: int, W: int, cout: int, in_scale: Optional[torch.Tensor] = None,
            extra: Optional[torch.Tensor] = None):
    """``in_scale``: fp32 device scalar multiplying the sample (scheduler.scale_model_input); ``extra``: NCHW
    [Bin, cin - sample channels, H, W] read as the trailing input channels, unscaled (9-channel inpainting UNets:
    cat([mask, masked_image_latents]), inpaint ref :320-321)"""
    lib = _lib.load()
    for name, t in (("sample", sample), ("out", out), ("w", w), ("bias", bias)):
        _req(t, f"conv_in.{name}")
    if in_scale is not None:
        _req(in_scale, "conv_in.in_scale", torch.float32)
    if extra is not None:
        _req(extra, "conv_in.extra")
        cin1, cin2 = sample.shape[1], extra.shape[1]
        if cin1 + cin2 != cin or extra.shape[0] != Bin or tuple(extra.shape[2:]) != (H, W):
            raise _lib.CidError(f"conv_in: sample {tuple(sample.shape)} + extra {tuple(extra.shape)} do not make {cin} channels")
        check(lib.cid_conv_in_cat_f16(_p(sample), cin1, _p(extra), cin2, _p(out), _p(w), _p(bias), B, Bin, H, W, cout,
                                      _p(in_scale), _stream()), "cid_conv_in_cat_f16")
        return out
    check(lib.cid_conv_in_f16(_p(sample), _p(out), _p(w), _p(bias), B, Bin, cin, H, W, cout, _p(in_scale), _stream()),
          "cid_conv_in_f16")
    return out


def gelu_(x: torch.Tensor):
    """in-place exact-erf GELU"""
    lib = _lib.load()
    _req(x, "gelu.x")
    check(lib.cid_gelu_f16(_p(x), x.numel(), _stream()), "cid_gelu_f16")
    return x


def small_attn(q: torch.Tensor, kv1: torch.Tensor, kv2: Optional[torch.Tensor], out: torch.Tensor, *, B: int, Lq: int,
               n1: int, n2: int, heads: int, dim_head: int = 64):
    """PerceiverAttention core: q [B*Lq, heads*64], kv* [B*n*, 2*heads*64] ([K | V] rows), out [B*Lq, heads*64]"""
    lib = _lib.load()
    for name, t in (("q", q), ("kv1", kv1), ("out", out)) + ((("kv2", kv2),) if kv2 is not None else ()):
        _req(t, f"small_attn.{name}")
    inner = heads * dim_head
    check(lib.cid_small_attn_f16(_p(q), inner, _p(kv1), n1, _p(kv2), n2, 2 * inner, _p(out), inner, B, Lq, heads,
                                 dim_head, dim_head ** -0.5, _stream()), "cid_small_attn_f16")
    return out


def softmax_rows(x: torch.Tensor, *, rows: int, cols: int, ld: int):
    """in-place base-2 softmax over fp16 rows (VAE mid-block attention scores)"""
    lib = _lib.load()
    _req(x, "softmax_rows.x")
    check(lib.cid_softmax_rows_f16(_p(x), rows, cols, ld, _stream()), "cid_softmax_rows_f16")
    return x


def conv3x3_small(x: torch.Tensor, out: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, *, B: int, Hi: int, Wi: int,
                  cin: int, cout: int, stride: int = 1, silu: bool = False):
    """token-major [B, Hi*Wi, cin] -> [B, Ho*Wo, cout]; small-channel direct convolution (ControlNet condition embedding)."""
    lib = _lib.load()
    for name, t in (("x", x), ("out", out), ("w", w), ("bias", bias)):
        _req(t, f"conv3x3_small.{name}")
    check(lib.cid_conv3x3_small_f16(_p(x), _p(out), _p(w), _p(bias), B, Hi, Wi, cin, cout, stride, int(silu), _stream()),
          "cid_conv3x3_small_f16")
    return out


def conv_out(x: torch.Tensor, out: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, *, B: int, H: int, W: int,
             cin: int, cout: int):
    lib = _lib.load()
    for name, t in (("x", x), ("out", out), ("w", w), ("bias", bias)):
        _req(t, f"conv_out.{name}")
    check(lib.cid_conv_out_f16(_p(x), _p(out), _p(w), _p(bias), B, H, W, cin, cout, _stream()), "cid_conv_out_f16")
    return out


def sincos_embed(v: torch.Tensor, out: torch.Tensor, *, rows: int, dim: int):
    lib = _lib.load()
    _req(v, "sincos_embed.v", torch.float32)
    _req(out, "sincos_embed.out")
    check(lib.cid_sincos_embed_f16(_p(v), _p(out), rows, dim, _stream()), "cid_sincos_embed_f16")
    return out


def linear_small(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], out: torch.Tensor, *, M: int, N: int,
                 K: int, ldx: Optional[int] = None, ldo: Optional[int] = None, add: Optional[torch.Tensor] = None,
                 ldadd: int = 0, act_in: int = 0, act_out: int = 0):
    lib = _lib.load()
    for name, t in (("x", x), ("w", w), ("out", out)):
        _req(t, f"linear_small.{name}")
    check(lib.cid_linear_small_f16(_p(x), ldx if ldx is not None else K, _p(w), _p(b), _p(add), ldadd, _p(out),
                                   ldo if ldo is not None else N, M, N, K, act_in, act_out, _stream()),
          "cid_linear_small_f16")
    return out


def cfg_ddim_step(eps: torch.Tensor, latents: torch.Tensor, coef: torch.Tensor, guidance: float, *, B: int,
                  per_sample: int, mask: Optional[torch.Tensor] = None, init: Optional[torch.Tensor] = None,
                  noise: Optional[torch.Tensor] = None):
    lib = _lib.load()
    _req(eps, "cfg_ddim_step.eps")
    _req(latents, "cfg_ddim_step.latents")
    _req(coef, "cfg_ddim_step.coef", torch.float32)
    check(lib.cid_cfg_ddim_step_f16(_p(eps), _p(latents), _p(coef), float(guidance), _p(mask), _p(init), _p(noise),
                                    B, per_sample, _stream()), "cid_cfg_ddim_step_f16")
    return latents


def add_inplace(y: torch.Tensor, a: torch.Tensor):
    lib = _lib.load()
    _req(y, "add_inplace.y")
    _req(a, "add_inplace.a")
    if hasattr(y, "_gn_stats"):
        del y._gn_stats              # the epilogue statistics describe the tensor BEFORE this in-place update
    check(lib.cid_add_inplace_f16(_p(y), _p(a), y.numel(), a.numel(), _stream()), "cid_add_inplace_f16")
    return y


class StepTable:
    """Per-step values of one generation as ONE device table + the buffers the captured step reads (cid_step_select):
    ``columns`` = [(destination tensor, per-step values [S, ...] of the same dtype / trailing shape)].  ``select()`` is
    the first launch of a step: row ``counter`` -> destinations, ``counter`` += 1.  The destinations keep their addresses
    (the captured graph stays valid); the table is rebuilt per generation."""

    def __init__(self, columns, device, alloc=None):
        """``alloc(name, tensor, dtype)`` -> a persistent device tensor holding ``tensor`` (the denoise engine's static-buffer
        pool: stable addresses across generations, so a captured step stays valid); default: fresh tensors."""
        # (exceptions, not asserts: the columns are built from run-time tensors, and a mismatch would let cid_step_select copy
        #  the wrong byte ranges into the buffers the captured graph reads -- python -O must not remove the check)
        if not 0 < len(columns) <= 8:
            raise ValueError(f"StepTable takes 1..8 columns (got {len(columns)})")
        S = columns[0][1].shape[0]
        rows, segs, off = [], [], 0
        for j, (dst, vals) in enumerate(columns):
            if vals.dtype != dst.dtype:
                raise TypeError(f"StepTable column {j}: values are {vals.dtype}, the destination is {dst.dtype}")
            if vals.shape[0] != S or vals[0].numel() != dst.numel():
                raise ValueError(f"StepTable column {j}: values {tuple(vals.shape)} do not match {S} rows of the destination "
                                 f"{tuple(dst.shape)}")
            _req(dst, "StepTable.dst", dst.dtype)
            b = vals.to(device).contiguous().view(S, -1).view(torch.uint8)
            if b.shape[1] % 4 != 0:
                raise ValueError(f"StepTable column {j}: {b.shape[1]} bytes per row (columns are multiples of 4 bytes)")
            pad = -b.shape[1] % 16          # columns start on 16-byte boundaries of a 16-byte-multiple row: cid_step_select
            rows.append(b)                  # moves 16 bytes per lane where source and destination allow it
            segs.append((dst, off, b.shape[1]))
            off += b.shape[1] + pad
            if pad:
                rows.append(torch.zeros(S, pad, dtype=torch.uint8, device=b.device))
        table = torch.cat(rows, dim=1).contiguous()
        counter = torch.zeros(1, dtype=torch.int32, device=device)
        self.table = alloc("step_table", table, torch.uint8) if alloc else table
        self.counter = alloc("step_counter", counter, torch.int32) if alloc else counter
        self.n_rows, self.row_bytes = S, off
        self._segs = (_lib.StepSeg * len(segs))(*[_lib.StepSeg(d.data_ptr(), o, n) for d, o, n in segs])
        self._keep = [d for d, _, _ in segs]

    def reset(self, step: int = 0):
        self.counter.fill_(int(step))

    def select(self):
        lib = _lib.load()
        check(lib.cid_step_select(self.table.data_ptr(), self.row_bytes, self.n_rows, self.counter.data_ptr(), self._segs,
                                  len(self._keep), _stream()), "cid_step_select")


# --------------------------------------------------------------------------- fp32 (SDXL VAE decode)
def gemm_f32(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, M: int, N: int, c: int, bias: Optional[torch.Tensor] = None,
             res: Optional[torch.Tensor] = None, taps: int = 1, Hi: int = 0, Wi: int = 0, up: int = 0, ldx: Optional[int] = None,
             ldo: Optional[int] = None, ldr: Optional[int] = None):
    lib = _lib.load()
    for name, t in (("x", x), ("w", w), ("out", out)) + ((("bias", bias),) if bias is not None else ()) \
            + ((("res", res),) if res is not None else ()):
        _req(t, f"gemm_f32.{name}", torch.float32)
    check(lib.cid_gemm_f32(_p(x), _p(w), _p(bias), _p(res), _p(out), M, N, c, taps, ldx if ldx is not None else c,
                           ldo if ldo is not None else N, ldr if ldr is not None else N, Hi, Wi, Hi << up, Wi << up, up,
                           _stream()), "cid_gemm_f32")
    return out


def groupnorm_f32(x: torch.Tensor, out: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, ws: torch.Tensor, *, B: int,
                  HW: int, C_: int, groups: int = 32, eps: float = 1e-6, silu: bool = True):
    lib = _lib.load()
    for name, t in (("x", x), ("out", out), ("gamma", gamma), ("beta", beta)):
        _req(t, f"groupnorm_f32.{name}", torch.float32)
    if ws.numel() * ws.element_size() < int(lib.cid_groupnorm_f32_ws_bytes(B, HW, C_)):
        raise _lib.CidError("groupnorm_f32: workspace too small")
    check(lib.cid_groupnorm_f32(_p(x), _p(out), _p(gamma), _p(beta), B, HW, C_, groups, eps, 1 if silu else 0, ws.data_ptr(),
                                _stream()), "cid_groupnorm_f32")
    return out


def groupnorm_f32_ws_bytes(B: int, HW: int, C_: int) -> int:
    return int(_lib.load().cid_groupnorm_f32_ws_bytes(B, HW, C_))


def softmax_rows_f32(x: torch.Tensor, *, rows: int, cols: int, ld: int):
    lib = _lib.load()
    _req(x, "softmax_rows_f32.x", torch.float32)
    check(lib.cid_softmax_rows_f32(_p(x), rows, cols, ld, _stream()), "cid_softmax_rows_f32")
    return x
