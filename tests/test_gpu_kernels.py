"""Parity of every HIP kernel (through the C ABI) against fp32 CPU references built from the
same fp16-representable inputs.  Tolerance: conftest.TOL_L2 / TOL_MAX (fp16 hot path)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import check_close, check_vs_fp16_arm, rel_l2

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,bias,res", [
    (300, 320, 320, True, True),        # 64x160 tiles, ragged M
    (6400, 640, 128, True, False),      # 128x160 tiles
    (25600, 320, 64, False, True),      # 128x320 tiles
    (100, 96, 64, True, False),         # 64x64 tiles (odd width)
    (2048, 1280, 2560, True, True),     # deep K, split-K
    (512, 1280, 11520, True, False),    # 8x8 level conv-sized K, split-K 8
])
def test_gemm_linear(dev, M, N, K, bias, res):
    from consistentid_amd import ops
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    b = rnd(N, seed=3) if bias else None
    r = rnd(M, N, seed=4) if res else None
    ref = x.float() @ w.float().T
    if bias:
        ref = ref + b.float()
    if res:
        ref = ref + r.float()
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    ops.gemm(x.to(dev), w.to(dev), out, M=M, N=N, c1=K, bias=b.to(dev) if bias else None,
             res=r.to(dev) if res else None, ws=ws)
    torch.cuda.synchronize()
    check_close(out, ref, f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("M,C", [(512, 64), (25600, 64), (3000, 320),
                                 # the UNet's own shapes: N-loop launches (one workgroup walks 10 / 5 / 2 n-tiles of its token tile)
                                 (32768, 320), (8192, 640), (2048, 1280), (4096 + 128, 320)])
def test_gemm_geglu(dev, M, C):
    from consistentid_amd import ops, weights
    x, w, b = rnd(M, C, seed=1), rnd(8 * C, C, seed=2, scale=C ** -0.5), rnd(8 * C, seed=3)
    h, gate = (x.float() @ w.float().T + b.float()).chunk(2, dim=-1)
    ref = h * F.gelu(gate)
    out = torch.empty(M, 4 * C, dtype=torch.float16, device=dev)
    ops.gemm(x.to(dev), weights._geglu_interleave(w).contiguous().to(dev), out, M=M, N=8 * C, c1=C,
             bias=weights._geglu_interleave(b).contiguous().to(dev), mode=1)
    torch.cuda.synchronize()
    check_close(out, ref, f"geglu M={M} C={C}")


@pytest.mark.parametrize("M,C,bias", [(65536, 320, True),     # 256 workgroups of linear_h32.hip walking all 16 n-tiles (CFG batch 16)
                                      (16384, 640, False),    # SDXL's 64 x 64 level at CFG batch 4, no bias
                                      (2048, 1280, True)])    # two n-tiles per workgroup, twenty channel slabs
def test_gemm_geglu_h32(dev, M, C, bias):
    """The GEGLU launches of linear_h32.hip (32 x 32 x 16 MFMA tiles, wave roles, N-loop, in-lane value / gate pairing and the
    lane-32 swap in front of the 16-byte stores): against the fp32 formula, bit-identical run to run, and within rounding of
    the 16 x 16 x 32 kernel it replaces on these shapes (CID_GEGLU_H32=0 is read once per process, so that kernel is reached
    through a shape it still owns: the same rows with a ragged tail)."""
    from consistentid_amd import ops, weights
    x, w, b = rnd(M, C, seed=1), rnd(8 * C, C, seed=2, scale=C ** -0.5), rnd(8 * C, seed=3)
    pre = x.float() @ w.float().T + (b.float() if bias else 0.0)
    h, gate = pre.chunk(2, dim=-1)
    ref = h * F.gelu(gate)
    wi = weights._geglu_interleave(w).contiguous().to(dev)
    bi = weights._geglu_interleave(b).contiguous().to(dev) if bias else None
    outs = []
    for _ in range(2):
        out = torch.full((M, 4 * C), float("nan"), dtype=torch.float16, device=dev)
        ops.gemm(x.to(dev), wi, out, M=M, N=8 * C, c1=C, bias=bi, mode=1)
        torch.cuda.synchronize()
        outs.append(out)
    check_close(outs[0], ref, f"geglu (linear_h32) M={M} C={C} bias={bias}")
    assert torch.equal(outs[0], outs[1])
    # the older kernel on the same rows (M + 64 is off the 256-token grid): same values up to the fp32 summation order
    xr = torch.cat([x, x[:64]]).to(dev)
    old = torch.empty(M + 64, 4 * C, dtype=torch.float16, device=dev)
    ops.gemm(xr, wi, old, M=M + 64, N=8 * C, c1=C, bias=bi, mode=1)
    torch.cuda.synchronize()
    d = (old[:M].float() - outs[0].float()).abs().max() / ref.abs().max()
    assert float(d) < 2e-3, float(d)


@pytest.mark.parametrize("M,C,mode,heads,shift", [
    (32768, 320, 0, 0, 0.0),      # q projection, 256-token tiles
    (8192, 640, 0, 0, 2.0),       # rows with a large mean: the mean term must cancel
    (300, 1280, 0, 0, 0.0),       # ragged M, 64-token tiles, 20 slabs
    (8192, 320, 1, 0, 0.5),       # norm3 + GEGLU
    (2048, 1280, 1, 0, 0.0),
    (2 * 4096, 320, 2, 8, 0.5),   # norm1 + fused q/k/v (transposed V epilogue takes the statistics across lanes)
    (2 * 256, 1280, 2, 8, 0.0),
    (2 * 128, 64, 2, 2, 0.0),     # tiny-UNet width (64 x 64 tiles)
])
def test_gemm_layernorm_fold(dev, M, C, mode, heads, shift):
    """cid_gemm_desc.ln_s / ln_b: LayerNorm folded into the projection -- the GEMM reads the raw residual stream, takes
    mean / rstd of every row from the activation fragments on their way to the MFMAs and applies
    rstd * (acc - mean * s) + b' in the epilogue.  Reference: F.layer_norm then the linear, fp64; arm: the same in stock fp16."""
    from consistentid_amd import ops, weights
    x = (rnd(M, C, seed=1, scale=1.3).float() + shift).half()
    g, be = (1 + 0.2 * rnd(C, seed=2).float()).half(), rnd(C, seed=3, scale=0.2)
    N = {0: C, 1: 8 * C, 2: 3 * C}[mode]
    w = rnd(N, C, seed=4, scale=C ** -0.5)
    bias = rnd(N, seed=5, scale=0.3) if mode != 2 else None
    ln = F.layer_norm(x.double(), (C,), g.double(), be.double(), 1e-5)
    lin = ln @ w.double().T + (bias.double() if bias is not None else 0)
    xh = x.to(dev)
    lnh = F.layer_norm(xh, (C,), g.to(dev), be.to(dev), 1e-5)
    linh = lnh @ w.to(dev).T + (bias.to(dev) if bias is not None else 0)
    if mode == 1:
        wi, bi = weights._geglu_interleave(w.float()), weights._geglu_interleave(bias.float())
        wl, s_, b_ = weights.fold_ln(wi.to(dev), g.to(dev), be.to(dev), bi.to(dev))
        out = torch.empty(M, 4 * C, dtype=torch.float16, device=dev)
        ops.gemm(xh, wl, out, M=M, N=N, c1=C, mode=1, ln=(s_.view(torch.float32), b_.view(torch.float32), 1e-5))
        torch.cuda.synchronize()
        h_, gate = lin.chunk(2, -1)
        ha, ga = linh.chunk(2, -1)
        check_vs_fp16_arm(out, h_ * F.gelu(gate), ha * F.gelu(ga), f"LN-folded GEGLU M={M} C={C}")
        return
    wl, s_, b_ = weights.fold_ln(w.float().to(dev), g.to(dev), be.to(dev), bias.to(dev) if bias is not None else None)
    lnp = (s_.view(torch.float32), b_.view(torch.float32), 1e-5)
    if mode == 0:
        res = rnd(M, N, seed=6)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        ops.gemm(xh, wl, out, M=M, N=N, c1=C, ln=lnp, res=res.to(dev), ldr=N)
        torch.cuda.synchronize()
        check_vs_fp16_arm(out, lin + res.double(), linh + res.to(dev), f"LN-folded linear M={M} C={C} shift={shift}")
        return
    d = C // heads
    B, Ntok = 2, M // 2
    qk = torch.empty(M, 2 * C, dtype=torch.float16, device=dev)
    vt = torch.zeros(B * heads * ops.dvp_of(d) * Ntok, dtype=torch.float16, device=dev)
    ops.gemm(xh, wl, qk, M=M, N=N, c1=C, mode=2, vt=vt, n_vt0=2 * C, heads=heads, dhead=d, ntok=Ntok, ln=lnp)
    torch.cuda.synchronize()
    check_vs_fp16_arm(qk, lin[:, :2 * C], linh[:, :2 * C], f"LN-folded q/k projection M={M} C={C}")
    v = lin[:, 2 * C:].reshape(B, Ntok, heads, d).transpose(1, 2)
    va = linh[:, 2 * C:].reshape(B, Ntok, heads, d).transpose(1, 2)
    t = torch.arange(Ntok)
    pos = (t & ~15) | (8 * ((t >> 2) & 1) + 4 * ((t >> 3) & 1) + (t & 3))
    got_v = vt.reshape(B, heads, ops.dvp_of(d), Ntok)[:, :, :d, :].cpu()[..., pos].transpose(-1, -2)
    check_vs_fp16_arm(got_v, v, va, f"LN-folded v^T image M={M} C={C}")


@pytest.mark.parametrize("B,C1,C2,Cout,H,taps", [
    (8, 320, 0, 320, 64, 9),        # halo conv, 256-token tiles, 10-channel units
    (8, 320, 320, 320, 64, 9),      # concat source through the halo kernel
    (8, 640, 0, 640, 32, 9),        # 32 x 32 level: conv3x3.hip on 128-token tiles (statistics blocks of 128 rows)
    (8, 320, 320, 640, 32, 9),      # ... two sources (ten channel slabs: the longest K the 128-token tiles take)
    (8, 640, 320, 640, 32, 1),      # 128-token tiles, 20-channel units, two sources
    (8, 640, 0, 640, 32, 1),        # proj_out-like linear with residual
    (2, 320, 0, 1280, 32, 1),       # 40-channel units
    (3, 64, 0, 320, 32, 1),         # 64-token tiles, odd sample count
])
def test_gemm_emits_groupnorm_statistics(dev, B, C1, C2, Cout, H, taps):
    """cid_gemm_desc.gn_stats: the plain epilogue emits (sum, sum of squares) of the fp16 outputs per block of tokens and
    unit of Cout / 32 channels; cid_groupnorm_stats_f16 folds them instead of running its own statistics pass.  Checks the
    raw statistics against the written tensor, and the GroupNorm built on them against torch's (fp64)."""
    from consistentid_amd import ops
    HW = H * H
    M = B * HW
    x1, x2 = rnd(M, C1, seed=1), (rnd(M, C2, seed=2) if C2 else None)
    K = taps * (C1 + C2)
    w, bias, res = rnd(Cout, K, seed=3, scale=K ** -0.5), rnd(Cout, seed=4), rnd(M, Cout, seed=5)
    out = torch.empty(M, Cout, dtype=torch.float16, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    kw = dict(taps=9, Hi=H, Wi=H, Ho=H, Wo=H) if taps == 9 else {}
    ops.gemm(x1.to(dev), w.to(dev), out, M=M, N=Cout, c1=C1, x2=x2.to(dev) if C2 else None, c2=C2, bias=bias.to(dev),
             res=res.to(dev), ldr=Cout, ws=ws, gn_hw=HW, **kw)
    torch.cuda.synchronize()
    assert hasattr(out, "_gn_stats"), "this launch was expected to emit statistics"
    st, rows = out._gn_stats
    u = Cout // 32
    o = out.double().cpu().reshape(M // rows, rows, 32, u)
    want = torch.stack([o.sum((1, 3)), (o * o).sum((1, 3))], -1)
    err = ((st.double().cpu() - want).abs() / (want.abs() + rows * u * 1e-3)).max()
    assert err < 2e-5, f"statistics differ from the tensor they describe: {err:.2e}"
    g, be = (1 + 0.1 * rnd(Cout, seed=6).float()).half(), rnd(Cout, seed=7, scale=0.1)
    ref = F.silu(F.group_norm(out.double().cpu().reshape(B, HW, Cout).transpose(1, 2), 32, g.double(), be.double(), 1e-5)).transpose(1, 2)
    gws = torch.zeros(ops.groupnorm_ws_bytes(B, Cout), dtype=torch.uint8, device=dev)
    y = torch.empty_like(out)
    ops.groupnorm(out, y, g.to(dev), be.to(dev), gws, B=B, HW=HW, c1=Cout, groups=32, eps=1e-5, silu=True)
    plain = out.clone()                      # same values, no statistics attached: the two-launch path
    y2 = torch.empty_like(out)
    ops.groupnorm(plain, y2, g.to(dev), be.to(dev), gws, B=B, HW=HW, c1=Cout, groups=32, eps=1e-5, silu=True)
    torch.cuda.synchronize()
    check_close(y.reshape(B, HW, Cout), ref, f"GroupNorm on epilogue statistics B{B} {C1}+{C2}->{Cout}")
    assert rel_l2(y, y2) < 2e-4
    # two sources with statistics of their own (skip concat 2 x Cout -> groups of 2 units)
    y3 = torch.empty(M, 2 * Cout, dtype=torch.float16, device=dev)
    g2, b2 = torch.cat([g, g]).to(dev), torch.cat([be, be]).to(dev)
    ops.groupnorm(out, y3, g2, b2, gws, B=B, HW=HW, c1=Cout, x2=out, c2=Cout, groups=32, eps=1e-5, silu=False)
    ref3 = F.group_norm(torch.cat([out, out], -1).double().cpu().reshape(B, HW, 2 * Cout).transpose(1, 2), 32,
                        g2.double().cpu(), b2.double().cpu(), 1e-5).transpose(1, 2)
    torch.cuda.synchronize()
    check_close(y3.reshape(B, HW, 2 * Cout), ref3, f"GroupNorm on two sources' epilogue statistics B{B} Cout={Cout}")
    for _ in range(3):                        # bit-stable (fixed summation order, no atomics)
        o2, yb = torch.empty_like(out), torch.empty_like(out)
        ops.gemm(x1.to(dev), w.to(dev), o2, M=M, N=Cout, c1=C1, x2=x2.to(dev) if C2 else None, c2=C2, bias=bias.to(dev),
                 res=res.to(dev), ldr=Cout, ws=ws, gn_hw=HW, **kw)
        ops.groupnorm(o2, yb, g.to(dev), be.to(dev), gws, B=B, HW=HW, c1=Cout, groups=32, eps=1e-5, silu=True)
        torch.cuda.synchronize()
        assert torch.equal(o2._gn_stats[0], st) and torch.equal(yb, y)


def _tok(x):   # NCHW -> [B*HW, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


@pytest.mark.parametrize("B,C1,C2,Cout,H,stride,up", [
    (2, 64, 0, 64, 16, 1, 0),
    (2, 64, 0, 96, 16, 2, 0),
    (2, 64, 0, 64, 8, 1, 1),
    (1, 64, 64, 160, 12, 1, 0),       # skip concat via two sources
    (2, 320, 0, 320, 64, 1, 0),       # SD1.5 level-0 shape, 128x320 tiles need >= 200 tiles -> B*HW=8192/128*1 = 64 -> mid tiles
    (8, 320, 0, 320, 64, 1, 0),       # 256x160 tiles
    (8, 640, 0, 640, 32, 1, 0),       # 32 x 32 level: 128-token tiles of conv3x3.hip (256 of them)
    (8, 1280, 0, 1280, 16, 1, 0),     # 16 x 16 level: a tile = one whole image, split-K 4 (gemm.hip halo kernel)
    (8, 1280, 640, 1280, 16, 1, 0),   # ... with a skip concat (30 channel slabs over four slices: uneven)
    (4, 320, 0, 320, 64, 1, 0),       # CFG-deduplicated level 0: 128-token tiles, two image rows each
    (8, 320, 320, 320, 64, 1, 0),     # 256-token tiles of conv3x3.hip, two sources
    (8, 640, 0, 640, 32, 1, 1),       # Upsample2D conv 32 -> 64 (level 0): conv3x3.hip with an input-resolution halo, 512 tiles
    (8, 1280, 0, 1280, 16, 1, 1),     # ... 16 -> 32 (level 1): a tile = eight output rows, four input rows
    (2, 320, 0, 640, 32, 1, 1),       # ... on 128-token tiles (two output rows, one input row + frame)
    (3, 320, 320, 320, 32, 1, 1),     # ... two sources, 96 tiles x 2 (odd sample count)
    (8, 1280, 1280, 1280, 8, 1, 0),   # 8x8 level, concat, split-K
])
def test_gemm_conv3x3(dev, B, C1, C2, Cout, H, stride, up):
    from consistentid_amd import ops, weights
    Wd = H
    x1 = rnd(B, C1, H, Wd, seed=1)
    x2 = rnd(B, C2, H, Wd, seed=2) if C2 else None
    w = rnd(Cout, C1 + C2, 3, 3, seed=3, scale=(9 * (C1 + C2)) ** -0.5)
    b = rnd(Cout, seed=4)
    temb = rnd(B, Cout, seed=5)
    xin = torch.cat([x1, x2], 1).float() if C2 else x1.float()
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.float(), b.float(), stride=stride, padding=1) + temb.float()[:, :, None, None]
    Ho, Wo = ref.shape[-2:]
    M = B * Ho * Wo
    out = torch.empty(M, Cout, dtype=torch.float16, device=dev)
    ops.gemm(_tok(x1).to(dev), weights._conv3(w, dev), out, M=M, N=Cout, c1=C1,
             x2=_tok(x2).to(dev) if C2 else None, c2=C2, bias=b.to(dev), rowbias=temb.to(dev), ld_rowbias=Cout,
             rows_per_sample=Ho * Wo, taps=9, Hi=H, Wi=Wd, Ho=Ho, Wo=Wo, stride=stride, up=up,
             ws=torch.empty(64 << 20, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    check_close(out, _tok(ref), f"conv3x3 B{B} C{C1}+{C2}->{Cout} H{H} s{stride} up{up}")


# ----------------------------------------------------------------------------- self attention
@pytest.mark.parametrize("B,N,C,heads", [
    (2, 256, 320, 8),     # d=40, QT=2
    (1, 4096, 320, 8),    # SD1.5 level 0
    (2, 1024, 640, 8),    # d=80
    (2, 256, 1280, 8),    # d=160
    (2, 64, 1280, 8),     # mid block
    (1, 1024, 640, 10),   # SDXL d=64
    (2, 128, 64, 2),      # d=32 (tiny UNet)
])
def test_qkv_gemm_and_self_attention(dev, B, N, C, heads):
    from consistentid_amd import ops
    from consistentid_amd.weights import LOG2E
    d = C // heads
    x = rnd(B, N, C, seed=1)
    wq, wk, wv = (rnd(C, C, seed=s, scale=sc * C ** -0.5) for s, sc in ((2, 2.0), (3, 2.0), (4, 1.0)))
    q, k, v = (x.float() @ w.float().T for w in (wq, wk, wv))
    sp = lambda t: t.reshape(B, N, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, N, C)
    xh = x.to(dev)
    qh, kh, vh = (xh @ w.to(dev).T for w in (wq, wk, wv))          # stock fp16 arm
    arm = F.scaled_dot_product_attention(sp(qh), sp(kh), sp(vh)).transpose(1, 2).reshape(B, N, C)
    wqkv = torch.cat([wq.float() * (d ** -0.5 * LOG2E), wk.float(), wv.float()], 0).half().to(dev)
    M = B * N
    qk = torch.empty(M, 2 * C, dtype=torch.float16, device=dev)
    vt = torch.zeros(B * heads * ops.dvp_of(d) * N, dtype=torch.float16, device=dev)
    ops.gemm(x.to(dev), wqkv, qk, M=M, N=3 * C, c1=C, mode=2, vt=vt, n_vt0=2 * C, heads=heads, dhead=d, ntok=N)
    torch.cuda.synchronize()
    # the projection itself (q is pre-scaled)
    check_close(qk[:, C:], k.reshape(M, C), f"k-proj C={C}")
    # V^T image: vt[b][h][dd][pos(t)]
    vt_ref = sp(v)                                             # [B, heads, N, d]
    t = torch.arange(N)
    pos = (t & ~15) | (8 * ((t >> 2) & 1) + 4 * ((t >> 3) & 1) + (t & 3))
    got_v = vt.reshape(B, heads, ops.dvp_of(d), N)[:, :, :d, :].cpu()[..., pos].transpose(-1, -2)
    check_close(got_v, vt_ref, f"v^T image C={C}")
    out = torch.empty(M, C, dtype=torch.float16, device=dev)
    ops.self_attn(qk, qk[:, C:], vt, out, B=B, N=N, heads=heads, d=d, ldq=2 * C, ldk=2 * C, ldo=C)
    torch.cuda.synchronize()
    check_vs_fp16_arm(out.reshape(B, N, C), ref, arm, f"self-attn N={N} C={C} d={d}")


def test_self_attention_online_softmax_rescale(dev):
    """Force the running-max rescale branch: one key far above the rest, late in the sequence."""
    from consistentid_amd import ops
    B, N, heads, d = 1, 256, 1, 64
    q, k, v = rnd(B, N, d, seed=1), rnd(B, N, d, seed=2), rnd(B, N, d, seed=3)
    k[0, 200] = (q[0, 7].float() * 3).half()       # spikes query 7 (and others) at key tile 3
    k[0, 3] = (q[0, 100].float() * 2).half()
    ref = F.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None], scale=1.0)[:, 0]
    t = torch.arange(N)
    pos = (t & ~15) | (8 * ((t >> 2) & 1) + 4 * ((t >> 3) & 1) + (t & 3))
    vt = torch.zeros(B, heads, 64, N, dtype=torch.float16)
    vt[0, 0][:, pos] = v[0].T
    from consistentid_amd.weights import LOG2E
    qs = (q.float() * LOG2E).half()
    ref = F.scaled_dot_product_attention(qs.float()[:, None] / LOG2E, k.float()[:, None], v.float()[:, None], scale=1.0)[:, 0]
    out = torch.empty(B * N, d, dtype=torch.float16, device=dev)
    ops.self_attn(qs.reshape(N, d).to(dev), k.reshape(N, d).to(dev), vt.to(dev), out, B=B, N=N, heads=heads, d=d,
                  ldq=d, ldk=d, ldo=d)
    torch.cuda.synchronize()
    check_close(out.reshape(B, N, d), ref, "self-attn rescale branch", tol_l2=2e-3, tol_max=8e-3)


# ----------------------------------------------------------------------------- norms
@pytest.mark.parametrize("M,C", [(1000, 320), (257, 640), (64, 1280), (128, 64),
                                 (32768, 320), (8200, 1280), (9000, 1024), (50, 200)])      # several rows per wave: 8 lanes per row, 16 (ragged M), 16 at 128 chunks; small launch
def test_layernorm(dev, M, C):
    from consistentid_amd import ops
    x, g, b = rnd(M, C, seed=1, scale=2.0), (1 + 0.1 * rnd(C, seed=2).float()).half(), rnd(C, seed=3, scale=0.1)
    x = (x.float() + 0.5).half()
    ref = F.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5)
    out = torch.empty(M, C, dtype=torch.float16, device=dev)
    ops.layernorm(x.to(dev), out, g.to(dev), b.to(dev), M=M, C_=C)
    torch.cuda.synchronize()
    check_close(out, ref, f"layernorm {M}x{C}")


@pytest.mark.parametrize("B,HW,C1,C2,silu,eps", [
    (2, 4096, 320, 0, True, 1e-5),
    (2, 1024, 640, 0, False, 1e-6),
    (2, 256, 1280, 640, True, 1e-5),    # groups straddle the concat boundary (60 ch / group)
    (1, 64, 1280, 1280, True, 1e-5),    # C = 2560 (> 256 chunk columns)
    (3, 100, 64, 0, True, 1e-5),        # ragged row count
    (8, 256, 1280, 1280, True, 1e-5),   # single-launch path (a (sample, group) slice in one workgroup's registers), XCD remap
    (16, 64, 1280, 0, True, 1e-5),      # ... 8 x 8 level, two samples per XCD
    (8, 1024, 640, 0, False, 1e-6),     # ... its largest slice: 1024 pixels x 20 channels
    (3, 256, 128, 0, True, 1e-5),       # ... 4 channels per group, sample count not a multiple of 8
])
def test_groupnorm(dev, B, HW, C1, C2, silu, eps):
    from consistentid_amd import ops
    C = C1 + C2
    x1 = (rnd(B, HW, C1, seed=1, scale=1.5).float() + 0.3).half()
    x2 = rnd(B, HW, C2, seed=2) if C2 else None
    g, b = (1 + 0.1 * rnd(C, seed=3).float()).half(), rnd(C, seed=4, scale=0.1)
    xin = torch.cat([x1, x2], -1).float() if C2 else x1.float()
    ref = F.group_norm(xin.transpose(1, 2), 32, g.float(), b.float(), eps).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    out = torch.empty(B * HW, C, dtype=torch.float16, device=dev)
    ws = torch.zeros(ops.groupnorm_ws_bytes(B, C), dtype=torch.uint8, device=dev)
    ops.groupnorm(x1.to(dev), out, g.to(dev), b.to(dev), ws, B=B, HW=HW, c1=C1, x2=x2.to(dev) if C2 else None,
                  c2=C2, groups=32, eps=eps, silu=silu)
    torch.cuda.synchronize()
    check_close(out.reshape(B, HW, C), ref, f"groupnorm B{B} HW{HW} C{C1}+{C2} silu={silu}")
    # the workspace is reusable as it is and the result is bit-stable (fixed reduction order, no atomics)
    for _ in range(5):
        out2 = torch.empty_like(out)
        ops.groupnorm(x1.to(dev), out2, g.to(dev), b.to(dev), ws, B=B, HW=HW, c1=C1, x2=x2.to(dev) if C2 else None,
                      c2=C2, groups=32, eps=eps, silu=silu)
        torch.cuda.synchronize()
        assert torch.equal(out, out2)


# ----------------------------------------------------------------------------- UNet ends / time path / glue
@pytest.mark.parametrize("cin", [4, 9, 6])      # 6: the generic (runtime channel split) instance of conv_in
def test_conv_in_out(dev, cin):
    from consistentid_amd import ops
    Bin, B, H, W, c = 2, 4, 16, 24, 320
    s, w, b = rnd(Bin, cin, H, W, seed=1), rnd(c, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(c, seed=3)
    ref = F.conv2d(torch.cat([s, s]).float(), w.float(), b.float(), padding=1)
    out = torch.empty(B * H * W, c, dtype=torch.float16, device=dev)
    ops.conv_in(s.to(dev), out, w.permute(0, 2, 3, 1).reshape(c, -1).contiguous().to(dev), b.to(dev),
                B=B, Bin=Bin, cin=cin, H=H, W=W, cout=c)
    torch.cuda.synchronize()
    check_close(out, _tok(ref), f"conv_in cin={cin}")
    x, w2, b2 = rnd(B, c, H, W, seed=4), rnd(4, c, 3, 3, seed=5, scale=(9 * c) ** -0.5), rnd(4, seed=6)
    ref2 = F.conv2d(x.float(), w2.float(), b2.float(), padding=1)
    out2 = torch.empty(B, 4, H, W, dtype=torch.float16, device=dev)
    ops.conv_out(_tok(x).to(dev), out2, w2.permute(0, 2, 3, 1).reshape(4, -1).contiguous().to(dev), b2.to(dev),
                 B=B, H=H, W=W, cin=c, cout=4)
    torch.cuda.synchronize()
    check_close(out2, ref2, "conv_out")


@pytest.mark.parametrize("B,H,W,c,cout", [(1, 5, 7, 64, 4),        # 35 pixels: a ragged last pixel group, one channel round
                                          (3, 9, 13, 192, 3),      # 24 chunks over 8 lanes, three outputs
                                          (8, 64, 64, 320, 4)])    # the SD1.5 level-0 shape of the denoise step
def test_conv_out_shapes(dev, B, H, W, c, cout):
    """cid_conv_out_f16 at shapes off its pixel-pair / eight-lane grid, and at the step's own shape; bit-reproducible"""
    from consistentid_amd import ops
    x, w, b = rnd(B, c, H, W, seed=14), rnd(cout, c, 3, 3, seed=15, scale=(9 * c) ** -0.5), rnd(cout, seed=16)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    xt, wt = _tok(x).to(dev), w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().to(dev)
    outs = []
    for _ in range(2):
        out = torch.full((B, cout, H, W), float("nan"), dtype=torch.float16, device=dev)
        ops.conv_out(xt, out, wt, b.to(dev), B=B, H=H, W=W, cin=c, cout=cout)
        torch.cuda.synchronize()
        outs.append(out)
    check_close(outs[0], ref, f"conv_out B={B} {H}x{W} cin={c} cout={cout}")
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K,taps,H", [(8192, 320, 320, 1, 0),       # 256-token tiles, plain epilogue
                                           (2048, 640, 640, 1, 0),       # 128-token tiles
                                           (512, 1280, 1280, 1, 0),      # 64-token tiles / split-K epilogue
                                           (2 * 64 * 64, 320, 320, 9, 64),   # conv3x3.hip, 256-token tiles
                                           (2 * 32 * 32, 640, 640, 9, 32)])  # conv3x3.hip, 128-token tiles
def test_gemm_second_destination(dev, M, N, K, taps, H):
    """cid_gemm_desc.out2: the producer writes its rows twice (the CFG duplication of a tensor both halves share) -- both
    copies bit-identical to the single-destination launch, on every epilogue that stores rows"""
    from consistentid_amd import ops
    x, w, b, r = rnd(M, K, seed=1), rnd(N, taps * K, seed=2, scale=(taps * K) ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    kw = dict(M=M, N=N, c1=K, bias=b.to(dev), res=r.to(dev), ws=ws)
    if taps == 9:
        kw.update(taps=9, Hi=H, Wi=H, Ho=H, Wo=H)
    one = torch.empty(M, N, dtype=torch.float16, device=dev)
    ops.gemm(x.to(dev), w.to(dev), one, **kw)
    two = torch.full((2 * M, N), float("nan"), dtype=torch.float16, device=dev)
    ops.gemm(x.to(dev), w.to(dev), two[:M], out2=two[M:], **kw)
    torch.cuda.synchronize()
    assert torch.equal(two[:M], one) and torch.equal(two[M:], one)
    with pytest.raises(Exception):
        ops.gemm(x.to(dev), rnd(2 * N, K, seed=5).to(dev), one, M=M, N=2 * N, c1=K, mode=1, out2=two[M:])   # GEGLU: no out2


def test_conv_in_two_sources(dev):
    """cid_conv_in_cat_f16: the 9-channel inpainting conv_in reads latents (scaled by in_scale) and cat([mask, masked image
    latents]) (not scaled) from two tensors -- torch.cat([latent_model_input, mask, masked_image_latents], dim=1) of
    pipelines/StableDIffusionInpaint_ConsistentID.py:320-321 without the copy."""
    from consistentid_amd import ops
    Bin, B, H, W, c = 2, 4, 16, 24, 320
    lat, ext = rnd(Bin, 4, H, W, seed=1), rnd(Bin, 5, H, W, seed=7)
    w, b = rnd(c, 9, 3, 3, seed=2, scale=81 ** -0.5), rnd(c, seed=3)
    wt = w.permute(0, 2, 3, 1).reshape(c, -1).contiguous().to(dev)
    for scale in (None, 0.37):
        x9 = torch.cat([lat.double() * (scale or 1.0), ext.double()], dim=1)
        ref = F.conv2d(torch.cat([x9, x9]), w.double(), b.double(), padding=1)
        out = torch.empty(B * H * W, c, dtype=torch.float16, device=dev)
        sc = None if scale is None else torch.tensor([scale], dtype=torch.float32, device=dev)
        ops.conv_in(lat.to(dev), out, wt, b.to(dev), B=B, Bin=Bin, cin=9, H=H, W=W, cout=c, in_scale=sc, extra=ext.to(dev))
        torch.cuda.synchronize()
        check_close(out, _tok(ref), f"conv_in 4 + 5 channels, in_scale={scale}")
    # one 9-channel source is the same convolution (the diffusers-style call with a concatenated sample)
    one = torch.empty_like(out)
    ops.conv_in(torch.cat([lat, ext], 1).contiguous().to(dev), one, wt, b.to(dev), B=B, Bin=Bin, cin=9, H=H, W=W, cout=c)
    two = torch.empty_like(out)
    ops.conv_in(lat.to(dev), two, wt, b.to(dev), B=B, Bin=Bin, cin=9, H=H, W=W, cout=c, extra=ext.to(dev))
    torch.cuda.synchronize()
    assert rel_l2(one, two) < 2e-4
    with pytest.raises(Exception):
        ops.conv_in(lat.to(dev), two, wt, b.to(dev), B=B, Bin=Bin, cin=9, H=H, W=W, cout=c, extra=ext[:, :3].contiguous().to(dev))


def test_time_path(dev):
    from consistentid_amd import ops
    from oracle.unet import timestep_embedding
    v = torch.tensor([981.0, 1.0, 1024.0, 0.0, 37.0])
    ref = timestep_embedding(v, 320)
    out = torch.empty(5, 320, dtype=torch.float16, device=dev)
    ops.sincos_embed(v.to(dev), out, rows=5, dim=320)
    torch.cuda.synchronize()
    check_close(out, ref, "sincos", tol_l2=2e-3, tol_max=4e-3)
    for M in (1, 5, 11):
        x, w, b, add = rnd(M, 1280, seed=1), rnd(1000, 1280, seed=2, scale=1280 ** -0.5), rnd(1000, seed=3), rnd(1, 1000, seed=4)
        ref = F.silu(F.silu(x.float()) @ w.float().T + b.float() + add.float())
        o = torch.empty(M, 1000, dtype=torch.float16, device=dev)
        ops.linear_small(x.to(dev), w.to(dev), b.to(dev), o, M=M, N=1000, K=1280, add=add.to(dev), ldadd=0,
                         act_in=1, act_out=1)
        torch.cuda.synchronize()
        check_close(o, ref, f"linear_small M={M}")


def test_cfg_ddim_and_blend_and_add(dev):
    from consistentid_amd import ops
    B, per = 3, 4 * 16 * 16
    eps, lat = rnd(2 * B, per, seed=1), rnd(B, per, seed=2)
    coef = torch.tensor([1.01, -0.07, 0.9, 0.43])
    g = 5.0
    e = eps[:B].float() + g * (eps[B:].float() - eps[:B].float())
    ref = coef[0] * lat.float() + coef[1] * e
    l1 = lat.clone().to(dev)
    ops.cfg_ddim_step(eps.to(dev), l1, coef.to(dev), g, B=B, per_sample=per)
    torch.cuda.synchronize()
    check_close(l1, ref, "cfg+ddim")
    mask = (torch.rand(B, per, generator=torch.Generator().manual_seed(3)) > 0.5).half()
    init, noise = rnd(B, per, seed=4), rnd(B, per, seed=5)
    ref2 = (1 - mask.float()) * (coef[2] * init.float() + coef[3] * noise.float()) + mask.float() * ref
    l2 = lat.clone().to(dev)
    ops.cfg_ddim_step(eps.to(dev), l2, coef.to(dev), g, B=B, per_sample=per, mask=mask.to(dev), init=init.to(dev),
                      noise=noise.to(dev))
    torch.cuda.synchronize()
    check_close(l2, ref2, "cfg+ddim+inpaint blend")
    y, a = rnd(4, 64, 32, seed=6), rnd(2, 64, 32, seed=7)
    yd = y.clone().to(dev)
    ops.add_inplace(yd, a.to(dev))
    torch.cuda.synchronize()
    check_close(yd, y.float() + torch.cat([a, a]).float(), "add_inplace (broadcast over CFG halves)")


def test_step_select_table(dev):
    """cid_step_select: row `counter` of the per-step device table -> the buffers a captured step reads, counter += 1
    (the reference's ``for i, t in enumerate(timesteps)`` with i, t, the scheduler coefficients and the embed-set choice
    as host values, pipline_StableDiffusion_ConsistentID.py:535-549).  Bit-exact copies of mixed dtypes, start row,
    clamping at the last row, and the same launch replayed from a hipGraph."""
    from consistentid_amd import ops
    S = 7
    g = torch.Generator().manual_seed(11)
    t_vals = torch.rand(S, 1, generator=g)
    coef = torch.rand(S, 5, generator=g)
    rows = torch.randint(0, 99, (S, 6), generator=g, dtype=torch.int32)
    temb = torch.randn(S, 1280, generator=g).half()
    t_buf, coef_buf = torch.zeros(1, device=dev), torch.zeros(5, device=dev)
    row_buf, temb_buf = torch.zeros(6, dtype=torch.int32, device=dev), torch.zeros(1, 1280, dtype=torch.float16, device=dev)
    tab = ops.StepTable([(t_buf, t_vals), (coef_buf, coef), (row_buf, rows), (temb_buf, temb)], dev)
    tab.reset(2)
    for i in range(2, S + 2):          # two launches past the end: clamped to the last row
        tab.select()
        torch.cuda.synchronize()
        j = min(i, S - 1)
        assert torch.equal(t_buf.cpu(), t_vals[j]) and torch.equal(coef_buf.cpu(), coef[j])
        assert torch.equal(row_buf.cpu(), rows[j]) and torch.equal(temb_buf.cpu()[0], temb[j])
        assert int(tab.counter.item()) == j + 1
    tab.reset(0)
    tab.select()                        # (warm-up outside the capture)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        tab.select()
    tab.reset(0)
    for i in range(S):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(coef_buf.cpu(), coef[i]) and torch.equal(row_buf.cpu(), rows[i])


# ----------------------------------------------------------------------------- fused ID cross attention
def _xattn_reference(x, ehs, W, heads, n_ip, ip_scale, ln=None, residual=False, arm_device=None):
    """fp32 restatement via the oracle's processor (attention.py:207-294) + optional LN / residual.
    With arm_device: the same modules run in fp16 with stock PyTorch ops on the GPU."""
    from oracle import processors as oproc
    from oracle.unet import Attention
    C, Dc = x.shape[-1], ehs.shape[-1]
    attn = Attention(C, Dc, heads)
    proc = oproc.Consistent_IPAttProcessor(hidden_size=C, cross_attention_dim=Dc, rank=W["rank"], scale=ip_scale,
                                           num_tokens=n_ip)
    with torch.no_grad():
        attn.to_q.weight.copy_(W["q"]); attn.to_k.weight.copy_(W["k"]); attn.to_v.weight.copy_(W["v"])
        attn.to_out[0].weight.copy_(W["o"]); attn.to_out[0].bias.copy_(W["bo"])
        for n in ("q", "k", "v", "out"):
            getattr(proc, f"to_{n}_lora").down.weight.copy_(W[f"{n}_down"])
            getattr(proc, f"to_{n}_lora").up.weight.copy_(W[f"{n}_up"])
        proc.to_k_ip.weight.copy_(W["kip"]); proc.to_v_ip.weight.copy_(W["vip"])
        dt, dv = (torch.float16, arm_device) if arm_device is not None else (torch.float32, "cpu")
        attn, proc = attn.to(dv, dt), proc.to(dv, dt)
        h = x.to(dv, dt)
        if ln is not None:
            h = F.layer_norm(h, (C,), ln[0].to(dv, dt), ln[1].to(dv, dt), 1e-5)
        o = proc(attn, h, encoder_hidden_states=ehs.to(dv, dt))
        if residual:
            o = o + x.to(dv, dt)
    return o


def _xattn_weights(C, Dc, rank, seed):
    f = lambda *s, sc=1.0, sd=0: rnd(*s, seed=seed + sd, scale=sc).float()
    return dict(rank=rank, q=f(C, C, sc=3 * C ** -0.5, sd=1), k=f(C, Dc, sc=3 * Dc ** -0.5, sd=2),
                v=f(C, Dc, sc=Dc ** -0.5, sd=3), o=f(C, C, sc=C ** -0.5, sd=4), bo=f(C, sc=0.1, sd=5),
                kip=f(C, Dc, sc=3 * Dc ** -0.5, sd=6), vip=f(C, Dc, sc=Dc ** -0.5, sd=7),
                q_down=f(rank, C, sc=1 / rank, sd=8), q_up=f(C, rank, sc=0.05, sd=9),
                k_down=f(rank, Dc, sc=1 / rank, sd=10), k_up=f(C, rank, sc=0.05, sd=11),
                v_down=f(rank, Dc, sc=1 / rank, sd=12), v_up=f(C, rank, sc=0.05, sd=13),
                out_down=f(rank, C, sc=1 / rank, sd=14), out_up=f(C, rank, sc=0.05, sd=15))


@pytest.mark.parametrize("B,N,C,heads,Dc,fused", [
    (2, 256, 320, 8, 768, False),
    (2, 4096, 320, 8, 768, True),      # SD1.5 level 0
    (2, 1024, 640, 8, 768, True),
    (8, 1024, 640, 8, 768, True),      # SD1.5 level 1 at BASELINE config 2's CFG batch (128-token tiles in mode 3)
    (2, 256, 1280, 8, 768, True),
    (3, 64, 1280, 8, 768, False),      # mid block
    (1, 1024, 640, 10, 2048, True),    # SDXL
    (1, 256, 1280, 20, 2048, True),
    (2, 128, 64, 2, 128, True),        # tiny UNet widths
    (1, 256, 128, 2, 128, False),
])
def test_id_cross_attention(dev, B, N, C, heads, Dc, fused):
    from consistentid_amd import ops
    from consistentid_amd.weights import LOG2E
    L, n_ip, ip_scale, rank = 81, 4, 0.8, 8
    W = _xattn_weights(C, Dc, rank, seed=C + heads)
    x = rnd(B, N, C, seed=1, scale=1.5)
    ehs = rnd(B + 1, L, Dc, seed=2)      # one spare row: kvrow indirection is exercised
    kvrow = torch.tensor([(i + 1) % (B + 1) for i in range(B)], dtype=torch.int32)
    ln = ((1 + 0.1 * rnd(C, seed=3).float()).half(), rnd(C, seed=4, scale=0.1)) if fused else None
    ref = _xattn_reference(x, ehs[kvrow.long()], W, heads, n_ip, ip_scale, ln, residual=fused)
    arm = _xattn_reference(x, ehs[kvrow.long()], W, heads, n_ip, ip_scale, ln, residual=fused, arm_device=dev)
    d = C // heads
    mq = (W["q"] + W["q_up"] @ W["q_down"]) * (d ** -0.5 * LOG2E)
    mk, mv = W["k"] + W["k_up"] @ W["k_down"], W["v"] + W["v_up"] @ W["v_down"]
    mo = W["o"] + W["out_up"] @ W["out_down"]
    R = B + 1
    kv_txt = torch.empty(R * L, 2 * C, dtype=torch.float16, device=dev)
    kv_ip = torch.empty(R * L, 2 * C, dtype=torch.float16, device=dev)
    e = ehs.to(dev)
    ops.gemm(e, torch.cat([mk, mv]).half().to(dev), kv_txt, M=R * L, N=2 * C, c1=Dc)
    ops.gemm(e, torch.cat([W["kip"], W["vip"]]).half().to(dev), kv_ip, M=R * L, N=2 * C, c1=Dc)
    ke, ve = ops.kv_pack_elems(C, heads)
    kp = torch.empty(R * ke, dtype=torch.float16, device=dev)
    vp = torch.empty(R * ve, dtype=torch.float16, device=dev)
    ops.kv_pack(kv_txt, kv_ip, kp, vp, R=R, C_=C, heads=heads, n_txt=L - n_ip, n_ip=n_ip)
    out = torch.empty(B, N, C, dtype=torch.float16, device=dev)
    xd = x.to(dev)
    ops.id_xattn(xd, out, wq=mq.half().to(dev).contiguous(), wo=mo.half().to(dev).contiguous(),
                 bo=W["bo"].half().to(dev), kp=kp, vp=vp, kvrow=kvrow.to(dev), B=B, N=N, C_=C, heads=heads,
                 n_txt=L - n_ip, n_ip=n_ip, ip_scale=ip_scale, residual=xd if fused else None,
                 ln_gamma=ln[0].to(dev) if fused else None, ln_beta=ln[1].to(dev) if fused else None)
    torch.cuda.synchronize()
    check_vs_fp16_arm(out, ref, arm, f"id-xattn N={N} C={C} heads={heads} fused={fused}")
    if fused and C >= 640:
        # the split path the engine uses for wide levels: LN -> GEMM -> attention core -> GEMM(+bias+residual)
        M = B * N
        ln2 = torch.empty(M, C, dtype=torch.float16, device=dev)
        ops.layernorm(xd, ln2, ln[0].to(dev), ln[1].to(dev), M=M, C_=C)
        q2 = torch.empty(M, C, dtype=torch.float16, device=dev)
        ops.gemm(ln2, mq.half().to(dev).contiguous(), q2, M=M, N=C, c1=C)
        o2 = torch.empty(M, C, dtype=torch.float16, device=dev)
        ops.id_xattn_core(q2, o2, kp=kp, vp=vp, kvrow=kvrow.to(dev), B=B, N=N, C_=C, heads=heads, n_txt=L - n_ip,
                          n_ip=n_ip, ip_scale=ip_scale)
        out2 = torch.empty(B, N, C, dtype=torch.float16, device=dev)
        ops.gemm(o2, mo.half().to(dev).contiguous(), out2, M=M, N=C, c1=C, bias=W["bo"].half().to(dev), res=xd, ldr=C)
        torch.cuda.synchronize()
        check_vs_fp16_arm(out2, ref, arm, f"id-xattn split path N={N} C={C} heads={heads}")
        if ops.qattn_supported(C, heads, N, L - n_ip, n_ip):
            # cid_gemm_f16 mode 3: the query projection runs the attention as its epilogue.  The Q tile is rounded to fp16
            # exactly like the q tensor of the split path and goes through the same unit function: O is bit-identical.
            o3 = torch.empty(M, C, dtype=torch.float16, device=dev)
            att = (kp, vp, kvrow.to(dev), L - n_ip, n_ip, ip_scale)
            ops.gemm(ln2, mq.half().to(dev).contiguous(), o3, M=M, N=C, c1=C, mode=3, heads=heads, dhead=C // heads, ntok=N,
                     att=att)
            torch.cuda.synchronize()
            assert torch.equal(o3, o2), f"attention epilogue differs from GEMM + core (rel {rel_l2(o3, o2):.2e})"
            # ... and with norm2 folded into the projection (what the engine launches at <= 2048 tokens)
            from consistentid_amd import weights
            wl, s_, b_ = weights.fold_ln(mq.half().to(dev), ln[0].to(dev), ln[1].to(dev), None)
            o4 = torch.empty(M, C, dtype=torch.float16, device=dev)
            ops.gemm(xd.view(M, C), wl, o4, M=M, N=C, c1=C, mode=3, heads=heads, dhead=C // heads, ntok=N,
                     ln=(s_.view(torch.float32), b_.view(torch.float32), 1e-5), att=att)
            out4 = torch.empty(B, N, C, dtype=torch.float16, device=dev)
            ops.gemm(o4, mo.half().to(dev).contiguous(), out4, M=M, N=C, c1=C, bias=W["bo"].half().to(dev), res=xd, ldr=C)
            torch.cuda.synchronize()
            check_vs_fp16_arm(out4, ref, arm, f"id-xattn, LN-folded q projection with attention epilogue N={N} C={C} heads={heads}")


# ----------------------------------------------------------------------------- fused ID cross attention, third generation
@pytest.mark.parametrize("B,N,n_ip,has_ln,residual,mean_shift", [
    (2, 4096, 4, True, True, 0.0),       # SD1.5 level 0, the engine's call: LayerNorm folded, residual = x
    (8, 4096, 4, True, True, 0.0),       # BASELINE config 2's CFG batch (512 workgroups, XCD remap active)
    (3, 64, 4, True, True, 3.0),         # one tile per sample, odd tile count (no remap); rows with a large mean
    (1, 6144, 4, True, True, 0.0),       # 512 x 768 (the reference's default inference size), one sample
    (2, 192, 4, False, False, 0.0),      # processor-level call: no LayerNorm, no residual; N % 128 != 0
    (2, 512, 0, True, True, 0.0),        # ControlNet's default attention: 81 plain keys
    (1, 64, 4, True, True, 0.0),         # the smallest launch: one workgroup
    (5, 320, 0, False, True, -2.0),      # 81 plain keys without LayerNorm, residual on, 25 tiles (no XCD remap), shifted rows
    (2, 1024, 4, True, False, 0.0),      # LayerNorm folded, no residual
])
def test_id_cross_attention_v3(dev, B, N, n_ip, has_ln, residual, mean_shift):
    """cid_id_xattn3_f16 (64-token tiles, weights streamed as packed A operands, LayerNorm statistics traded through
    LDS, in-place epilogue transpose) against the oracle processor, same criterion as the other generations."""
    from consistentid_amd import ops, xattn_pack
    from consistentid_amd.weights import LOG2E
    C, heads, Dc, L, ip_scale, rank = 320, 8, 768, 81, 0.8, 8
    n_txt = L - n_ip
    assert ops.id_xattn3_supported(C, heads, n_txt, n_ip)
    W = _xattn_weights(C, Dc, rank, seed=C + heads)
    x = (rnd(B, N, C, seed=1, scale=1.5).float() + mean_shift).half()
    ehs = rnd(B + 1, L, Dc, seed=2)
    kvrow = torch.tensor([(i + 1) % (B + 1) for i in range(B)], dtype=torch.int32)
    ln = ((1 + 0.1 * rnd(C, seed=3).float()).half(), rnd(C, seed=4, scale=0.1)) if has_ln else None
    ref = _xattn_reference(x, ehs[kvrow.long()], W, heads, n_ip, ip_scale, ln, residual=residual)
    arm = _xattn_reference(x, ehs[kvrow.long()], W, heads, n_ip, ip_scale, ln, residual=residual, arm_device=dev)
    d = C // heads
    mq = (W["q"] + W["q_up"] @ W["q_down"]) * (d ** -0.5 * LOG2E)
    mk, mv = W["k"] + W["k_up"] @ W["k_down"], W["v"] + W["v_up"] @ W["v_down"]
    mo = W["o"] + W["out_up"] @ W["out_down"]
    R = B + 1
    kv_txt = torch.empty(R * L, 2 * C, dtype=torch.float16, device=dev)
    kv_ip = torch.empty(R * L, 2 * C, dtype=torch.float16, device=dev)
    e = ehs.to(dev)
    ops.gemm(e, torch.cat([mk, mv]).half().to(dev), kv_txt, M=R * L, N=2 * C, c1=Dc)
    ops.gemm(e, torch.cat([W["kip"], W["vip"]]).half().to(dev), kv_ip, M=R * L, N=2 * C, c1=Dc)
    ke, ve = ops.kv_pack2_elems(C, heads)
    kp = torch.empty(R * ke, dtype=torch.float16, device=dev)
    vp = torch.empty(R * ve, dtype=torch.float16, device=dev)
    ops.kv_pack2(kv_txt, kv_ip, kp, vp, R=R, L=L, C_=C, heads=heads, n_txt=n_txt, n_ip=n_ip, order="reg")
    wq_f, qs, qb = xattn_pack.fold_layernorm(mq.to(dev), ln[0].to(dev) if has_ln else None, ln[1].to(dev) if has_ln else None)
    wq_p, wo_p = xattn_pack.pack_w3(wq_f), xattn_pack.pack_w3(mo.half().to(dev).contiguous())
    out = torch.full((B, N, C), float("nan"), dtype=torch.float16, device=dev)
    xd = x.to(dev)
    x_before = xd.clone()
    ops.id_xattn3(xd, out, wq_p=wq_p, q_rowsum=qs, q_bias=qb, wo_p=wo_p, bo=W["bo"].half().to(dev),
                  kp=kp, vp=vp, kvrow=kvrow.to(dev), B=B, N=N, C_=C, heads=heads, n_txt=n_txt, n_ip=n_ip,
                  ip_scale=ip_scale, has_ln=has_ln, add_residual=residual)
    torch.cuda.synchronize()
    assert torch.equal(xd, x_before), "input was modified"
    check_vs_fp16_arm(out, ref, arm, f"id-xattn3 B={B} N={N} n_ip={n_ip} ln={has_ln} res={residual} shift={mean_shift}")
    # run-to-run determinism (DMA / barrier protocol): ten more launches, bit for bit
    for _ in range(10):
        out2 = torch.empty_like(out)
        ops.id_xattn3(xd, out2, wq_p=wq_p, q_rowsum=qs, q_bias=qb, wo_p=wo_p, bo=W["bo"].half().to(dev), kp=kp, vp=vp, kvrow=kvrow.to(dev), B=B, N=N, C_=C, heads=heads,
                      n_txt=n_txt, n_ip=n_ip, ip_scale=ip_scale, has_ln=has_ln, add_residual=residual)
        torch.cuda.synchronize()
        assert torch.equal(out, out2), "non-deterministic output"
