"""Size-independent properties of the hot kernels at BASELINE.json's full shapes (SD1.5 512x512, CFG batch 8: 32768 tokens x
320 channels at level 0), where the CPU oracle is too slow to be the checker: exact linearity / equivariance / batch
independence, and agreement between alternative launch paths of the same contraction."""
import pytest
import torch

from conftest import check_close

pytestmark = pytest.mark.gpu
B2, SIDE, C = 8, 64, 320


def _rnd(dev, *shape, seed=0, scale=0.5):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=dev) * scale).half()


def _conv(dev, x, w, bias=None, ws=None, side=SIDE, cin=C, cout=C):
    from consistentid_amd import ops
    M = x.shape[0]
    out = torch.empty(M, cout, dtype=torch.float16, device=dev)
    ops.gemm(x, w, out, M=M, N=cout, c1=cin, bias=bias, taps=9, Hi=side, Wi=side, Ho=side, Wo=side, ws=ws)
    return out


def test_conv3x3_full_size_is_linear_and_batch_local(dev):
    """level-0 3x3 conv (halo kernel, 60 GFLOP): scaling the input by 2 scales the output by 2 BIT-EXACTLY (powers of two
    commute with fp16 rounding and fp32 accumulation), and a sample's output does not depend on the other samples"""
    x, w = _rnd(dev, B2 * SIDE * SIDE, C, seed=1), _rnd(dev, C, 9 * C, seed=2, scale=0.02)
    y1 = _conv(dev, x, w)
    y2 = _conv(dev, x * 2, w)
    torch.cuda.synchronize()
    assert torch.isfinite(y1.float()).all() and y1.float().abs().max() > 0.1
    normal = y1.float().abs() >= 2.0 ** -14          # fp16 subnormal outputs round on a fixed grid: there 2 a need not be 2 round(a)
    assert torch.equal(y2[normal], (y1 * 2)[normal])
    assert (y2.float() - 2 * y1.float()).abs().max() <= 2.0 ** -24
    xm = x.clone()
    xm[SIDE * SIDE:] = _rnd(dev, (B2 - 1) * SIDE * SIDE, C, seed=3)          # change every sample but the first
    y3 = _conv(dev, xm, w)
    torch.cuda.synchronize()
    assert torch.equal(y3[:SIDE * SIDE], y1[:SIDE * SIDE])


def test_conv3x3_full_size_translation_equivariance(dev):
    """shifting the image one pixel right (zero column in) shifts the output one pixel right, bit-exactly away from the
    left / right borders: every tap of every interior output sees the same operands in the same order"""
    x, w = _rnd(dev, B2 * SIDE * SIDE, C, seed=4), _rnd(dev, C, 9 * C, seed=5, scale=0.02)
    img = x.view(B2, SIDE, SIDE, C)
    sh = torch.zeros_like(img)
    sh[:, :, 1:] = img[:, :, :-1]
    y = _conv(dev, x, w).view(B2, SIDE, SIDE, C)
    ys = _conv(dev, sh.reshape(-1, C).contiguous(), w).view(B2, SIDE, SIDE, C)
    torch.cuda.synchronize()
    assert torch.equal(ys[:, :, 2:-1], y[:, :, 1:-2])


def test_split_k_agrees_with_single_pass(dev):
    """8x8 level (M = 512, K = 11520): the split-K path (fp32 partials + reduce epilogue) against the same contraction
    without a workspace (single pass)"""
    side, c = 8, 1280
    x, w, b = _rnd(dev, B2 * side * side, c, seed=6), _rnd(dev, c, 9 * c, seed=7, scale=0.01), _rnd(dev, c, seed=8)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    y_split = _conv(dev, x, w, b, ws=ws, side=side, cin=c, cout=c)
    y_single = _conv(dev, x, w, b, ws=None, side=side, cin=c, cout=c)
    torch.cuda.synchronize()
    check_close(y_split, y_single.float(), "split-K vs single pass", tol_l2=5e-4, tol_max=2e-3)


def test_self_attention_full_size_properties(dev):
    """level-0 self-attention (N = 4096, 8 heads of 40): rows of softmax.V are convex combinations of V (bounded by V's
    range per channel), samples and heads are independent, and permuting the keys (with their values) changes nothing
    beyond fp32 summation order"""
    from consistentid_amd import ops
    N, heads, d = SIDE * SIDE, 8, 40
    B = 2
    x, w = _rnd(dev, B * N, C, seed=9), _rnd(dev, 3 * C, C, seed=10, scale=0.08)

    def run(xin):
        qk = torch.empty(B * N, 2 * C, dtype=torch.float16, device=dev)
        vt = torch.empty(B * heads * ops.dvp_of(d) * N, dtype=torch.float16, device=dev)
        ops.gemm(xin, w, qk, M=B * N, N=3 * C, c1=C, mode=2, vt=vt, n_vt0=2 * C, heads=heads, dhead=d, ntok=N)
        o = torch.empty(B * N, C, dtype=torch.float16, device=dev)
        ops.self_attn(qk, qk[:, C:], vt, o, B=B, N=N, heads=heads, d=d, ldq=2 * C, ldk=2 * C, ldo=C)
        return o
    o = run(x)
    v = (x.float() @ w[2 * C:].float().t()).view(B, N, C)
    torch.cuda.synchronize()
    ov = o.float().view(B, N, C)
    assert (ov <= v.max(dim=1, keepdim=True).values + 2e-2).all() and (ov >= v.min(dim=1, keepdim=True).values - 2e-2).all()
    xm = x.clone()
    xm[N:] = _rnd(dev, N, C, seed=11)
    assert torch.equal(run(xm)[:N], o[:N])                                   # sample 0 does not see sample 1
    perm = torch.randperm(N, generator=torch.Generator(device=dev).manual_seed(12), device=dev)
    xp = torch.cat([x[:N][perm], x[N:]])                                     # permute the tokens of sample 0
    op = run(xp)
    torch.cuda.synchronize()
    check_close(op[:N], o[:N][perm].float(), "key/query permutation equivariance", tol_l2=5e-4, tol_max=4e-3)


def test_fused_xattn_full_size_properties(dev):
    """level-0 fused ID cross-attention: the residual is added after everything else (exact additivity in fp32, one
    rounding), the ID stream vanishes with ip_scale = 0 exactly as with no ID tokens, samples are independent"""
    from consistentid_amd import ops
    N, heads = SIDE * SIDE, 8
    x = _rnd(dev, B2, N, C, seed=13)
    wq, wo, bo = _rnd(dev, C, C, seed=14, scale=0.05), _rnd(dev, C, C, seed=15, scale=0.05), _rnd(dev, C, seed=16)
    lg, lb = _rnd(dev, C, seed=17) + 1, _rnd(dev, C, seed=18, scale=0.1)
    kv_txt, kv_ip = _rnd(dev, B2 * 81, 2 * C, seed=19), _rnd(dev, B2 * 81, 2 * C, seed=20)
    ke, ve = ops.kv_pack_elems(C, heads)
    kvrow = torch.arange(B2, dtype=torch.int32, device=dev)

    def run(n_txt, n_ip, ip_scale, res):
        kp, vp = torch.empty(B2 * ke, dtype=torch.float16, device=dev), torch.empty(B2 * ve, dtype=torch.float16, device=dev)
        L = n_txt + n_ip                                   # kv rows are [sample][L]: keep the first L context rows per sample
        kt = kv_txt.view(B2, 81, -1)[:, :L].reshape(B2 * L, -1).contiguous()
        ki = kv_ip.view(B2, 81, -1)[:, :L].reshape(B2 * L, -1).contiguous()
        ops.kv_pack(kt, ki, kp, vp, R=B2, C_=C, heads=heads, n_txt=n_txt, n_ip=n_ip)
        out = torch.empty_like(x)
        ops.id_xattn(x, out, wq=wq, wo=wo, bo=bo, kp=kp, vp=vp, kvrow=kvrow, B=B2, N=N, C_=C, heads=heads, n_txt=n_txt,
                     n_ip=n_ip, ip_scale=ip_scale, residual=res, ln_gamma=lg, ln_beta=lb)
        return out
    full = run(77, 4, 1.0, x)
    nores = run(77, 4, 1.0, None)
    torch.cuda.synchronize()
    check_close(full, nores.float() + x.float(), "residual additivity", tol_l2=4e-4, tol_max=2e-3)
    off = run(77, 4, 0.0, x)
    none = run(77, 0, 1.0, x)
    torch.cuda.synchronize()
    # (77 + 4 tokens runs the compile-time-specialised softmax, 77 + 0 the generic one: same math, other summation order)
    check_close(off, none.float(), "ip_scale = 0 == no ID tokens", tol_l2=3e-4, tol_max=2e-3)
    assert (off.float() - full.float()).abs().max() > 1e-2          # while the ID stream does contribute at scale 1


def test_fused_xattn3_full_size_properties(dev):
    """the shipped level-0 kernel (third generation) at BASELINE's CFG batch 8 x 4096 tokens, no oracle needed:
    * the residual is one fp16 add on the fp16-rounded attn2 output (the reference's arithmetic): BIT-exact;
    * with ip_scale = 0 the ID stream is gone: the output does not depend on the ID keys / values, BIT-exactly;
    * samples are independent, and tokens too: permuting the tokens of a sample permutes its output rows, BIT-exactly
      (a token is one MFMA column; its LayerNorm statistics and softmax never see its neighbours)."""
    from consistentid_amd import ops, xattn_pack
    N, heads, L = SIDE * SIDE, 8, 81
    x = _rnd(dev, B2, N, C, seed=23)
    wq, wo, bo = _rnd(dev, C, C, seed=24, scale=0.05), _rnd(dev, C, C, seed=25, scale=0.05), _rnd(dev, C, seed=26)
    lg, lb = _rnd(dev, C, seed=27) + 1, _rnd(dev, C, seed=28, scale=0.1)
    kv_txt, kv_ip = _rnd(dev, B2 * L, 2 * C, seed=29), _rnd(dev, B2 * L, 2 * C, seed=30)
    kvrow = torch.arange(B2, dtype=torch.int32, device=dev)
    wq_f, qs, qb = xattn_pack.fold_layernorm(wq.float(), lg, lb)
    wq_p, wo_p = xattn_pack.pack_w3(wq_f), xattn_pack.pack_w3(wo)
    ke, ve = ops.kv_pack2_elems(C, heads)

    def pack(kv_ip_, order):
        kp, vp = torch.empty(B2 * ke, dtype=torch.float16, device=dev), torch.empty(B2 * ve, dtype=torch.float16, device=dev)
        ops.kv_pack2(kv_txt, kv_ip_, kp, vp, R=B2, L=L, C_=C, heads=heads, n_txt=77, n_ip=4, order=order)
        return kp, vp

    def run3(xin, kpvp, ip_scale=1.0, res=True):
        out = torch.empty_like(xin)
        ops.id_xattn3(xin, out, wq_p=wq_p, q_rowsum=qs, q_bias=qb, wo_p=wo_p, bo=bo, kp=kpvp[0], vp=kpvp[1], kvrow=kvrow, B=B2,
                      N=N, C_=C, heads=heads, n_txt=77, n_ip=4, ip_scale=ip_scale, has_ln=True, add_residual=res)
        return out
    kv3 = pack(kv_ip, "reg")
    full, nores = run3(x, kv3), run3(x, kv3, res=False)
    torch.cuda.synchronize()
    assert torch.isfinite(full.float()).all() and (full.float() - x.float()).abs().max() > 1e-2
    assert torch.equal(full, nores + x), "residual is not one fp16 add on the rounded block output"
    off_a, off_b = run3(x, kv3, ip_scale=0.0), run3(x, pack(_rnd(dev, B2 * L, 2 * C, seed=31, scale=2.0), "reg"), ip_scale=0.0)
    torch.cuda.synchronize()
    assert torch.equal(off_a, off_b), "ip_scale = 0 still lets the ID keys / values through"
    assert (off_a.float() - full.float()).abs().max() > 1e-2          # while the ID stream does contribute at scale 1
    xm = x.clone()
    xm[1:] = _rnd(dev, B2 - 1, N, C, seed=32)                         # change every sample but the first
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).to(dev)
    xm[0] = x[0][perm]                                                # ... and shuffle the first one's tokens
    moved = run3(xm, kv3)
    torch.cuda.synchronize()
    assert torch.equal(moved[0], full[0][perm]), "a token's output depends on its position or on other samples"


# ----------------------------------------------------------------------------- run-to-run determinism
def _repeat_equal(fn, reps=5):
    outs = [fn() for _ in range(reps)]
    torch.cuda.synchronize()
    return all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("N,c,heads", [(4096, 320, 8), (1024, 640, 8), (256, 1280, 8), (64, 1280, 8), (1024, 640, 10),
                                       (1024, 1280, 20), (256, 64, 2)])
def test_self_attention_is_deterministic(dev, N, c, heads):
    """The d = 40 kernel once read MFMA results through an inline-asm v_max3 chain for which hipcc inserts no MFMA -> VALU
    wait states: correct to 1 ulp, but different from run to run.  Every attention configuration of the engine is rerun
    on identical inputs and compared bit for bit (tools/determinism_probe.py as a test)."""
    from consistentid_amd import ops
    B, d = 2, c // heads
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.randn(B * N, c, generator=g, device=dev) * 0.5).half()
    w = (torch.randn(3 * c, c, generator=g, device=dev) * 0.08).half()
    qk = torch.empty(B * N, 2 * c, dtype=torch.float16, device=dev)
    vt = torch.empty(B * heads * ops.dvp_of(d) * N, dtype=torch.float16, device=dev)
    ops.gemm(x, w, qk, M=B * N, N=3 * c, c1=c, mode=2, vt=vt, n_vt0=2 * c, heads=heads, dhead=d, ntok=N)

    def run():
        o = torch.empty(B * N, c, dtype=torch.float16, device=dev)
        ops.self_attn(qk, qk[:, c:], vt, o, B=B, N=N, heads=heads, d=d, ldq=2 * c, ldk=2 * c, ldo=c)
        return o
    assert _repeat_equal(run, 6)


def test_conv_and_splitk_are_deterministic(dev):
    from consistentid_amd import ops
    g = torch.Generator(device=dev).manual_seed(2)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for B, side, cin, cout in ((8, 64, 320, 320), (8, 8, 1280, 1280)):       # halo kernel; split-K + reduce
        M = B * side * side
        x = (torch.randn(M, cin, generator=g, device=dev) * 0.5).half()
        w = (torch.randn(cout, 9 * cin, generator=g, device=dev) * 0.02).half()
        b = (torch.randn(cout, generator=g, device=dev) * 0.1).half()

        def run():
            o = torch.empty(M, cout, dtype=torch.float16, device=dev)
            ops.gemm(x, w, o, M=M, N=cout, c1=cin, bias=b, taps=9, Hi=side, Wi=side, Ho=side, Wo=side, ws=ws)
            return o
        assert _repeat_equal(run), (side, cin)


def test_engine_cross_attention_paths_agree_at_640_channels(dev):
    """HipUNet.cross_attention picks its launch sequence per level by measurement (unet._fused_gen1): at 640 channels one
    launch of the first-generation fused kernel when >= 16 k tokens are in flight (SDXL's 64 x 64 level); otherwise the query
    projection with the attention as its epilogue (cid_gemm_f16 mode 3) + out GEMM, or -- where that does not apply --
    LayerNorm + GEMM + core + GEMM.  All three sequences are the same arithmetic: on the shape where the rule switches they
    must agree to fp16 rounding (the last two bit for bit), and the rule must actually switch."""
    from consistentid_amd import synth, unet_spec
    from consistentid_amd.unet import HipUNet
    cfg = unet_spec.UNetConfig(sample_size=64, block_out_channels=(640, 640), layers_per_block=1,
                               down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                               up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), transformer_layers_per_block=(1, 1),
                               num_attention_heads=(10, 10), cross_attention_dim=256)
    sd = synth.random_unet_state_dict(cfg, seed=0, device=dev)
    ad = synth.random_adapter_state_dict(cfg, sd, rank=8, seed=1, device=dev)
    hip = HipUNet(cfg, sd, ad, device=dev)
    hip.set_context(_rnd(dev, 3, 81, 256, seed=4))
    layer = hip.packed.xattn_layers[0]
    B, N, c = 4, 4096, 640
    x = _rnd(dev, B * N, c, seed=5, scale=0.8)
    kvrow = torch.tensor([0, 1, 2, 1], dtype=torch.int32, device=dev)
    assert hip._fused_gen1(c, B * N) and not hip._fused_gen1(c, B * N // 2) and not hip._fused_gen1(1280, 1 << 20)
    assert "one launch" in hip.cross_attention_path(layer, c, B, N)
    fused = hip.cross_attention(layer, x, B, N, c, 10, kvrow).clone()
    rule, hip._fused_gen1 = hip._fused_gen1, (lambda c_, tokens: False)
    try:
        assert "attention epilogue" in hip.cross_attention_path(layer, c, B, N)
        epi = hip.cross_attention(layer, x, B, N, c, 10, kvrow).clone()
        hip._qattn = False
        assert "four launches" in hip.cross_attention_path(layer, c, B, N)
        split = hip.cross_attention(layer, x, B, N, c, 10, kvrow).clone()
    finally:
        hip._fused_gen1 = rule
        hip._qattn = True
    torch.cuda.synchronize()
    assert torch.isfinite(fused.float()).all() and (fused.float() - x.float()).abs().max() > 1e-2     # the block did something
    check_close(fused, split, "640-channel cross-attention: fused vs split launch sequence", tol_l2=2e-3, tol_max=2e-2)
    assert torch.equal(epi, split), "query projection with the attention epilogue differs from LayerNorm + GEMM + core + GEMM"
