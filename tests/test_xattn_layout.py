"""CPU check of the operand layouts of the fused identity cross-attention (consistentid_amd/csrc/xattn3.hip; the register-level
data flow it shares with its deleted predecessor, generation 2): a lane-level numpy emulation of the kernel's data flow -- MFMA fragment
semantics, accumulator-layout -> B-operand reuse, the K / V gather tables of consistentid_amd/xattn_pack.py, the
two-heads-per-wave channel split, the 0/1 "denominator" operand and the LayerNorm fold -- against the plain
formula of /root/reference/attention.py:236-282.  Runs in float64, so agreement is to rounding: any index slip
in the tables or in the kernel's addressing rules shows up as an O(1) error."""
import numpy as np
import pytest
import torch

from consistentid_amd import xattn_pack as xp

LANE = np.arange(64)
L16, LQ = LANE & 15, LANE >> 4


def mfma16(a, b, c):
    """v_mfma_f32_16x16x32_f16: a[l, j] = A[l16][8 lq + j], b[l, j] = B[8 lq + j][l16], c[l, i] = C[4 lq + i][l16]"""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for j in range(8):
        A[L16, 8 * LQ + j] = a[:, j]
        B[8 * LQ + j, L16] = b[:, j]
    Cm = A @ B
    out = c.copy()
    for i in range(4):
        out[:, i] += Cm[4 * LQ + i, L16]
    return out


def emulate(x, wq_f, qs, qb, wo, bo, kp, vp, n_txt, n_ip, ip_scale, eps, has_ln, add_res, order="slot"):
    """x [128, 320] one token tile; kp / vp the packed context row; returns out [128, 320]"""
    C = 320
    T = x.copy()                       # LDS token tile: x, later O
    out = np.zeros_like(x)
    n_all = n_txt + n_ip
    q_regs, resid = {}, {}
    # ---- phase A, per wave
    for wm in range(2):
        for wn in range(4):
            acc = {(ct, tt): np.zeros((64, 4)) for ct in range(5) for tt in range(4)}
            ssum, ssq = np.zeros((4, 64)), np.zeros((4, 64))
            for kstep in range(10):
                for tt in range(4):
                    tok = wm * 64 + tt * 16 + L16
                    b = np.stack([x[tok, 32 * kstep + 8 * LQ + j] for j in range(8)], 1)
                    ssum[tt] += b.sum(1)
                    ssq[tt] += (b * b).sum(1)
                    for ct in range(5):
                        row = wn * 80 + ct * 16 + L16
                        a = np.stack([wq_f[row, 32 * kstep + 8 * LQ + j] for j in range(8)], 1)
                        acc[ct, tt] = mfma16(a, b, acc[ct, tt])
            for tt in range(4):
                tot = np.zeros(64)
                tot2 = np.zeros(64)
                for q in range(4):          # rows_sum: over the four lane rows of a token
                    tot += ssum[tt][(L16 + 16 * q)]
                    tot2 += ssq[tt][(L16 + 16 * q)]
                mean = tot / C if has_ln else np.zeros(64)
                rstd = 1 / np.sqrt(np.maximum(tot2 / C - mean * mean, 0) + eps) if has_ln else np.ones(64)
                for ct in range(5):
                    q = np.zeros((64, 4))
                    for i in range(4):
                        ch = wn * 80 + ct * 16 + 4 * LQ + i
                        q[:, i] = rstd * (acc[ct, tt][:, i] - mean * qs[ch]) + qb[ch]
                    q_regs[wm, wn, ct, tt] = q
    # ---- phase B, per wave
    for wm in range(2):
        for wn in range(4):
            for ct in range(5):
                for tt in range(4):
                    r = np.zeros((64, 4))
                    for i in range(4):
                        r[:, i] = T[wm * 64 + tt * 16 + L16, wn * 80 + ct * 16 + 4 * LQ + i]
                    resid[wm, wn, ct, tt] = r
            ones_a = np.zeros((3, 64, 8))
            for ks in range(3):
                for j in range(8):
                    key = xp.slot_key(2 * ks + (j >> 2), LQ, j & 3, n_txt, n_ip, order)
                    ones_a[ks][:, j] = (((L16 & 3) == 0) & (key >= 0) & (key < n_txt)) | (((L16 & 3) == 1) & (key >= n_txt))
            for hh in range(2):
                h = 2 * wn + hh
                kf = kp.reshape(8, 6, 2, 64, 8)[h]
                vf = vp.reshape(8, 3, 3, 64, 8)[h]
                for tt in range(4):
                    qh = [q_regs[wm, wn, ct, tt] for ct in range(5)]
                    qb0 = np.concatenate([qh[0], qh[1]], 1) if hh == 0 else np.concatenate([qh[3], qh[4]], 1)
                    qb1 = np.concatenate([qh[2], qh[2]], 1)
                    s = []
                    for kt in range(6):
                        v = mfma16(kf[kt, 0], qb0, np.zeros((64, 4)))
                        s.append(mfma16(kf[kt, 1], qb1, v))
                    s = np.stack(s)                                   # [kt, lane, i]
                    key = xp.slot_key(np.arange(6)[:, None, None], LQ[None, :, None], np.arange(4)[None, None, :], n_txt, n_ip, order)
                    is_t, is_i = (key >= 0) & (key < n_txt), key >= n_txt
                    mt = np.where(is_t, s, -np.inf).max((0, 2))
                    mi = np.where(is_i, s, -np.inf).max((0, 2))
                    mt = np.max([mt[L16 + 16 * q] for q in range(4)], 0)   # rows_max
                    mi = np.max([mi[L16 + 16 * q] for q in range(4)], 0)
                    with np.errstate(invalid="ignore"):
                        p = np.where(is_t, np.exp2(s - mt[None, :, None]), np.where(is_i, np.exp2(s - mi[None, :, None]), 0.0))
                    p = np.nan_to_num(p)
                    pk = [np.concatenate([p[2 * ks], p[2 * ks + 1]], 1) for ks in range(3)]
                    l = np.zeros((64, 4))
                    for ks in range(3):
                        l = mfma16(ones_a[ks], pk[ks], l)
                    rho = np.where(l[:, 1] > 0, ip_scale * l[:, 0] / np.where(l[:, 1] > 0, l[:, 1], 1), 0.0)
                    p2 = np.where(is_i, p * rho[None, :, None], p)
                    pk = [np.concatenate([p2[2 * ks], p2[2 * ks + 1]], 1) for ks in range(3)]
                    for dt in range(3):
                        o = np.zeros((64, 4))
                        for ks in range(3):
                            o = mfma16(vf[dt, ks], pk[ks], o)
                        o = o / l[:, 0:1]
                        ct = dt + 2 * hh
                        for i in range(4):
                            wr = np.ones(64, bool) if ct != 2 else ((LQ < 2) if hh == 0 else (LQ >= 2))
                            rows = wm * 64 + tt * 16 + L16
                            cols = wn * 80 + ct * 16 + 4 * LQ + i
                            T[rows[wr], cols[wr]] = o[wr, i]
    # ---- phase C
    for wm in range(2):
        for wn in range(4):
            for ct in range(5):
                for tt in range(4):
                    acc = np.zeros((64, 4))
                    tok = wm * 64 + tt * 16 + L16
                    for kstep in range(10):
                        b = np.stack([T[tok, 32 * kstep + 8 * LQ + j] for j in range(8)], 1)
                        row = wn * 80 + ct * 16 + L16
                        a = np.stack([wo[row, 32 * kstep + 8 * LQ + j] for j in range(8)], 1)
                        acc = mfma16(a, b, acc)
                    for i in range(4):
                        ch = wn * 80 + ct * 16 + 4 * LQ + i
                        v = acc[:, i] + bo[ch]
                        if add_res:
                            v = v + resid[wm, wn, ct, tt][:, i]
                        out[tok, ch] = v
    return out


def reference(x, wq, gamma, beta, wo, bo, k_t, v_t, k_i, v_i, n_txt, n_ip, ip_scale, eps, has_ln, add_res, heads=8):
    """attention.py:236-282 on one sample: q pre-scaled by d^-0.5 log2(e) -> base-2 softmaxes"""
    xn = x
    if has_ln:
        mu = x.mean(1, keepdims=True)
        var = x.var(1, keepdims=True)
        xn = (x - mu) / np.sqrt(var + eps) * gamma + beta
    q = xn @ wq.T
    N, C = x.shape
    d = C // heads
    o = np.zeros_like(x)
    for h in range(heads):
        sl = slice(h * d, (h + 1) * d)
        st = q[:, sl] @ k_t[:n_txt, sl].T
        pt = np.exp2(st - st.max(1, keepdims=True))
        o[:, sl] = (pt / pt.sum(1, keepdims=True)) @ v_t[:n_txt, sl]
        if n_ip:
            si = q[:, sl] @ k_i[n_txt:, sl].T
            pi = np.exp2(si - si.max(1, keepdims=True))
            o[:, sl] += ip_scale * (pi / pi.sum(1, keepdims=True)) @ v_i[n_txt:, sl]
    out = o @ wo.T + bo
    return out + x if add_res else out


@pytest.mark.parametrize("n_txt,n_ip,has_ln,add_res,order", [
    # "slot": the key order of the deleted second generation (the tables still build it: the data flow is order-independent)
    (77, 4, True, True, "slot"), (77, 4, False, False, "slot"), (81, 0, True, True, "slot"), (60, 7, True, False, "slot"),
    # register-major key order of the shipped third generation (same register-level data flow, other K / V gather tables)
    (77, 4, True, True, "reg"), (81, 0, True, False, "reg"), (60, 7, False, True, "reg")])
def test_fused_xattn_dataflow_matches_reference(n_txt, n_ip, has_ln, add_res, order):
    rng = np.random.default_rng(5)
    C, heads, L = 320, 8, n_txt + n_ip
    x = rng.standard_normal((128, C)) * 1.3 + 0.4
    wq = rng.standard_normal((C, C)) * C ** -0.5
    wo = rng.standard_normal((C, C)) * C ** -0.5
    bo = rng.standard_normal(C) * 0.1
    gamma, beta = 1 + 0.1 * rng.standard_normal(C), 0.1 * rng.standard_normal(C)
    kv_txt = rng.standard_normal((L, 2 * C))       # [K | V] rows from the text projection (every context row)
    kv_ip = rng.standard_normal((L, 2 * C))        # ... and from the ID projection
    if has_ln:
        wq_f = wq * gamma[None, :]
        qs, qb = wq_f.sum(1), wq @ beta
    else:
        wq_f, qs, qb = wq, np.zeros(C), np.zeros(C)
    k_idx, v_idx = xp.kv_index_tables(C, heads, n_txt, n_ip, order)

    def gather(idx):
        flat_a, flat_b = kv_txt.reshape(-1), kv_ip.reshape(-1)
        off = idx & (xp.IP_FLAG - 1)
        val = np.where(idx & xp.IP_FLAG, flat_b[np.minimum(off, flat_b.size - 1)], flat_a[np.minimum(off, flat_a.size - 1)])
        return np.where(idx < 0, 0.0, val)

    kp, vp = gather(k_idx.astype(np.int64)), gather(v_idx.astype(np.int64))
    got = emulate(x, wq_f, qs, qb, wo, bo, kp, vp, n_txt, n_ip, 0.8, 1e-5, has_ln, add_res, order)
    ref = reference(x, wq, gamma, beta, wo, bo, kv_txt[:, :C], kv_txt[:, C:], kv_ip[:, :C], kv_ip[:, C:], n_txt, n_ip,
                    0.8, 1e-5, has_ln, add_res)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-9, err


def test_fold_layernorm_identity():
    g = torch.Generator().manual_seed(0)
    C = 320
    w = torch.randn(C, C, generator=g, dtype=torch.float64) * C ** -0.5
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g, dtype=torch.float64), 0.1 * torch.randn(C, generator=g, dtype=torch.float64)
    x = torch.randn(7, C, generator=g, dtype=torch.float64) * 2 + 3          # a large mean: the cancellation case
    wf, s, b = xp.fold_layernorm(w.float(), gamma.float(), beta.float())
    mu, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    got = rstd * (x @ wf.double().T - mu * s.double()) + b.double()
    # the fold is exact for the ROUNDED weight: compare with LN applied in front of that same weight
    ref = ((x - mu) * rstd) @ wf.double().T + beta @ w.T
    assert (got - ref).abs().max() < 1e-5
    # and within fp16 weight rounding of the unfolded formula
    ref2 = ((x - mu) * rstd * gamma + beta) @ w.T
    assert (got - ref2).abs().max() / ref2.abs().max() < 2e-3


@pytest.mark.parametrize("order", ["slot", "reg"])
def test_kv_tables_cover_every_value_once(order):
    """every real (key, channel) of K and of V appears in exactly one fragment slot"""
    C, heads, n_txt, n_ip = 320, 8, 77, 4
    L = n_txt + n_ip
    k_idx, v_idx = xp.kv_index_tables(C, heads, n_txt, n_ip, order)
    for idx, base in ((k_idx, 0), (v_idx, C)):
        ok = idx[idx >= 0].astype(np.int64)
        off = ok & (xp.IP_FLAG - 1)
        key, col = off // (2 * C), off % (2 * C)
        assert ((ok & xp.IP_FLAG) != 0).tolist() == (key >= n_txt).tolist()
        assert col.min() >= base and col.max() < base + C and key.max() == L - 1
        pairs = key * C + (col - base)
        assert len(np.unique(pairs)) == len(pairs) == L * C


def test_pack_w3_is_the_a_operand_stream():
    """xattn_pack.pack_w3: 1-KiB block (wave, k-step, row tile) holds lane l's 8 halfs of the A operand
    a[l][j] = W[80 wave + 16 tile + (l & 15)][32 step + 8 (l >> 4) + j] (csrc/xattn3.hip load_w)"""
    w = torch.arange(320 * 320, dtype=torch.float32).reshape(320, 320)
    p = xp.pack_w3(w).reshape(4, 10, 5, 64, 8).numpy()
    wn = w.numpy()
    rng = np.random.default_rng(0)
    for _ in range(200):
        wave, step, tile, lane, j = (int(rng.integers(n)) for n in (4, 10, 5, 64, 8))
        assert p[wave, step, tile, lane, j] == wn[80 * wave + 16 * tile + (lane & 15), 32 * step + 8 * (lane >> 4) + j]
    assert sorted(p.reshape(-1).tolist()) == list(range(320 * 320))          # a permutation
    with pytest.raises(ValueError):
        xp.pack_w3(torch.zeros(640, 640))
