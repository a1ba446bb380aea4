"""Host-side operand preparation for the fused identity cross-attention
(csrc/xattn3.hip, ``cid_id_xattn3_f16``; the "slot" key order is that of its deleted predecessor and is kept for the CPU layout emulation in tests/test_xattn_layout.py): index tables that put the projected K / V of one context row
into MFMA-fragment order, and the LayerNorm fold of the query projection.

Fragment conventions of ``v_mfma_f32_16x16x32_f16`` (lane l, l16 = l & 15, lq = l >> 4):
  A operand  a[j] = A[row l16][k = 8 lq + j]        B operand  b[j] = B[k = 8 lq + j][col l16]
  C / D      c[i] = C[row 4 lq + i][col l16]
The kernel keeps activations in the C/D layout (lane = token, 4 consecutive channels per register quad) and
feeds them back as B operands, two register quads per k-step.  That fixes the contraction order
  k-slot (lq, j)  <->  index 16 * (j >> 2) + 4 * lq + (j & 3)   of the 32 values of a k-step,
which the A operands (K rows, V^T rows) must follow -- that is all these tables encode.

A wave owns 80 channels = two heads of 40: head A = channel tiles 0, 1 and rows 0..7 of tile 2, head B = rows
8..15 of tile 2 and tiles 3, 4.  k-slots / rows that belong to the other head (or to no channel, no key) are zero.

Everything here is index arithmetic on the host (numpy) and a few torch ops at weight-load time; the packing
itself runs on the GPU (``cid_gather_pack_f16``).  Reference semantics: /root/reference/attention.py:236-279.
"""
from __future__ import annotations

from functools import lru_cache
from typing import Tuple

import numpy as np
import torch

HEAD_DIM = 40
KT, QK_STEPS = 6, 2        # 96 key slots as six 16-key tiles; head dim 40 as two 32-deep k-steps
DT, PV_STEPS = 3, 3        # V^T rows as three 16-row tiles; 96 keys as three 32-deep k-steps
K_FRAGS, V_FRAGS = KT * QK_STEPS, DT * PV_STEPS
IP_FLAG = 1 << 30


def slot_index(lq: np.ndarray, j: np.ndarray) -> np.ndarray:
    """position (0..31) inside a k-step of the value held in k-slot (lq, j) of a B operand built from two
    accumulator register quads"""
    return 16 * (j >> 2) + 4 * lq + (j & 3)


def slot_key(kt, lq, i, n_txt: int, n_ip: int, order: str = "slot"):
    """context key (or -1) held by score register (kt, i) in lane row lq -- equivalently row 4 lq + i of key tile kt,
    k-slot 16 kt + 4 lq + i of the P.V contraction.
      order "slot" (second generation): key = 16 kt + 4 lq + i, text then ID keys in slot order;
      order "reg"  (third generation): register-major -- register rho = 4 kt + i holds keys 4 rho .. 4 rho + 3 in its four
        lane rows, text keys first, the ID keys start at the next WHOLE register.  Every register is then all-text,
        all-ID or absent in every lane row except at most one partly filled register per stream (77 + 4: registers
        0..18 text, 19 = key 76 in lane row 0, 20 = the four ID keys), which is what keeps predicates out of the
        softmax arithmetic."""
    kt, lq, i = np.asarray(kt), np.asarray(lq), np.asarray(i)
    L = n_txt + n_ip
    if order == "slot":
        key = 16 * kt + 4 * lq + i
        return np.where(key < L, key, -1)
    assert order == "reg", order
    rho = 4 * kt + i
    ip0 = (n_txt + 3) // 4
    tkey = 4 * rho + lq
    ikey = n_txt + 4 * (rho - ip0) + lq
    return np.where(tkey < n_txt, tkey, np.where((rho >= ip0) & (ikey < L), ikey, -1))


def k_channel(parity: int, ks: int, lq: np.ndarray, j: np.ndarray) -> np.ndarray:
    """head-dim index d (or -1) that k-slot (lq, j) of Q.K^T k-step ``ks`` carries for the even (0) / odd (1) head of
    a wave's head pair"""
    if ks == 0:
        d = slot_index(lq, j) + (8 if parity else 0)                 # tiles (0, 1) resp. (3, 4)
        return d
    # k-step 1 = channel tile 2 (first register quad only): rows 0..7 are head A's d = 32..39, rows 8..15 head B's d = 0..7
    row = 4 * lq + (j & 3)
    if parity == 0:
        return np.where((j < 4) & (row < 8), 32 + row, -1)
    return np.where((j < 4) & (row >= 8), row - 8, -1)


def v_row_channel(parity: int, dt: int, r: np.ndarray) -> np.ndarray:
    """head-dim index d (or -1) of row r of V^T tile ``dt`` (tile dt of head A is channel tile dt of the wave, tile dt of
    head B is channel tile dt + 2)"""
    if parity == 0:
        return np.where(r < 8, 32 + r, -1) if dt == 2 else 16 * dt + r
    return np.where(r >= 8, r - 8, -1) if dt == 0 else 8 + 16 * (dt - 1) + r


@lru_cache(maxsize=16)
def kv_index_tables(C: int, heads: int, n_txt: int, n_ip: int, order: str = "slot") -> Tuple[np.ndarray, np.ndarray]:
    """(k_idx, v_idx): int32 gather tables for ONE context row.  Source = the [L, 2C] block of projected [K | V]
    rows (text projection for keys < n_txt, bit 30 set = take the ID projection); -1 = zero.  ``order``: see slot_key."""
    D = C // heads
    assert D == HEAD_DIM and C == heads * D, "the fused level-0 kernel is built for 40-wide heads"
    L = n_txt + n_ip
    assert 0 < n_txt and 0 <= n_ip and L <= 16 * KT
    assert order == "slot" or 4 * ((n_txt + 3) // 4) + n_ip <= 16 * KT
    lane = np.arange(64)
    l16, lq = (lane & 15)[:, None], (lane >> 4)[:, None]
    j = np.arange(8)[None, :]
    k_idx = np.full((heads, KT, QK_STEPS, 64, 8), -1, dtype=np.int64)
    v_idx = np.full((heads, DT, PV_STEPS, 64, 8), -1, dtype=np.int64)
    for h in range(heads):
        par = h & 1
        for kt in range(KT):
            key = np.broadcast_to(slot_key(kt, l16 >> 2, l16 & 3, n_txt, n_ip, order), (64, 8))      # tile row l16 = 4 lq' + i
            for ks in range(QK_STEPS):
                d = np.broadcast_to(k_channel(par, ks, lq, j), (64, 8))
                ok = (key >= 0) & (d >= 0)
                off = key * (2 * C) + h * D + d + np.where(key >= n_txt, IP_FLAG, 0)
                k_idx[h, kt, ks] = np.where(ok, off, -1)
        for dt in range(DT):
            d = np.broadcast_to(v_row_channel(par, dt, l16), (64, 8))
            for ks in range(PV_STEPS):
                key = np.broadcast_to(slot_key(2 * ks + (j >> 2), lq, j & 3, n_txt, n_ip, order), (64, 8))
                ok = (key >= 0) & (d >= 0)
                off = key * (2 * C) + C + h * D + d + np.where(key >= n_txt, IP_FLAG, 0)
                v_idx[h, dt, ks] = np.where(ok, off, -1)
    return k_idx.reshape(-1).astype(np.int32), v_idx.reshape(-1).astype(np.int32)


def fold_layernorm(wq_scaled: torch.Tensor, gamma: torch.Tensor | None, beta: torch.Tensor | None):
    """LayerNorm folded into the query projection:  LN(x) Wq^T = rstd * (x Wq'^T - mean * s) + b'  with
    Wq' = Wq diag(gamma) (rounded to fp16), s = row sums of the ROUNDED Wq' (so the mean term cancels exactly),
    b' = Wq beta.  ``wq_scaled``: fp32 [C, C] merged to_q weight already multiplied by d^-0.5 * log2(e)."""
    w = wq_scaled.float()
    if gamma is None:
        wf = w.half().contiguous()
        z = torch.zeros(w.shape[0], dtype=torch.float32, device=w.device)
        return wf, z, z.clone()
    wf = (w * gamma.float()[None, :]).half().contiguous()
    return wf, wf.float().sum(1).contiguous(), (w @ beta.float()).contiguous()


def pack_w3(w: torch.Tensor) -> torch.Tensor:
    """[320, 320] projection matrix (rows = output channels) -> the A-operand stream of the third-generation fused
    kernel (csrc/xattn3.hip): [wave 4][k-step 10][row tile 5][lane 64][8] with
    element (wave, step, tile, lane, j) = w[80 wave + 16 tile + (lane & 15)][32 step + 8 (lane >> 4) + j] --
    one 1-KiB global load per MFMA A operand, a wave's whole slice (50 KiB) contiguous."""
    if tuple(w.shape) != (320, 320):
        raise ValueError(f"pack_w3: [320, 320] expected, got {tuple(w.shape)}")
    # (wave, tile, l16, step, lq, j) -> (wave, step, tile, lq, l16, j);  lane = 16 lq + l16
    return w.reshape(4, 5, 16, 10, 4, 8).permute(0, 3, 1, 4, 2, 5).contiguous().reshape(320, 320)
