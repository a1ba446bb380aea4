"""Property tests (hypothesis) of the host-side logic: batch sharding, weight-layout permutations, scheduler tables,
topology inventories.  The reference has no tests at all; these pin the invariants the engine relies on."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from consistentid_amd import distributed, scheduler, unet_spec, weights


@settings(max_examples=200, deadline=None)
@given(n=st.integers(0, 5000), world=st.integers(1, 64))
def test_shard_range_partitions_the_batch(n, world):
    """images are the independent units: contiguous, disjoint, covering, balanced to within one image"""
    spans = [distributed.shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [hi - lo for lo, hi in spans]
    assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1


@settings(max_examples=50, deadline=None)
@given(blocks=st.integers(1, 12), cols=st.integers(1, 5))
def test_geglu_interleave_is_a_block_permutation(blocks, cols):
    """rows [value | gate] -> 16-row blocks v0 g0 v1 g1 ...: every row appears once, value block j lands at 32 j, its gate at 32 j + 16"""
    half = 16 * blocks
    t = torch.arange(2 * half * cols, dtype=torch.float32).reshape(2 * half, cols)
    out = weights._geglu_interleave(t)
    assert sorted(out[:, 0].tolist()) == sorted(t[:, 0].tolist())
    for j in range(blocks):
        assert torch.equal(out[32 * j:32 * j + 16], t[16 * j:16 * j + 16])
        assert torch.equal(out[32 * j + 16:32 * j + 32], t[half + 16 * j:half + 16 * j + 16])


@settings(max_examples=40, deadline=None)
@given(steps=st.integers(1, 200))
def test_scheduler_tables(steps):
    """DDIM: timesteps strictly decreasing, in range, last update lands on alphas_cumprod[0]; Euler: same timesteps, sigmas
    strictly decreasing to 0, the step increments telescope to -sigma_0, model-input scale = 1/sqrt(sigma^2+1) in (0, 1)"""
    d, e = scheduler.DDIMScheduler(), scheduler.EulerDiscreteScheduler()
    d.set_timesteps(steps)
    e.set_timesteps(steps)
    ts = d.timesteps
    assert len(ts) == steps and (np.diff(ts) < 0).all() if steps > 1 else True
    assert ts.min() >= 1 and ts.max() <= 1000
    assert np.array_equal(ts.astype(np.float32), e.timesteps)
    td, te = d.coefficient_table(inpaint=True), e.coefficient_table(inpaint=True)
    assert td.shape == te.shape == (steps, 5) and np.isfinite(td).all() and np.isfinite(te).all()
    assert (td[:, 4] == 1).all() and (td[:, 0] > 0).all()
    sig = e.sigmas
    assert sig[-1] == 0 and (np.diff(sig) < 0).all()
    assert abs(te[:, 1].sum() + sig[0]) < 1e-3 * max(1.0, sig[0])
    assert ((te[:, 4] > 0) & (te[:, 4] < 1)).all()
    assert abs(e.init_noise_sigma - (sig[0] ** 2 + 1) ** 0.5) < 1e-4


@settings(max_examples=30, deadline=None)
@given(levels=st.integers(2, 4), layers=st.integers(1, 3), attn=st.lists(st.booleans(), min_size=4, max_size=4))
def test_topology_inventories_agree(levels, layers, attn):
    """for random block layouts: one ControlNet zero conv per UNet skip tensor, processor names unique and ordered
    down -> up -> mid, every transformer layer has a self- and a cross-attention processor"""
    boc = tuple(64 * (i + 1) for i in range(levels))
    down = tuple("CrossAttnDownBlock2D" if attn[i] else "DownBlock2D" for i in range(levels))
    up = tuple("CrossAttnUpBlock2D" if attn[levels - 1 - i] else "UpBlock2D" for i in range(levels))
    cfg = unet_spec.UNetConfig(sample_size=32, block_out_channels=boc, down_block_types=down, up_block_types=up,
                               layers_per_block=layers, transformer_layers_per_block=(1,) * levels,
                               num_attention_heads=(2,) * levels, cross_attention_dim=64)
    downs, mid, ups = unet_spec.walk(cfg)
    n_skips = 1 + sum(len(b.resnets) + (1 if b.sampler else 0) for b in downs)
    assert len(unet_spec.controlnet_zero_conv_channels(cfg)) == n_skips
    assert sum(len(b.resnets) for b in ups) == n_skips            # every skip is consumed exactly once on the way up
    names = unet_spec.attn_processor_names(cfg)
    assert len(names) == len(set(names)) and len(names) % 2 == 0
    order = [n.split(".")[0] for n in names]
    assert order == sorted(order, key=lambda s: ("down_blocks", "up_blocks", "mid_block").index(s))
    shapes = unet_spec.unet_param_shapes(cfg)
    enc = unet_spec.unet_param_shapes(cfg, encoder_only=True)
    assert set(enc) < set(shapes) and all(shapes[k] == v for k, v in enc.items())
    assert not any(k.startswith(("up_blocks", "conv_out", "conv_norm_out")) for k in enc)
