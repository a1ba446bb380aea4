import numpy as np
import torch

from consistentid_amd import distributed, unet_spec, weights


def test_geglu_interleave_blocks():
    t = torch.arange(64).reshape(64, 1).float()
    out = weights._geglu_interleave(t).reshape(-1)
    assert out[:16].tolist() == list(range(0, 16))          # value block 0
    assert out[16:32].tolist() == list(range(32, 48))       # gate block 0
    assert out[32:48].tolist() == list(range(16, 32))       # value block 1
    assert out[48:].tolist() == list(range(48, 64))


def test_shard_range_partitions_contiguously():
    for total, world in ((64, 8), (16, 8), (7, 3), (3, 8), (0, 4)):
        got = [distributed.shard_range(total, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total
        for (a, b), (c, d) in zip(got, got[1:]):
            assert b == c and b >= a
        sizes = [b - a for a, b in got]
        assert max(sizes) - min(sizes) <= 1


def test_flatten_roundtrip_alignment():
    g = torch.Generator().manual_seed(0)
    ts = [torch.randn(3, 5, generator=g).half(), torch.randn(17, generator=g).half(), torch.randn(2, 2, 2, generator=g).half()]
    flat, meta = distributed.flatten(ts)
    assert all(off % 8 == 0 for _, off in meta)
    for a, b in zip(ts, distributed.unflatten(flat, meta)):
        assert torch.equal(a, b)


def test_walk_matches_known_sd15_topology():
    downs, mid, ups = unet_spec.walk(unet_spec.sd15_config())
    assert [r.cin for b in ups for r in b.resnets] == [2560, 2560, 2560, 2560, 2560, 1920, 1920, 1280, 960, 960, 640, 640]
    assert sum(len(b.resnets) for b in downs + [mid] + ups) == 22
    assert sum(t.n_layers for b in downs + [mid] + ups for t in b.attentions) == 16
    d2, m2, u2 = unet_spec.walk(unet_spec.sdxl_config())
    assert sum(t.n_layers for b in d2 + [m2] + u2 for t in b.attentions) == 70
    assert sum(len(b.resnets) for b in d2 + [m2] + u2) == 17


def test_hidden_size_rule():
    cfg = unet_spec.sd15_config()
    names = unet_spec.attn_processor_names(cfg)
    hs = [unet_spec.hidden_size_of(cfg, n) for n in names]
    assert hs[:12] == [320] * 4 + [640] * 4 + [1280] * 4
    assert hs[12:30] == [1280] * 6 + [640] * 6 + [320] * 6 and hs[30:] == [1280, 1280]


def test_euler_tables_match_oracle_scheduler():
    """product EulerDiscreteScheduler (coefficient table form) vs the oracle's step-by-step restatement"""
    import numpy as np
    import torch
    from consistentid_amd import scheduler
    from oracle import ddim
    p, o = scheduler.EulerDiscreteScheduler(), ddim.EulerDiscreteScheduler()
    p.set_timesteps(30)
    o.set_timesteps(30)
    assert np.array_equal(p.timesteps, o.timesteps.numpy()) and abs(p.init_noise_sigma - o.init_noise_sigma) < 1e-4   # fp32 table vs float()
    tab = p.coefficient_table(inpaint=True)
    x, eps, init, noise = torch.randn(4, 7, dtype=torch.float64, generator=torch.Generator().manual_seed(0)).unbind(0)
    for i, t in enumerate(o.timesteps):
        cx, ce, ci, cn, cin = [float(v) for v in tab[i]]
        # the table is fp32, the oracle steps in fp64: agreement to fp32 rounding of the coefficients
        assert torch.allclose(cx * x + ce * eps, o.step(eps, t, x), rtol=1e-5, atol=1e-5)
        assert torch.allclose(cin * x, o.scale_model_input(x, t), rtol=1e-5, atol=1e-5)
        if i < len(o.timesteps) - 1:
            assert torch.allclose(ci * init + cn * noise, o.add_noise(init, noise, o.timesteps[i + 1]), rtol=1e-5, atol=1e-5)
