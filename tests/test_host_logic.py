import numpy as np
import pytest
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


@pytest.mark.parametrize("spacing", ["leading", "linspace", "trailing"])
def test_euler_tables_match_oracle_scheduler(spacing):
    """product EulerDiscreteScheduler (coefficient table form) vs the oracle's step-by-step restatement"""
    import numpy as np
    import torch
    from consistentid_amd import scheduler
    from oracle import ddim
    p, o = scheduler.EulerDiscreteScheduler(timestep_spacing=spacing), ddim.EulerDiscreteScheduler(timestep_spacing=spacing)
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


@pytest.mark.parametrize("spacing", ["leading", "linspace", "trailing"])
def test_ddim_tables_match_oracle_scheduler(spacing):
    """product DDIMScheduler (coefficient table form) vs the oracle's step-by-step restatement, every timestep spacing"""
    import numpy as np
    import torch
    from consistentid_amd import scheduler
    from oracle import ddim
    p, o = scheduler.DDIMScheduler(timestep_spacing=spacing), ddim.DDIMScheduler(timestep_spacing=spacing)
    p.set_timesteps(50)
    o.set_timesteps(50)
    assert np.array_equal(p.timesteps, o.timesteps.numpy()) and len(set(p.timesteps.tolist())) == 50
    assert (np.diff(p.timesteps) < 0).all() and 0 <= p.timesteps.min() and p.timesteps.max() <= 999
    tab = p.coefficient_table(inpaint=True)
    x, eps, init, noise = torch.randn(4, 9, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).unbind(0)
    for i, t in enumerate(o.timesteps):
        cx, ce, ci, cn, cin = [float(v) for v in tab[i]]
        assert torch.allclose(cx * x + ce * eps, o.step(eps, t, x), rtol=1e-5, atol=1e-5) and cin == 1.0
        if i < len(o.timesteps) - 1:
            assert torch.allclose(ci * init + cn * noise, o.add_noise(init, noise, o.timesteps[i + 1]), rtol=1e-5, atol=1e-5)


def test_scheduler_from_config():
    """``Scheduler.from_config(pipe.scheduler.config)`` of the reference scripts (infer.py:33, demo/controlnet_demo.py:67): the
    Stable Diffusion v1 scheduler_config.json (PNDM, no timestep_spacing key -> that class's "leading") and SDXL's Euler
    config; unsupported options are refused, not ignored"""
    from consistentid_amd import scheduler
    sd15 = {"_class_name": "PNDMScheduler", "_diffusers_version": "0.6.0", "beta_end": 0.012, "beta_schedule": "scaled_linear",
            "beta_start": 0.00085, "num_train_timesteps": 1000, "set_alpha_to_one": False, "skip_prk_steps": True,
            "steps_offset": 1, "trained_betas": None, "clip_sample": False}
    sdxl = {"_class_name": "EulerDiscreteScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085,
            "clip_sample": False, "interpolation_type": "linear", "num_train_timesteps": 1000, "prediction_type": "epsilon",
            "sample_max_value": 1.0, "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1,
            "timestep_spacing": "leading", "trained_betas": None, "use_karras_sigmas": False}
    for cfg in (sd15, sdxl):
        e, d = scheduler.EulerDiscreteScheduler.from_config(cfg), scheduler.DDIMScheduler.from_config(cfg)
        assert (e.timestep_spacing, e.steps_offset, d.timestep_spacing, d.steps_offset) == ("leading", 1, "leading", 1)
        e.set_timesteps(50)
        ref = scheduler.EulerDiscreteScheduler()
        ref.set_timesteps(50)
        assert (e.coefficient_table() == ref.coefficient_table()).all()
    lin = scheduler.EulerDiscreteScheduler.from_config(dict(sdxl, timestep_spacing="linspace"))
    lin.set_timesteps(30)
    assert lin.timesteps[0] == 999.0 and lin.timesteps[-1] == 0.0 and abs(lin.init_noise_sigma - float(lin.sigmas.max())) < 1e-6
    for bad in (dict(sdxl, use_karras_sigmas=True), dict(sdxl, prediction_type="v_prediction"),
                dict(sdxl, beta_schedule="linear"), dict(sd15, clip_sample=True)):
        with pytest.raises(NotImplementedError):
            scheduler.EulerDiscreteScheduler.from_config(bad)
    with pytest.raises(ValueError):
        scheduler.DDIMScheduler(timestep_spacing="karras")


def test_pipeline_scheduler_assignment_reaches_the_engine(tmp_path):
    """infer.py:33 replaces the scheduler AFTER the pipeline is built -- ``pipe.scheduler = EulerDiscreteScheduler.from_config(
    pipe.scheduler.config)``: the assignment must change what the denoise engine steps with (it used to keep the
    constructor's DDIM), the config must carry the base model's values over, and a diffusers object is refused loudly"""
    import json
    from types import SimpleNamespace
    from consistentid_amd import loader, pipeline, scheduler
    (tmp_path / "scheduler").mkdir()
    (tmp_path / "scheduler" / "scheduler_config.json").write_text(json.dumps(
        {"_class_name": "PNDMScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085,
         "num_train_timesteps": 1000, "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1, "trained_betas": None,
         "clip_sample": False, "timestep_spacing": "trailing"}))
    base = loader.read_scheduler(tmp_path)
    assert isinstance(base, scheduler.DDIMScheduler) and base.timestep_spacing == "trailing"
    assert loader.read_scheduler(tmp_path / "scheduler") is None
    pipe = pipeline.ConsistentIDStableDiffusionPipeline(SimpleNamespace(device="cpu"), scheduler=base, use_graph=False)
    assert pipe.scheduler is base and pipe._engine.scheduler is base
    pipe.scheduler = scheduler.EulerDiscreteScheduler.from_config(pipe.scheduler.config)
    assert isinstance(pipe._engine.scheduler, scheduler.EulerDiscreteScheduler) and pipe._engine.scheduler is pipe.scheduler
    assert pipe.scheduler.timestep_spacing == "trailing" and pipe.scheduler.steps_offset == 1
    with pytest.raises(TypeError):
        pipe.scheduler = object()


def test_bench_gpu_state_sampler_parses_rocm_smi(monkeypatch):
    """bench.GpuStateSampler: one `rocm-smi --json` record -> sclk / power / junction temperature; no rocm-smi -> no samples,
    summary None (the bench line then carries gpu_state: null instead of failing)"""
    import subprocess
    import bench

    class R:
        stdout = ('WARNING: something on stderr-like first line\n{"card0": {"Temperature (Sensor junction) (C)": "51.0", '
                  '"Temperature (Sensor memory) (C)": "40.0", "sclk clock speed:": "(2157Mhz)", "mclk clock speed:": "(2000Mhz)", '
                  '"Current Socket Graphics Package Power (W)": "1093.0"}}')

    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R())
    s = bench.GpuStateSampler(0)
    assert s._read() == {"temp_c": 51.0, "sclk_mhz": 2157.0, "power_w": 1093.0}
    s.samples = [s._read(), {"sclk_mhz": 2143.0, "power_w": 900.0, "temp_c": 50.0}]
    out = s.summary()
    assert out["samples"] == 2 and out["sclk_mhz"] == {"min": 2143.0, "mean": 2150.0, "max": 2157.0}

    def boom(*a, **k):
        raise FileNotFoundError("rocm-smi")
    monkeypatch.setattr(subprocess, "run", boom)
    s2 = bench.GpuStateSampler(0)
    assert s2._read() is None and s2.summary() is None


def test_bench_cfg_shared_prefix_flops():
    """bench.cfg_shared_gflop: the FLOPs of the CFG pair's shared prefix (computed once by the engine) against the oracle's own
    layer shapes -- conv_in, the first ResnetBlock2D's two 3x3 convolutions, the first transformer's proj_in, q / k / v, out
    projection and self-attention at SD1.5's 64 x 64 level; SDXL has no attention at level 0 (nothing shared)."""
    import bench
    hw, c = 64 * 64, 320
    want = (2 * hw * c * 36 + 2 * (2 * hw * c * 9 * c) + 5 * (2 * hw * c * c) + 4 * hw * hw * c) * 1e-9
    got = bench.cfg_shared_gflop("sd15", 512, 512)
    assert abs(got - want) < 1e-9 and 0.02 < got / (2 * bench.UNET_GFLOP_PER_SAMPLE["sd15"]) < 0.03
    assert bench.cfg_shared_gflop("sdxl", 1024, 1024) == 0.0
