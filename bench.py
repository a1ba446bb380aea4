#!/usr/bin/env python
"""bench.py -- the driver contract.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): 512x512 SD1.5 images/sec/node at 50 DDIM steps, plus the fused ID
cross-attention kernel's fraction of the MFMA roofline.
One "step" = one full generation of the local batch: cross-attention K/V projection of the
three embed sets + 50 x [UNet on the CFG batch 2B + CFG combine + DDIM update].  Inputs are
resident in HBM before the timed region.  N = 1 runs BASELINE.json configs[1] (SD1.5, 512x512,
50 steps, batch 4); N > 1 runs configs[2]'s shard, 8 images per GPU (weak scaling, images are
independent: no collective inside the timed region; one RCCL weight broadcast before it), and adds
configs[3]'s shard (SDXL 1024x1024, 30 steps, 2 images per GPU) as `secondary` on every rank count.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: dense fp16/bf16 MFMA
MFMA_PEAK_CLOCK_MHZ = 2400.0    # ... at the chip's maximum clock
UNET_GFLOP_PER_SAMPLE = {"sd15": 804.2, "sdxl": 6769.0}      # whole UNet forward per sample, LoRA merged


class GpuStateSampler:
    """Shader clock / socket power / temperature of the GPU while a region runs: a thread that calls `rocm-smi --json` in a
    loop (about two samples a second; the timed work is hipGraph replays, the host is idle) -- the chip clocks to its power
    budget (MI355X_MICROARCH.md "DVFS give-back"), so a rate without the clock it was measured at explains nothing."""

    def __init__(self, device_index: int = 0):
        import shutil
        import threading
        self.exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        self.card = f"card{device_index}"
        self.samples = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        import subprocess
        try:
            r = subprocess.run([self.exe, "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout[r.stdout.index("{"):])
            c = d.get(self.card) or next(iter(d.values()))
            num = lambda v: float("".join(ch for ch in str(v) if ch.isdigit() or ch == "."))
            out = {}
            for k, v in c.items():
                kl = k.lower()
                if kl.startswith("sclk clock speed"):
                    out["sclk_mhz"] = num(v)
                elif "power" in kl and "(w)" in kl:
                    out["power_w"] = num(v)
                elif "junction" in kl:
                    out["temp_c"] = num(v)
            return out or None
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            smp = self._read()
            if smp:
                self.samples.append(smp)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=15)

    def summary(self):
        """min / mean / max of every quantity over the samples taken inside the region (None without rocm-smi)"""
        if not self.samples:
            return None
        out = {"samples": len(self.samples)}
        for key in ("sclk_mhz", "power_w", "temp_c"):
            v = [s[key] for s in self.samples if key in s]
            if v:
                out[key] = {"min": round(min(v), 1), "mean": round(sum(v) / len(v), 1), "max": round(max(v), 1)}
        return out


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--family", default="sd15", choices=["sd15", "sdxl", "cn-inpaint"],
                   help="cn-inpaint = BASELINE config 5: SD1.5 + native ControlNet encoder + inpaint blend, batch 8")
    p.add_argument("--batch-per-gpu", type=int, default=None)
    p.add_argument("--ddim-steps", type=int, default=None)
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-full", action="store_true", help="time BASELINE config 1 (4 steps, B=1) in full on the CPU (the default for sd15)")
    p.add_argument("--cpu-baseline-short", action="store_true", help="2 DDIM steps at B=1, extrapolated, instead of config 1 in full")
    p.add_argument("--no-torch-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-secondary", action="store_true",
                   help="skip the secondary workloads (SDXL batch 2, ControlNet-inpaint batch 8, SD1.5 batch 8) of the default N=1 run")
    p.add_argument("--lora-rank", type=int, default=128)
    return p.parse_args()


def cfg_shared_gflop(family: str, H: int, W_: int) -> float:
    """GFLOP of the part of one UNet forward that the engine computes ONCE per CFG pair (HipUNet.forward_tokens: until the
    first cross-attention both halves of the CFG batch are the same computation on the same data -- conv_in, the first
    ResnetBlock2D, the first transformer's proj_in / q,k,v / self-attention / out projection).  The algorithmic count of
    SURVEY.md 8(d) charges them to both halves; `roofline.step` reports both numbers.  SDXL has no attention at level 0:
    nothing is shared there."""
    if family != "sd15":
        return 0.0
    hw, c = (H // 8) * (W_ // 8), 320
    conv3 = 2.0 * hw * c * 9 * c
    return (2.0 * hw * c * 9 * 4 + 2 * conv3 + 2.0 * hw * c * c * 5 + 4.0 * hw * hw * c) * 1e-9   # conv_in, 2 convs, proj_in + qkv + out, attention


def xattn_flops(B2, N, C, L=81):
    """4 N C^2 + 4 N L C per (sample, layer), LoRA merged, K/V precomputed."""
    return B2 * (4.0 * N * C * C + 4.0 * N * L * C)


def measure_xattn_roofline(unet, B2, N, C, heads, iters=50, sample_clock=True):
    """Live HIP-event timing of the fused ID cross-attention kernel at the UNet's level-0 shape -- the very
    instantiation, weights and packed K/V the denoise loop launches -- on the stream the kernel runs on
    (ops launch on torch's current stream, which is what torch.cuda.Event records on)."""
    from consistentid_amd import ops
    dev = unet.device
    layer = next(b for b in unet.packed.xattn_layers if unet.W[f"{b}.attn2.bo"].shape[0] == C)
    W, ctx = unet.W, unet._ctx
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B2, N, C, generator=g, device=dev).half()
    kvrow = (torch.arange(B2, dtype=torch.int32, device=dev) % ctx.rows).contiguous()
    heads_of = unet._heads_of(layer)
    assert heads_of == heads
    xt = x.view(B2 * N, C)

    def run():      # the launch sequence the denoise step uses for this layer (HipUNet.cross_attention)
        return unet.cross_attention(layer, xt, B2, N, C, heads, kvrow)

    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    # the clock this kernel sustains: the same launch back to back for ~1.5 s with the sampler running (the 50-launch timing
    # above is over in 1.5 ms, shorter than one rocm-smi call)
    clock = None
    if sample_clock:
        n_long = max(iters, int(1.5 / (ms * 1e-3)))
        with GpuStateSampler(dev.index or 0) as smp:
            for _ in range(n_long):
                run()
            torch.cuda.synchronize()
        clock = smp.summary()
    L = ctx.n_txt + ctx.n_ip
    fl = xattn_flops(B2, N, C, L)
    achieved = fl / (ms * 1e-3) / 1e12
    gen = int(ctx.v2.get(layer) or 0)       # generation of the fused kernel serving this layer (0: not the level-0 geometry)
    kernel = f"id_xattn{gen}_kernel<{ctx.n_txt},{ctx.n_ip}>" if gen else f"id_xattn_kernel<{C},{C // heads},...>"
    path = unet.cross_attention_path(layer, C, B2, N)
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (they cannot be sampled in-process).  The committed
    # summary is keyed by kernel + shape and carries the digest of the kernel sources it was measured on: a summary taken
    # from other code is NOT reported (null) instead of silently going stale.
    traffic, note = None, "no PMC summary for this kernel/shape"
    try:
        with open(os.path.join(ROOT, "profiles", "roofline_pmc.json")) as f:
            pmc = json.load(f)
        ent = pmc.get(f"{kernel}@B2={B2},N={N},C={C}")
        if ent is not None:
            if ent.get("kernel_digest") == kernel_digest():
                traffic, note = ent.get("hbm_bytes"), f"rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, {ent.get('source', 'profiles/')}"
            elif gen == 3 and ent.get("metric_kernel_digest") == kernel_digest(METRIC_KERNEL_SOURCES):
                traffic = ent.get("hbm_bytes")
                note = (f"rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, {ent.get('source', 'profiles/')} (taken at library digest "
                        f"{ent.get('kernel_digest')}; this kernel's own sources {' '.join(METRIC_KERNEL_SOURCES)} unchanged since)")
            else:
                note = "committed PMC summary was taken from different kernel sources (digest mismatch): not reported"
    except OSError:
        pass
    # the same kernel's average launch inside the denoise step (rocprofv3 --kernel-trace of this bench command, committed
    # under profiles/ with the digest of the kernel sources it was taken on): reported beside the back-to-back timing above
    in_step = None
    try:
        import csv
        import glob
        for stats_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")), reverse=True):
            with open(stats_path) as f:
                head = f.readline()
                if kernel_digest() not in head:
                    continue
                for row in csv.DictReader(f):
                    if gen and f"id_xattn{gen}_kernelILi{ctx.n_txt}ELi{ctx.n_ip}E" in row["kernel"]:
                        us = float(row["avg_us"])
                        in_step = {"avg_launch_us": us, "calls": int(row["calls"]), "frac": round(fl / (us * 1e-6) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                                   "source": os.path.relpath(stats_path, ROOT)}
                        break
            if in_step:
                break
    except (OSError, KeyError, ValueError):
        pass
    # algorithmic bytes: x in + out once, Wq + Wo once, K/V of the B2 context rows
    alg_bytes = 2 * B2 * N * C * 2 + 2 * C * C * 2 + B2 * 2 * L * C * 2
    # How far the 2.5 PFLOP/s roof is from what this kernel could ever issue: the roof assumes 2.4 GHz and no padding.  The
    # generation-3 kernel issues 592 MFMAs of 16x16x32 per wave, four waves per 64-token tile (40-wide heads padded to 48 in
    # QK^T / PV: csrc/xattn3.hip), and the chip runs it at the sampled clock, not at 2.4 GHz.
    sustained = None
    if gen == 3:
        issued = (B2 * N // 64) * 4 * 592 * 16384.0
        share = fl / issued
        sclk = (clock or {}).get("sclk_mhz", {}).get("mean")
        sustained = {"mfma_flops_issued_per_launch": issued, "non_padded_share": round(share, 4), "sclk_mhz": sclk,
                     "note": "peak x (reported clock / 2400 MHz) x (algorithmic / issued MFMA flops); sclk = what rocm-smi reports "
                             "while the kernel runs back to back -- under dense MFMA load the cycle counter shows a lower effective "
                             "clock than rocm-smi does (1.4 vs 1.65 GHz, profiles/r05_power_clock.txt), so this roof is an upper bound"}
        if sclk:
            roof = MFMA_F16_PEAK_TFLOPS * sclk / MFMA_PEAK_CLOCK_MHZ * share
            sustained["roof_tflops"] = round(roof, 1)
            sustained["frac_of_sustained"] = round(achieved / roof, 4)
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_F16_PEAK_TFLOPS, 4), "frac_of_sustained": (sustained or {}).get("frac_of_sustained"),
            "sustained": sustained, "gpu_state": clock, "traffic": traffic, "traffic_note": note,
            "algorithmic_bytes": alg_bytes, "kernel": kernel, "launches": path, "shape": {"B2": B2, "N": N, "C": C, "L": L},
            "flops_per_launch": fl, "avg_launch_us": round(ms * 1e3, 2), "in_step": in_step, "kernel_digest": kernel_digest()}


def _oracle_unet(family: str, device, dtype):
    """the oracle's UNet + ConsistentID processors (reference path restated in plain PyTorch) with N(0, 0.02) weights"""
    from oracle import processors as oproc
    from oracle import unet as ounet
    cfg = ounet.sd15_config() if family == "sd15" else ounet.sdxl_config()
    with torch.device("meta"):
        m = ounet.UNet2DConditionModel(cfg)
        oproc.set_ip_adapter(m, lora_rank=128)
    m = m.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g, device=device) * 0.02)
    return m.to(dtype).eval(), cfg


def _reference_loop_rate(m, cfg, family, ddim_steps, B, device, dtype, min_steps, budget_s, sync):
    """time `min_steps`+ steps of the reference denoise loop (UNet on the CFG batch 2B + CFG + DDIM, the oracle's
    restatement of pipline_StableDiffusion_ConsistentID.py:537-571) and scale linearly to a generation"""
    from oracle import ddim as oddim
    hw = cfg.sample_size
    lat = torch.randn(B, 4, hw, hw, device=device, dtype=dtype)
    ehs = torch.randn(2 * B, 81, cfg.cross_attention_dim, device=device, dtype=dtype)
    kw = {}
    if family == "sdxl":
        kw = dict(added_cond_kwargs={"text_embeds": torch.randn(2 * B, 1280, device=device, dtype=dtype),
                                     "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]], device=device,
                                                              dtype=dtype).repeat(2 * B, 1)})
    sch = oddim.DDIMScheduler()
    sch.set_timesteps(ddim_steps)

    def step(n, lat):
        t = int(sch.timesteps[n % ddim_steps])
        eps = m(torch.cat([lat] * 2), t, ehs, **kw).sample
        eu, ec = eps.chunk(2)
        return sch.step((eu + 5.0 * (ec - eu)).float(), t, lat.float()).to(dtype)

    with torch.no_grad():
        if sync is not None:        # GPU arm: one untimed step (allocator, kernel selection)
            lat = step(0, lat)
            sync()
        n, t0 = 0, time.perf_counter()
        while True:
            lat = step(n, lat)
            n += 1
            if sync is not None:
                sync()
            el = time.perf_counter() - t0
            if n >= min_steps and (el > budget_s or n >= ddim_steps):
                break
    return el / n, n


def cpu_baseline(family: str, ddim_steps: int, full_config1: bool = False):
    """The fp32 oracle (CPU restatement of the reference path) timed on this box's host cores (torch's intra-op pool =
    the physical cores; `cores` reports that count).  SD1.5 (the default run): BASELINE config 1 IN FULL -- 4 DDIM steps,
    B = 1 (CFG batch 2), ~46 s on 128 cores -- `config1_seconds` is that measurement, `value` the same per-step rate
    scaled to the headline's step count (labelled extrapolated).  Other families / --cpu-baseline-short: 2 steps,
    extrapolated.  Reported, not a target."""
    # threads actually used = torch's intra-op pool, which defaults to the PHYSICAL cores.  Forcing it to os.cpu_count()
    # (the logical CPUs, 2 per core on the GPU boxes) was measured here: 291 s per step on 256 threads against 12.6 s on
    # 128 -- oversubscribing the SMT siblings is not "more cores".
    cores = torch.get_num_threads()
    logical = os.cpu_count() or cores
    m, cfg = _oracle_unet(family, "cpu", torch.float32)
    steps = 4 if full_config1 else 2
    per_step, n = _reference_loop_rate(m, cfg, family, 4 if full_config1 else ddim_steps, 1, "cpu", torch.float32,
                                       min_steps=steps, budget_s=0.0, sync=None)
    res = {"value": round(1.0 / (per_step * ddim_steps), 6), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": f"{n} DDIM steps of the fp32 oracle UNet at B=1 (CFG batch 2) on {cores} host threads "
                     f"({logical} logical CPUs on the box), {per_step:.2f} s/step, extrapolated linearly to {ddim_steps} steps"}
    if full_config1:
        res["config1_seconds"] = round(per_step * n, 2)
        res["sample"] = (f"BASELINE config 1 in full: 4 DDIM steps, B=1 (CFG batch 2), {per_step * n:.1f} s on {cores} host "
                         f"threads ({logical} logical CPUs); value = the same rate scaled to {ddim_steps} steps")
    return res


def torch_fp16_baseline(family: str, ddim_steps: int, B: int, device):
    """BASELINE.md section 4 comparator: the reference path as stock PyTorch-ROCm eager fp16 on ONE GPU -- the
    oracle's modules .half() (torch SDPA / rocBLAS / MIOpen kernels, no code of this repository) -- at the bench's
    per-GPU batch; >= 2 timed steps after one warm-up step, scaled linearly to a generation."""
    m, cfg = _oracle_unet(family, device, torch.float16)
    per_step, n = _reference_loop_rate(m, cfg, family, ddim_steps, B, device, torch.float16, min_steps=3, budget_s=3.0,
                                       sync=torch.cuda.synchronize)
    del m
    torch.cuda.empty_cache()
    return {"value": round(B / (per_step * ddim_steps), 4), "unit": "images/s", "kind": "stock PyTorch-ROCm eager fp16",
            "sample": f"{n} DDIM steps of the oracle UNet (.half(), torch {torch.__version__}) at B={B} (CFG batch {2 * B}), "
                      f"{per_step * 1e3:.1f} ms/step, extrapolated linearly to {ddim_steps} steps"}


def respawn_under_torchrun(a):
    """`python bench.py --gpus N` without a launcher: re-exec as N ranks of ONE node under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1).  With WORLD_SIZE already set (the driver's own torchrun
    command) this is a no-op."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


# the translation unit of the roofline kernel (xattn3.hip and the headers it includes): a PMC summary of THAT kernel stays
# valid while these files are unchanged, whatever happens to the other kernels of the library
METRIC_KERNEL_SOURCES = ("common.h", "xattn3.hip", "xattn_frag.h")


def kernel_digest(only=None) -> str:
    """sha256 over the kernel sources (all of csrc/, or the files named in ``only``): ties committed profile summaries to the
    code they were taken from"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "consistentid_amd", "csrc")
    for n in sorted(only if only is not None else os.listdir(d)):
        h.update(n.encode())
        with open(os.path.join(d, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def run_workload(a, family, cn, bpg, steps, warmup, rank, world, dev, ddim_override=None, sample=False):
    """Build the engine(s) of one workload, run `warmup` untimed + `steps` timed generations of the local batch between
    barrier + synchronize pairs; returns (seconds max-over-ranks, description dict, unet, pipe)."""
    from consistentid_amd import distributed, pipeline, synth, unet_spec
    from consistentid_amd.unet import HipUNet
    from consistentid_amd.weights import PackedUNet
    import torch.distributed as dist
    if family == "sd15":
        cfg, H, W_, ddim_steps, guidance, merge = unet_spec.sd15_config(), 512, 512, 50, 5.0, 30
    else:
        cfg, H, W_, ddim_steps, guidance, merge = unet_spec.sdxl_config(), 1024, 1024, 30, 7.5, 18
    ddim_steps = ddim_override or ddim_steps
    global_batch = bpg * world

    # ---- weights: rank 0 builds + packs, everyone else receives the arena over RCCL/xGMI
    if rank == 0:
        sd = synth.random_unet_state_dict(cfg, seed=0, device=dev)
        ad = synth.random_adapter_state_dict(cfg, sd, rank=a.lora_rank, seed=1, device=dev)
        unet = HipUNet(cfg, sd, ad, device=dev)
        del sd, ad
        torch.cuda.empty_cache()
    if world > 1:
        meta = [unet.packed.meta() if rank == 0 else None]
        dist.broadcast_object_list(meta, src=0)
        named = distributed.broadcast_weights(unet.W if rank == 0 else {}, dev, src=0)
        if rank != 0:
            unet = HipUNet(cfg, device=dev, packed=PackedUNet.from_tensors(cfg, named, meta[0], dev))
    pipe_cls = pipeline.ConsistentIDStableDiffusionPipeline if family == "sd15" else \
        pipeline.ConsistentIDStableDiffusionXLPipeline
    if cn:
        from consistentid_amd.controlnet import HipControlNet
        if rank == 0:
            cnet = HipControlNet(cfg, synth.random_controlnet_state_dict(cfg, seed=3, device=dev), device=dev)
        if world > 1:
            meta = [cnet.packed.meta() if rank == 0 else None]
            dist.broadcast_object_list(meta, src=0)
            named = distributed.broadcast_weights(cnet.W if rank == 0 else {}, dev, src=0)
            if rank != 0:
                cnet = HipControlNet(cfg, device=dev, packed=PackedUNet.from_tensors(cfg, named, meta[0], dev))
        pipe = pipeline.StableDiffusionControlNetInpaintConsistentIDPipeline(unet, controlnet=cnet,
                                                                               use_graph=not a.no_graph)
    else:
        pipe = pipe_cls(unet, use_graph=not a.no_graph)
    kw = workload_inputs(family, cn, cfg, bpg, H, W_, ddim_steps, guidance, merge, rank, world, dev)
    dt, out = time_generations(pipe, kw, steps, warmup, world, dev, sample=sample, poll=(rank == 0))
    desc = {"family": family, "cn": cn, "cfg": cfg, "H": H, "W": W_, "ddim_steps": ddim_steps, "merge": merge,
            "bpg": bpg, "global_batch": global_batch}
    return dt, desc, unet, pipe


def workload_inputs(family, cn, cfg, bpg, H, W_, ddim_steps, guidance, merge, rank, world, dev):
    """every rank derives its own images from (seed + global image index)"""
    from consistentid_amd import distributed, synth
    lo, hi = distributed.shard_range(bpg * world, rank, world)
    inp = synth.random_inputs(cfg, hi - lo, H, W_, seed_latents=2024 + lo, seed_embeds=1 + lo, device=dev)
    pe = torch.cat([inp["null"], inp["augmented"], inp["text"]])
    kw = dict(prompt_embeds=pe, latents=inp["latents"], num_inference_steps=ddim_steps, guidance_scale=guidance,
              start_merge_step=merge, output_type="latent")
    if cn:
        g = torch.Generator(device=dev).manual_seed(77 + lo)
        n, h8, w8 = hi - lo, H // 8, W_ // 8
        kw.update(control_image=torch.rand(n, 3, H, W_, generator=g, device=dev).half(), controlnet_conditioning_scale=0.5,
                  image_latents=torch.randn(n, 4, h8, w8, generator=g, device=dev).half(),
                  noise=torch.randn(n, 4, h8, w8, generator=g, device=dev).half(),
                  mask_latents=(torch.rand(n, 1, h8, w8, generator=g, device=dev) > 0.5).half())
    if family == "sdxl":
        kw.update(pooled_prompt_embeds=inp["pooled_augmented"], pooled_prompt_embeds_text_only=inp["pooled_text"],
                  negative_pooled_prompt_embeds=inp["pooled_null"], add_time_ids=inp["time_ids"])
    return kw


def time_generations(pipe, kw, steps, warmup, world, dev, sample=False, poll=True):
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        out = pipe(**kw)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = pipe(**kw)
    barrier()
    dt = time.perf_counter() - t0
    if sample:
        # clock / power / temperature while the SAME generations run, in two extra untimed ones on every rank (rank 0 polls):
        # polling rocm-smi beside the timed region was measured to slow it by 12 % (545 -> 625 ms per generation)
        smp = GpuStateSampler(dev.index or 0) if poll else None
        if smp:
            smp.__enter__()
        for _ in range(2):
            pipe(**kw)
        torch.cuda.synchronize()
        if smp:
            smp.__exit__()
            time_generations.last_gpu_state = smp.summary()
            if time_generations.last_gpu_state:
                time_generations.last_gpu_state["note"] = ("rocm-smi polled during two extra untimed generations of this workload "
                                                           "(polling slows the run: these generations are not the timed ones)")
        barrier()
    assert torch.isfinite(out.images.float()).all(), "non-finite latents"
    if world > 1:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    return dt, out


time_generations.last_gpu_state = None


def workload_name(d, lora_rank, no_graph):
    head = "sd15 ControlNet-inpaint (native ControlNet encoder, scale 0.5, mask blend)" if d["cn"] else d["family"]
    return (f"{head} ConsistentID {d['H']}x{d['W']}, {d['ddim_steps']} DDIM steps, batch {d['bpg']}/GPU "
            f"(global {d['global_batch']}), CFG batch {2 * d['bpg']}, LoRA rank {lora_rank} merged, "
            f"start_merge_step {d['merge']}, hipGraph {'off' if no_graph else 'on'}")


def secondary_workloads(a, unet_sd15, dev):
    """The other single-GPU shapes of BASELINE.json's configs, one warm + two timed generations each (default N = 1 run
    only): config 3's per-GPU shape (SD1.5 batch 8), config 5 (SD1.5 + native ControlNet + inpaint blend, batch 8) and
    config 4's per-GPU shape (SDXL 1024^2, 30 DDIM steps, batch 2).  Same timing brackets as the headline."""
    from consistentid_amd import pipeline, unet_spec
    out = {}

    def entry(dt, d, steps):
        return {"images_per_s": round(d["bpg"] * steps / dt, 4), "ms_per_generation": round(dt / steps * 1e3, 1),
                "workload": workload_name(d, a.lora_rank, a.no_graph), "timed_generations": steps, "warmup": 1}

    # SD1.5 batch 8: the headline's engine, a fresh pipeline (its graph is captured for the new batch)
    cfg = unet_spec.sd15_config()
    pipe = pipeline.ConsistentIDStableDiffusionPipeline(unet_sd15, use_graph=not a.no_graph)
    kw = workload_inputs("sd15", False, cfg, 8, 512, 512, 50, 5.0, 30, 0, 1, dev)
    dt, _ = time_generations(pipe, kw, 2, 1, 1, dev)
    out["sd15_b8"] = entry(dt, {"family": "sd15", "cn": False, "H": 512, "W": 512, "ddim_steps": 50, "merge": 30,
                               "bpg": 8, "global_batch": 8}, 2)
    del pipe, kw
    torch.cuda.empty_cache()
    for key, family, cn, bpg in (("cn_inpaint_b8", "sd15", True, 8), ("sdxl_b2_30steps", "sdxl", False, 2)):
        dt, d, u, p = run_workload(a, family, cn, bpg, 2, 1, 0, 1, dev)
        out[key] = entry(dt, d, 2)
        del u, p
        torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    respawn_under_torchrun(a)
    from consistentid_amd import distributed
    import torch.distributed as dist

    rank, local_rank, world = distributed.init_process_group()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # one rank per GPU; CID_BENCH_SHARE_GPU=1 (test rigs with a single GPU) folds the ranks onto device 0
    dev_index = local_rank % torch.cuda.device_count() if os.environ.get("CID_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")

    cn = a.family == "cn-inpaint"
    default_shape = a.family == "sd15" and a.batch_per_gpu is None and a.ddim_steps is None
    default_run = default_shape and world == 1
    if cn:
        a.family = "sd15"
        a.no_cpu_baseline = True      # the cpu_baseline leg times the plain SD1.5 loop
    # N = 1: BASELINE config 2 (batch 4).  N > 1: config 3's shard, 8 images per GPU (64 on 8 GPUs) -- with the SDXL shard of
    # config 4 (2 images per GPU, 16 on 8) as `secondary` on every rank count, so that a scaling run lands on BASELINE shapes
    bpg = a.batch_per_gpu or (8 if cn else (4 if world == 1 else 8) if a.family == "sd15" else 2)
    dt, d, unet, pipe = run_workload(a, a.family, cn, bpg, a.steps, a.warmup, rank, world, dev, a.ddim_steps, sample=True)
    cfg, H, W_, ddim_steps, global_batch = d["cfg"], d["H"], d["W"], d["ddim_steps"], d["global_batch"]
    gpu_state = time_generations.last_gpu_state
    # the metric kernel's roofline block, on rank 0, while the headline's engine still exists (N > 1 frees it for the SDXL shard)
    xroof = None
    if rank == 0 and not a.no_roofline:
        c0 = cfg.block_out_channels[0] if a.family == "sd15" else cfg.block_out_channels[1]
        heads = cfg.num_attention_heads[0] if a.family == "sd15" else cfg.num_attention_heads[1]
        n0 = (H // 8) * (W_ // 8) if a.family == "sd15" else (H // 16) * (W_ // 16)
        xroof = measure_xattn_roofline(unet, 2 * bpg, n0, c0, heads)
    sdxl_scaling = None
    if world > 1 and default_shape and not cn and not a.no_secondary:
        del pipe, unet
        pipe = unet = None
        torch.cuda.empty_cache()
        dt2, d2, u2, p2 = run_workload(a, "sdxl", False, 2, 2, 1, rank, world, dev)
        sdxl_scaling = {"images_per_s": round(d2["global_batch"] * 2 / dt2, 4), "ms_per_generation": round(dt2 / 2 * 1e3, 1),
                        "workload": workload_name(d2, a.lora_rank, a.no_graph), "timed_generations": 2, "warmup": 1,
                        "n_gpus": world, "scaling": "weak"}
        del u2, p2
        torch.cuda.empty_cache()

    if rank == 0:
        value = global_batch * a.steps / dt
        res = {
            "metric": f"512x512 SD1.5 ControlNet-inpaint images/sec/node @{ddim_steps} DDIM steps" if cn
            else f"512x512 SD1.5 images/sec/node @{ddim_steps} DDIM steps" if a.family == "sd15"
            else f"1024x1024 SDXL images/sec/node @{ddim_steps} DDIM steps",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": workload_name(d, a.lora_rank, a.no_graph),
                       "global_batch": global_batch, "parallelism": f"dp{world} (images sharded, no in-step collective)"},
            # clock / power / temperature of GPU 0 sampled (rocm-smi) while the timed generations ran
            "gpu_state": gpu_state,
            # every A/B switch of the library / engine present in the environment of this run (none = the shipped defaults)
            "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("CID_")},
        }
        if world > 1:
            res["config"]["note"] = ("N > 1 runs BASELINE config 3's shard (8 images per GPU); the N = 1 line is config 2 (batch 4) "
                                     "and carries config 3's shard as secondary.sd15_b8 -- compare per-GPU rates with that")
        # the whole denoise step against the MFMA roof: every image costs 2 (CFG) x ddim_steps UNet forwards of SURVEY 8(d)'s FLOPs
        if not cn:
            step_tf = value * 2 * ddim_steps * UNET_GFLOP_PER_SAMPLE[a.family] * 1e-3 / world      # TFLOP/s per GPU
            shared = cfg_shared_gflop(a.family, H, W_) if os.environ.get("CID_CFG_DEDUP", "1") != "0" else 0.0
            exec_tf = value * ddim_steps * (2 * UNET_GFLOP_PER_SAMPLE[a.family] - shared) * 1e-3 / world
            step = {"flops_per_image": 2 * ddim_steps * UNET_GFLOP_PER_SAMPLE[a.family] * 1e9, "achieved_tflops_per_gpu": round(step_tf, 1),
                    "frac": round(step_tf / MFMA_F16_PEAK_TFLOPS, 4),
                    # the CFG pair's shared prefix is computed once: what the GPU executes is a little less than the algorithmic count
                    "executed_flops_per_image": ddim_steps * (2 * UNET_GFLOP_PER_SAMPLE[a.family] - shared) * 1e9,
                    "executed_tflops_per_gpu": round(exec_tf, 1), "executed_frac": round(exec_tf / MFMA_F16_PEAK_TFLOPS, 4)}
            sclk = (gpu_state or {}).get("sclk_mhz", {}).get("mean")
            if sclk:
                step["frac_of_peak_at_sampled_clock"] = round(step_tf / (MFMA_F16_PEAK_TFLOPS * sclk / MFMA_PEAK_CLOCK_MHZ), 4)
        if xroof is not None:
            res["roofline"] = xroof
            if not cn:
                res["roofline"]["step"] = step
        elif not cn:
            res["roofline"] = {"step": step}
        pipe = None
        if default_run and not cn and not a.no_secondary:
            res["secondary"] = secondary_workloads(a, unet, dev)
        if sdxl_scaling is not None:
            res["secondary"] = {"sdxl_b2_30steps": sdxl_scaling}
        if world == 1 and not cn and not a.no_torch_baseline:
            unet = None
            torch.cuda.empty_cache()
            res["torch_fp16_baseline"] = torch_fp16_baseline(a.family, ddim_steps, bpg, dev)
        if world == 1 and not a.no_cpu_baseline:
            full = (a.family == "sd15" and not a.cpu_baseline_short) or a.cpu_baseline_full
            res["cpu_baseline"] = cpu_baseline(a.family, ddim_steps, full)
        elif world > 1:
            # the CPU leg is timed on rank 0 of the N = 1 run only (it would hold N - 1 GPUs idle for a minute here)
            res["cpu_baseline"] = {"value": None, "unit": "images/s", "kind": "port",
                                   "sample": "not timed at N > 1: see the cpu_baseline object of the N = 1 line (same box class; "
                                             "BASELINE.md section 3 lists the measured values)"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
