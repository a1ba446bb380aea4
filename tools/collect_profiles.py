#!/usr/bin/env python
"""Copy the summaries of one `tools/refresh_profiles.sh` run into profiles/, each stamped with the commit and the
kernel-source digest it was measured on, and rebuild profiles/roofline_pmc.json (the HBM traffic of the roofline
kernel that bench.py reports -- only while the digest still matches the sources).
usage: python tools/collect_profiles.py gpurun_out/final r02
       python tools/collect_profiles.py gpurun_out/final r02 --pmc-only     (on the GPU box, between the PMC passes and the
                                                bench run of tools/refresh_profiles.sh: only profiles/roofline_pmc.json)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src, tag = sys.argv[1], sys.argv[2]
pmc_only = "--pmc-only" in sys.argv
stats_only = "--stats-only" in sys.argv        # only the in-step kernel tables (bench.py's roofline.in_step reads them)
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "(no git here)"
dirty = subprocess.run(["git", "status", "--porcelain", "--", "consistentid_amd", "bench.py"], cwd=ROOT, capture_output=True,
                       text=True).stdout.strip()
digest = bench.kernel_digest()
stamp = f"commit {commit}{' + uncommitted changes' if dirty else ''}, kernel sources sha256[:16] {digest}, 1x MI355X via gpurun"
out = os.path.join(ROOT, "profiles")


def put(name, text, comment="#"):
    with open(os.path.join(out, name), "w") as f:
        if comment:
            f.write(f"{comment} {stamp}\n")
        f.write(text)


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


for name in ("bench_default", "bench_default_final", "bench_default_run1", "bench_sdxl", "bench_sdxl_lnfold_always", "bench_cn_inpaint",
             "bench_sd15_batch8", "bench_default_lock", "bench_default_again", "bench_sdxl_again"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p) and not pmc_only and not stats_only:
        d = last_json(p)
        d["_stamp"] = stamp
        put(f"{tag}_{name}.json", json.dumps(d, indent=1) + "\n", comment="")
for a, b in (("prof/kernel_stats.csv", "bench_kernel_stats.csv"), ("kbench.txt", "kbench.txt"),
             ("pmc_xattn.txt", "pmc_xattn.txt"), ("xattn_trace.txt", "xattn_trace.txt"), ("xattn_levels.txt", "xattn_levels.txt"),
             ("pmc_conv0.txt", "pmc_conv0.txt"), ("conv_trace.txt", "conv_trace.txt"), ("conv_trace_lock.txt", "conv_trace_lock.txt"),
             ("kbench_lock.txt", "kbench_lockstep_build.txt"), ("abl.txt", "gemm_ablation.txt"), ("simd_map.txt", "simd_map.txt"),
             ("prof_sdxl/kernel_stats.csv", "bench_sdxl_kernel_stats.csv"), ("prof/kernel_shapes.csv", "bench_kernel_shapes.csv"),
             ("prof_sdxl/kernel_shapes.csv", "bench_sdxl_kernel_shapes.csv"), ("issue_rates.txt", "issue_rates.txt"),
             ("pmc_selfattn.txt", "pmc_selfattn.txt"), ("attn_ablation.txt", "attn_ablation.txt"), ("ab.txt", "ab_conv3x3.txt" if tag <= "r05" else "ab_switches.txt"), ("kbench_conv_halo_kernel.txt", "kbench_conv_halo_kernel.txt"),
             ("pytest_gpu.txt", "pytest_gpu.txt")):
    p = os.path.join(src, a)
    if os.path.exists(p) and not pmc_only and (not stats_only or a.startswith("prof/")):
        put(f"{tag}_{b}", open(p).read())
# HBM traffic of the roofline kernel from the PMC passes: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports
# half of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> hbm_bytes = (2 * FETCH + WRITE) * 1024 per launch
p = os.path.join(src, "pmc_xattn.txt")
if os.path.exists(p) and not stats_only:
    txt = open(p).read()
    get = lambda k: float(re.search(rf"{k}\s+([0-9.]+)", txt).group(1))
    fetch, write = get("FETCH_SIZE"), get("WRITE_SIZE")
    # (tools/pmc_one.py xattn3 runs the shape of the default bench line's roofline block)
    kname = re.search(r"id_xattn(\d)_kernelILi(\d+)ELi(\d+)E", txt)
    B2, N, C, L = 8, 4096, 320, 81
    key = f"id_xattn{kname.group(1)}_kernel<{kname.group(2)},{kname.group(3)}>@B2={B2},N={N},C={C}"
    d = {"algorithmic_bytes": 2 * B2 * N * C * 2 + 2 * C * C * 2 + B2 * 2 * L * C * 2}
    pmc = {"_comment": "HBM-side traffic of the roofline kernel from rocprofv3 --pmc passes (tools/pmc_run.sh xattn3; raw counters in "
                       f"{tag}_pmc_xattn.txt). FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a wide "
                       "coalesced read (MI355X_MICROARCH.md, HBM section): hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, per "
                       "launch. bench.py reports it only while kernel_digest matches the kernel sources (or, for the roofline kernel, "
                       "while metric_kernel_digest matches the files of its own translation unit).",
           key: {"fetch_size_kb": fetch, "write_size_kb": write, "hbm_bytes": int((2 * fetch + write) * 1024),
                 "algorithmic_bytes": d["algorithmic_bytes"], "kernel_digest": digest,
                 "metric_kernel_digest": bench.kernel_digest(bench.METRIC_KERNEL_SOURCES), "source": f"profiles/{tag}_pmc_xattn.txt",
                 "commit": commit}}
    put("roofline_pmc.json", json.dumps(pmc, indent=2) + "\n", comment="")
print("profiles/ updated:", stamp)
