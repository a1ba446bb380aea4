#!/bin/bash
# Round 6: ONE GPU call that takes the end-of-round artefacts kept under profiles/r06_* (then, here:
# `python tools/collect_profiles.py gpurun_out/final r06`).  The GPU suite (tools/calls/suite.sh), the counter passes of the GEMM
# family (tools/calls/r06_pmc.sh) and the round's A/B calls (tools/calls/r06_*.sh) are calls of their own.
set -u
O=gpurun_out/final
rm -rf $O; mkdir -p $O
# PMC passes of the metric kernel first: bench.py reports roofline.traffic only from a summary taken on the kernel sources it runs
bash tools/pmc_run.sh xattn3 $O/pmc_xattn > $O/pmc_xattn.txt 2>&1
rm -rf $O/pmc_xattn/
python tools/collect_profiles.py $O r06 --pmc-only
bash tools/profile_bench.sh $O/prof --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/prof.log 2>&1      # (no roofline block: its back-to-back loop of the metric kernel must not mix into the in-step table)
rm -rf $O/prof/raw
bash tools/profile_bench.sh $O/prof_sdxl --family sdxl --steps 2 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-roofline > $O/prof_sdxl.log 2>&1
rm -rf $O/prof_sdxl/raw
python tools/collect_profiles.py $O r06 --stats-only
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --family sdxl --steps 2 --warmup 1 --no-cpu-baseline --no-torch-baseline > $O/bench_sdxl.json 2> $O/bench_sdxl.err
python bench.py --family cn-inpaint --steps 2 --warmup 1 --no-torch-baseline > $O/bench_cn_inpaint.json 2> $O/bench_cn.err
python tools/kbench.py 2>&1 | grep -v amdgpu.ids > $O/kbench.txt
python tools/xattn_levels.py 2>&1 | grep -v amdgpu.ids > $O/xattn_levels.txt
# same-box A/B of the round's switches
# (with the reported shader clock and socket power of each run: DESIGN.md 4.8 -- a denser kernel is answered with a lower clock)
ab() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); g = d.get('gpu_state') or {}
m = lambda k: (g.get(k) or {}).get('mean')
print('%-40s %.4f images/s  %.2f ms/generation  sclk %s MHz  %s W' % ('$*', d['value'], d['ms_per_step'], m('sclk_mhz'), m('power_w')))" >> $O/ab.txt; }
for i in 1 2; do ab X=shipped; ab CID_XCD_2D=0; ab CID_GEGLU_H32=0; ab CID_GEMM_PREFER128=0; ab CID_GEGLU_FOLD_MAX=2048; ab CID_CONV_H32=0; ab CID_XCD_2D=0 CID_GEGLU_H32=0 CID_GEMM_PREFER128=0 CID_CONV_H32=0; done
tail -1 $O/bench_default.json | cut -c1-400; cat $O/ab.txt
