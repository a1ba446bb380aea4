#!/bin/bash
# Round 5: ONE GPU call that takes every measured artefact kept under profiles/r05_* (then, here:
# `python tools/collect_profiles.py gpurun_out/final r05`).  The GPU suite is a call of its own (tools/calls/suite.sh).
set -u
O=gpurun_out/final
rm -rf $O; mkdir -p $O
bash tools/pmc_run.sh conv0 $O/pmc_conv0 > $O/pmc_conv0.txt 2>&1
rm -rf $O/pmc_conv0/
bash tools/profile_bench.sh $O/prof --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/prof.log 2>&1      # (no roofline block: its back-to-back loop of the metric kernel must not mix into the in-step table)
rm -rf $O/prof/raw
python tools/collect_profiles.py $O r05 --stats-only
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --family sdxl --steps 2 --warmup 1 --no-cpu-baseline --no-torch-baseline > $O/bench_sdxl.json 2> $O/bench_sdxl.err
python bench.py --family cn-inpaint --steps 2 --warmup 1 --no-torch-baseline > $O/bench_cn_inpaint.json 2> $O/bench_cn.err
python tools/kbench.py 2>&1 | grep -v amdgpu.ids > $O/kbench.txt
CID_CONV_H32=0 python tools/kbench.py --only gemm 2>&1 | grep "conv3" > $O/kbench_conv_halo_kernel.txt
python tools/xattn_levels.py 2>&1 | grep -v amdgpu.ids > $O/xattn_levels.txt
# same-box A/B of the round's structural change: conv3x3.hip on / off
ab() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.4f images/s  %.2f ms/generation' % ('$*', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2; do ab X=shipped; ab CID_CONV_H32=0; done
tail -1 $O/bench_default.json | cut -c1-400; cat $O/ab.txt
