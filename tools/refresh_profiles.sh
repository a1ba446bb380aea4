#!/bin/bash
# One GPU call that regenerates every measured artefact kept under profiles/ (run through gpurun, then
# `python tools/collect_profiles.py gpurun_out/final r03` copies the summaries into profiles/ with their stamps).
# usage: tools/refresh_profiles.sh [core|extra|all]   (two shorter GPU calls instead of one long one: core, then extra)
set -u
STAGE=${1:-all}
O=gpurun_out/final
if [ "$STAGE" != extra ]; then
rm -rf $O; mkdir -p $O
# PMC passes first: bench.py reports roofline.traffic only from a summary taken on the very kernel sources it runs
bash tools/pmc_run.sh xattn3 $O/pmc_xattn > $O/pmc_xattn.txt 2>&1
rm -rf $O/pmc_xattn/p*/
python tools/collect_profiles.py $O r03 --pmc-only
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/profile_bench.sh $O/prof --no-cpu-baseline --no-torch-baseline > $O/prof.log 2>&1
rm -rf $O/prof/raw
python tools/kbench.py > $O/kbench.txt 2>&1
python tools/xattn_levels.py 2>&1 | grep -v amdgpu.ids > $O/xattn_levels.txt
python bench.py --family sdxl --no-cpu-baseline > $O/bench_sdxl.json 2>/dev/null
CID_LN_FOLD=1 python bench.py --family sdxl --no-cpu-baseline --no-torch-baseline --no-roofline > $O/bench_sdxl_lnfold_always.json 2>/dev/null
python bench.py --family cn-inpaint > $O/bench_cn_inpaint.json 2>/dev/null
python bench.py --batch-per-gpu 8 --no-cpu-baseline --no-torch-baseline > $O/bench_sd15_batch8.json 2>/dev/null
CID_LIBRARY=consistentid_amd/libcid_trace.so python tools/x2_trace.py --gen 3 > $O/xattn_trace.txt 2>&1
tail -1 $O/bench_default.json | cut -c1-600
fi
[ "$STAGE" = core ] && exit 0
mkdir -p $O
# 3x3 convolution: counters, phase stamps of the shipped (half-slab offset) pipeline, and the lock-step build beside it
bash tools/pmc_run.sh conv0 $O/pmc_conv0 > $O/pmc_conv0.txt 2>&1
rm -rf $O/pmc_conv0/
CID_LIBRARY=consistentid_amd/libcid_ctr.so python tools/conv_trace.py 2>&1 | grep -v amdgpu.ids > $O/conv_trace.txt
CID_LIBRARY=consistentid_amd/libcid_ctr_lock.so python tools/conv_trace.py 2>&1 | grep -v amdgpu.ids > $O/conv_trace_lock.txt
CID_LIBRARY=$PWD/consistentid_amd/libcid_lock.so python tools/kbench.py --only gemm 2>&1 | grep -v amdgpu.ids > $O/kbench_lock.txt
CID_LIBRARY=$PWD/consistentid_amd/libcid_lock.so python bench.py --no-cpu-baseline --no-torch-baseline --no-roofline > $O/bench_default_lock.json 2>/dev/null
python bench.py --no-cpu-baseline --no-torch-baseline --no-roofline > $O/bench_default_again.json 2>/dev/null
CID_LIBRARY=$PWD/consistentid_amd/libcid_abl.so python tools/abl.py 2>&1 | grep -v amdgpu.ids > $O/abl.txt
./tools/probes/simd_map > $O/simd_map.txt 2>&1
bash tools/profile_bench.sh $O/prof_sdxl --family sdxl --no-cpu-baseline --no-torch-baseline --no-roofline > $O/prof_sdxl.log 2>&1
rm -rf $O/prof_sdxl/raw
tail -3 $O/conv_trace.txt; cat $O/bench_default_lock.json $O/bench_default_again.json | cut -c1-200
