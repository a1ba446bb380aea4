#!/bin/bash
# One GPU call that regenerates every measured artefact kept under profiles/ (run through gpurun, then
# `python tools/collect_profiles.py gpurun_out/final r03` copies the summaries into profiles/ with their stamps).
set -u
O=gpurun_out/final
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
CID_LN_FOLD=0 python bench.py --family sdxl --no-cpu-baseline --no-torch-baseline --no-roofline > $O/bench_sdxl_nolnfold.json 2>/dev/null
python bench.py --family cn-inpaint > $O/bench_cn_inpaint.json 2>/dev/null
python bench.py --batch-per-gpu 8 --no-cpu-baseline --no-torch-baseline > $O/bench_sd15_batch8.json 2>/dev/null
CID_LIBRARY=consistentid_amd/libcid_trace.so python tools/x2_trace.py --gen 3 > $O/xattn_trace.txt 2>&1
tail -1 $O/bench_default.json | cut -c1-600
