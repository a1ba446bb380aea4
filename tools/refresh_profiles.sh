#!/bin/bash
# One GPU call that regenerates every measured artefact kept under profiles/ (run through gpurun, then copy).
set -u
O=gpurun_out/final
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/profile_bench.sh $O/prof --no-cpu-baseline > $O/prof.log 2>&1
rm -rf $O/prof/raw
python tools/kbench.py > $O/kbench.txt 2>&1
python bench.py --family sdxl --no-cpu-baseline > $O/bench_sdxl.json 2>/dev/null
python bench.py --family cn-inpaint --no-roofline > $O/bench_cn_inpaint.json 2>/dev/null
tail -1 $O/bench_default.json | cut -c1-400
