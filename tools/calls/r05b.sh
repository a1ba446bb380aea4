#!/bin/bash
# partial re-take: the in-step kernel table without the roofline block's back-to-back loop, then the default bench line that reads it
set -u
O=gpurun_out/final
mkdir -p $O; rm -rf $O/prof
bash tools/profile_bench.sh $O/prof --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/prof.log 2>&1
rm -rf $O/prof/raw
python tools/collect_profiles.py $O r05 --stats-only
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-300
