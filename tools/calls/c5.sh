#!/bin/bash
set -u
O=gpurun_out/c5; rm -rf $O; mkdir -p $O
bash tools/profile_bench.sh $O/prof --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/prof.log 2>&1
rm -rf $O/prof/raw
CID_CONV_H32=0 bash tools/profile_bench.sh $O/prof_old --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/prof_old.log 2>&1
rm -rf $O/prof_old/raw
head -30 $O/prof/kernel_stats.csv | cut -c1-150
