#!/bin/bash
# in-step kernel table of the SDXL shard (config 4's per-GPU shape), without the roofline block
O=gpurun_out/final; mkdir -p $O; rm -rf $O/prof_sdxl
bash tools/profile_bench.sh $O/prof_sdxl --family sdxl --steps 2 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-roofline > $O/prof_sdxl.log 2>&1
rm -rf $O/prof_sdxl/raw
head -16 $O/prof_sdxl/kernel_stats.csv | cut -c1-150
