#!/bin/bash
O=gpurun_out/c18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_properties.py -q -m gpu 2>&1 | tail -3 > $O/props.txt
CID_BENCH_SHARE_GPU=1 CID_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 > $O/bench_gpus2.json 2> $O/bench_gpus2.err
cat $O/props.txt; tail -1 $O/bench_gpus2.json | cut -c1-1500; tail -3 $O/bench_gpus2.err
