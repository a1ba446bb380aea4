#!/bin/bash
set -u
O=gpurun_out/r10; rm -rf $O; mkdir -p $O
for i in 1 2; do for g in 1 5 10 25; do CID_GRAPH_STEPS=$g python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/graph_steps=$g /" >> $O/bench.txt; done; done
cat $O/bench.txt
CID_GRAPH_STEPS=5 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config2 or config5" -s 2>&1 | grep "parity\|passed\|failed" | cut -c1-200
