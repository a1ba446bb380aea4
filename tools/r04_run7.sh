#!/bin/bash
# round-4 GPU call 7: halo weight ring of three stages only where few token tiles share a weight slab (M <= 2048)
set -u
O=gpurun_out/r7; rm -rf $O; mkdir -p $O
CID_HALO_WRING=3 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -x -q -m gpu -k "conv or gemm or determin" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for i in 1 2 3; do for v in 2 0; do CID_HALO_WRING=$v python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/wring=$v /" >> $O/bench.txt; done; done
cat $O/bench.txt
