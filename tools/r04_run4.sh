#!/bin/bash
# round-4 GPU call 4: three-stage DMA ring for the latency-bound 128- / 64-token tiles (A/B) + parity
set -u
O=gpurun_out/r4; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -x -q -m gpu > $O/pytest_default.txt 2>&1; tail -2 $O/pytest_default.txt
CID_GEMM_NBUF=3 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -x -q -m gpu > $O/pytest_nbuf3.txt 2>&1; tail -2 $O/pytest_nbuf3.txt
for nb in 2 0 3; do echo "== CID_GEMM_NBUF=$nb" >> $O/kb.txt; CID_GEMM_NBUF=$nb timeout 300 python tools/kbench.py --only gemm 2>&1 | grep -v "gn-stats\|amdgpu" >> $O/kb.txt; done
cat $O/kb.txt | grep -v "^kernel\|conv3 L0\|conv3 L1 \|conv3 L1up\|conv3 L2 \|sum over"
for nb in 2 0 3 2 0; do CID_GEMM_NBUF=$nb python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-120 | sed "s/^/nbuf=$nb /" >> $O/bench.txt; done
cat $O/bench.txt
for nb in 2 0 3; do CID_GEMM_NBUF=$nb python bench.py --family sdxl --no-cpu-baseline --no-torch-baseline --no-roofline 2>/dev/null | cut -c1-120 | sed "s/^/sdxl nbuf=$nb /" >> $O/bench_sdxl.txt; done
cat $O/bench_sdxl.txt
