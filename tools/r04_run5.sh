#!/bin/bash
# round-4 GPU call 5: three-stage rings on the 256-token tile and in the halo convolution (A/B in one call) + parity
set -u
O=gpurun_out/r5; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -x -q -m gpu > $O/pytest_default.txt 2>&1; tail -2 $O/pytest_default.txt
CID_HALO_WRING=3 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -x -q -m gpu > $O/pytest_wr3.txt 2>&1; tail -2 $O/pytest_wr3.txt
b() { env "$@" python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/$* /" >> $O/bench.txt; }
b CID_GEMM_NBUF_A256=0 CID_HALO_WRING=2
b CID_GEMM_NBUF_A256=1 CID_HALO_WRING=2
b CID_GEMM_NBUF_A256=1 CID_HALO_WRING=3
b CID_GEMM_NBUF_A256=0 CID_HALO_WRING=3
b CID_GEMM_NBUF_A256=0 CID_HALO_WRING=2
b CID_GEMM_NBUF_A256=1 CID_HALO_WRING=2
b CID_GEMM_NBUF_A256=1 CID_HALO_WRING=3
cat $O/bench.txt
for v in "CID_HALO_WRING=2" "CID_HALO_WRING=3"; do echo "== $v" >> $O/kb.txt; env $v timeout 300 python tools/kbench.py --only gemm 2>&1 | grep "conv3" | grep -v gn-stats >> $O/kb.txt; done
cat $O/kb.txt
s() { env "$@" python bench.py --family sdxl --no-cpu-baseline --no-torch-baseline --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/sdxl $* /" >> $O/bench_sdxl.txt; }
s CID_GEMM_NBUF_A256=0 CID_HALO_WRING=2
s CID_GEMM_NBUF_A256=1 CID_HALO_WRING=2
s CID_GEMM_NBUF_A256=1 CID_HALO_WRING=3
cat $O/bench_sdxl.txt
