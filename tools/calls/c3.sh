#!/bin/bash
set -u
O=gpurun_out/c3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3 or groupnorm_statistics or gemm_linear" 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python -m pytest tests/test_gpu_properties.py -x -q -m gpu -k "split_k or deterministic" 2>&1 | tail -8 >> $O/pytest.txt
timeout 300 python tools/kbench.py --only gemm 2>&1 | grep -v amdgpu.ids > $O/kbench.txt
CID_CONV_H32=0 timeout 300 python tools/kbench.py --only gemm 2>&1 | grep "conv3" > $O/kbench_old.txt
cat $O/pytest.txt; grep conv3 $O/kbench.txt; echo OLD; cat $O/kbench_old.txt
