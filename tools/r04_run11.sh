#!/bin/bash
set -u
O=gpurun_out/r11; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "id_cross_attention" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 300 python tools/xattn_levels.py 2>&1 | grep -v amdgpu > $O/levels.txt; cat $O/levels.txt
