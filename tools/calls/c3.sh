#!/bin/bash
set -u
O=gpurun_out/c3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3 or groupnorm_statistics or gemm_linear" 2>&1 | tail -5 > $O/pytest.txt
timeout 300 python -m pytest tests/test_gpu_properties.py -x -q -m gpu -k "split_k or deterministic" 2>&1 | tail -5 >> $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_controlnet.py -x -q -m gpu 2>&1 | tail -5 >> $O/pytest.txt
cat $O/pytest.txt
