#!/bin/bash
O=gpurun_out/c9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3 or groupnorm_statistics" 2>&1 | tail -3 > $O/pytest.txt
timeout 300 python -m pytest tests/test_gpu_properties.py -x -q -m gpu -k "split_k or deterministic" 2>&1 | tail -3 >> $O/pytest.txt
for v in "X=1" "CID_CONV_STAGES=3" "CID_CONV_STAGES=4" "CID_CONV_H32=0"; do echo "== $v"; env $v python tools/calls/convcold.py 2>&1 | grep -v amdgpu; done > $O/cold.txt
cat $O/pytest.txt $O/cold.txt
