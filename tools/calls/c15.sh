#!/bin/bash
O=gpurun_out/c15; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -12 > $O/pytest.txt
python tools/kbench.py --only gemm 2>&1 | grep "conv3 up" > $O/kb_new.txt
CID_CONV_H32=0 python tools/kbench.py --only gemm 2>&1 | grep "conv3 up" > $O/kb_old.txt
cat $O/pytest.txt; cat $O/kb_new.txt; echo OLD; cat $O/kb_old.txt
