#!/bin/bash
set -u
O=gpurun_out/r12; rm -rf $O; mkdir -p $O
for i in 1 2; do for q in 1 0; do CID_QATTN=$q python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/qattn=$q /" >> $O/bench.txt; done; done
cat $O/bench.txt
for q in 1 0; do CID_QATTN=$q python bench.py --family sdxl --no-cpu-baseline --no-torch-baseline --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/sdxl qattn=$q /" >> $O/bench_sdxl.txt; done
cat $O/bench_sdxl.txt
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_controlnet.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
