#!/bin/bash
set -u
O=gpurun_out/r9; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/new /" >> $O/bench.txt
CID_LIBRARY=$PWD/consistentid_amd/libcid_oldends.so python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/old /" >> $O/bench.txt
done
cat $O/bench.txt
bash tools/profile_bench.sh $O/prof --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/prof.log 2>&1; rm -rf $O/prof/raw
grep "conv_in\|conv_out\|step_select" $O/prof/kernel_stats.csv | cut -c1-140
