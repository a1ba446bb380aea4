#!/bin/bash
set -u
O=gpurun_out/c7; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3 or groupnorm_statistics" 2>&1 | tail -3 > $O/pytest.txt
timeout 300 python -m pytest tests/test_gpu_properties.py -x -q -m gpu -k "split_k or deterministic" 2>&1 | tail -3 >> $O/pytest.txt
bash tools/profile_bench.sh $O/prof --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/prof.log 2>&1
rm -rf $O/prof/raw
run() { local tag=$1; shift; env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
run h32 X=1
run old CID_CONV_H32=0
run h32 X=1
run old CID_CONV_H32=0
cat $O/pytest.txt $O/ab.txt; grep "conv_h32\|halo" $O/prof/kernel_shapes.csv
