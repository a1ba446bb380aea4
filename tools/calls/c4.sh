#!/bin/bash
# step-level A/B: shipped (conv3x3.hip on) vs CID_CONV_H32=0, plus the unet / fullsize parity tests that touch the convs
set -u
O=gpurun_out/c4; mkdir -p $O
run() { local tag=$1; shift; env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
run h32 X=1
run old CID_CONV_H32=0
run h32 X=1
run old CID_CONV_H32=0
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest.txt
cat $O/ab.txt $O/pytest.txt
