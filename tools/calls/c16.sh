#!/bin/bash
set -u
O=gpurun_out/c16; rm -rf $O; mkdir -p $O
run() { local tag=$1; shift; env "$@" timeout 400 python bench.py --family sdxl --steps 2 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2; do
run sdxl-h32 X=1
run sdxl-old CID_CONV_H32=0
done
cat $O/ab.txt
