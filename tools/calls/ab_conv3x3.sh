#!/bin/bash
set -u
O=gpurun_out/c10; rm -rf $O; mkdir -p $O
run() { local tag=$1; shift; env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2 3; do
run h32-all X=1
run h32-256only CID_CONV_H32=2
run old CID_CONV_H32=0
done
cat $O/ab.txt
