#!/bin/bash
# round 6: the GEGLU projection on linear_h32.hip -- parity, kbench rows with the kernel on / off, headline A/B
set -u
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -k "geglu" -s 2>&1 | grep -a "parity\|passed\|failed\|Error\|error" | tail -14
for i in 1 2; do
  python tools/kbench.py --only gemm 2>/dev/null | grep -a "geglu" > $O/kb_h32_$i.txt
  CID_GEGLU_H32=0 python tools/kbench.py --only gemm 2>/dev/null | grep -a "geglu" > $O/kb_old_$i.txt
done
paste -d'|' $O/kb_h32_1.txt $O/kb_old_1.txt | cut -c1-75,120-200
paste -d'|' $O/kb_h32_2.txt $O/kb_old_2.txt | cut -c1-75,120-200
run() { local tag=$1; shift; env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2; do
run geglu-h32 X=1
run geglu-old CID_GEGLU_H32=0
done
cat $O/ab.txt
