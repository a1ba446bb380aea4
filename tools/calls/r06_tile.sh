#!/bin/bash
# round 6: 128-token tiles for every linear that has >= 256 of them (plan_gemm, CID_GEMM_PREFER128) -- kbench rows and step A/B
set -u
O=gpurun_out/r06t; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -k "gemm" 2>&1 | tail -1
for v in 1 0; do
  echo "== CID_GEMM_PREFER128=$v  SD1.5 CFG batch 8" >> $O/kb.txt
  CID_GEMM_PREFER128=$v python tools/kbench.py --only gemm 2>/dev/null | grep -a "^lin " >> $O/kb.txt
  echo "== CID_GEMM_PREFER128=$v  SD1.5 CFG batch 16" >> $O/kb.txt
  CID_GEMM_PREFER128=$v python tools/kbench.py --only gemm --b2 16 2>/dev/null | grep -a "^lin " >> $O/kb.txt
  echo "== CID_GEMM_PREFER128=$v  SDXL CFG batch 4" >> $O/kb.txt
  CID_GEMM_PREFER128=$v python tools/kbench.py --only gemm --family sdxl --b2 4 2>/dev/null | grep -a "^lin " >> $O/kb.txt
done
cat $O/kb.txt
run() { local tag=$1; shift; env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline $FAM 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-26s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2; do
FAM=""; run sd15-prefer128 X=1; run sd15-old-rule CID_GEMM_PREFER128=0
FAM="--batch-per-gpu 8"; run sd15-b8-prefer128 X=1; run sd15-b8-old-rule CID_GEMM_PREFER128=0
FAM="--family sdxl"; run sdxl-prefer128 X=1; run sdxl-old-rule CID_GEMM_PREFER128=0
done
cat $O/ab.txt
