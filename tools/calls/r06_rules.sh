#!/bin/bash
# round 6: the launch rules of rounds 3-5 re-checked at the step level after the tile rule changed (each line = one bench run,
# alternating; the shipped setting first)
set -u
O=gpurun_out/r06r; rm -rf $O; mkdir -p $O
run() { local fam=$1; shift; env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline $fam 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-16s %-34s %.4f images/s  %.2f ms/generation' % ('$fam', '$*', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2; do
for fam in "" "--family sdxl"; do
  run "$fam" X=shipped
  run "$fam" CID_GEMM_NBUF=2
  run "$fam" CID_GEMM_NBUF=3
  run "$fam" CID_LN_FOLD=0
  run "$fam" CID_LN_FOLD=1
  run "$fam" CID_QATTN_LNFOLD_MAX=0
  run "$fam" CID_QATTN_LNFOLD_MAX=32768
  run "$fam" CID_GEGLU_FOLD_MAX=32768
  run "$fam" CID_GN_EPILOGUE_STATS=0
done; done
cat $O/ab.txt
