#!/bin/bash
# round 6: linear_h32.hip routed for deep K only -- kbench rows (SD1.5 + SDXL) with the kernel on / off, headline + SDXL A/B
set -u
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -k "geglu" 2>&1 | tail -2
python tools/kbench.py --only gemm 2>/dev/null | grep -a "geglu" > $O/kb_h32.txt
CID_GEGLU_H32=0 python tools/kbench.py --only gemm 2>/dev/null | grep -a "geglu" > $O/kb_old.txt
python tools/kbench.py --only gemm --family sdxl --b2 4 2>/dev/null | grep -a "geglu" >> $O/kb_h32.txt
CID_GEGLU_H32=0 python tools/kbench.py --only gemm --family sdxl --b2 4 2>/dev/null | grep -a "geglu" >> $O/kb_old.txt
paste -d'|' $O/kb_h32.txt $O/kb_old.txt | cut -c1-75,120-200
run() { local tag=$1; shift; env "$@" timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline $FAM 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2; do
FAM=""; run sd15-geglu-h32 X=1; run sd15-geglu-old CID_GEGLU_H32=0
FAM="--family sdxl"; run sdxl-geglu-h32 X=1; run sdxl-geglu-old CID_GEGLU_H32=0
done
cat $O/ab.txt
