#!/bin/bash
# round 6: 2-D XCD tile partition (gemm_args.h xcd_tile / choose_xcd_pn) on / off -- parity, fabric read traffic (FETCH_SIZE),
# kbench rows, headline + SDXL A/B
set -u
O=gpurun_out/r06x; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -k "gemm" 2>&1 | tail -2
R=$PWD
cd /tmp && export TMPDIR=/tmp
for t in conv2 conv3 geglu2 lin2; do for v in 1 0; do
  CID_XCD_2D=$v timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/f_${t}_$v -o r -- python $R/tools/pmc_one.py $t 3 > /dev/null 2>&1
  python - <<PY >> $R/$O/fetch.txt
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$R/$O/f_${t}_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("igemm", "conv_h32", "geglu_h32", "splitk")):
            d[k[:60]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print("%-7s XCD_2D=$v  %-62s FETCH_SIZE %9.1f KB (x2 on gfx950 = %6.1f MB read)" % ("$t", k, sum(v) / len(v), 2 * sum(v) / len(v) / 1024))
PY
  rm -rf $R/$O/f_${t}_$v
done; done
cd $R
cat $O/fetch.txt
python tools/kbench.py --only gemm 2>/dev/null > $O/kb_2d.txt
CID_XCD_2D=0 python tools/kbench.py --only gemm 2>/dev/null > $O/kb_lin.txt
paste -d'|' $O/kb_2d.txt $O/kb_lin.txt | cut -c1-75,120-180
run() { local tag=$1; shift; env "$@" timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline $FAM 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2; do
FAM=""; run sd15-xcd2d X=1; run sd15-linear CID_XCD_2D=0
FAM="--family sdxl"; run sdxl-xcd2d X=1; run sdxl-linear CID_XCD_2D=0
done
cat $O/ab.txt
