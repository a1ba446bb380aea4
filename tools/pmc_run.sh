#!/bin/bash
# usage: tools/pmc_run.sh <kernel-tag> <outdir>   -- separate rocprofv3 passes per counter group
set -u
TAG=$1; OUT=$2; R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/$OUT/p$i -o r -- python $R/tools/pmc_one.py $TAG 3 > $R/$OUT/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "igemm" in k or "attn" in k or "conv_h32" in k or "geglu_h32" in k:
            agg[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
