#!/bin/bash
# GPU suite file by file inside a time budget (the whole suite takes ten minutes on one MI355X): each file's summary line is
# kept, files that did not fit are listed as NOT RUN.   usage: tools/pytest_gpu_budget.sh <seconds> <outfile>
set -u
BUDGET=$1; OUT=$2; T0=$(date +%s)
mkdir -p $(dirname $OUT); : > $OUT
for f in tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_properties.py tests/test_gpu_controlnet.py \
         tests/test_gpu_vae.py tests/test_gpu_idstack.py tests/test_gpu_fullsize.py; do
  left=$(( BUDGET - ($(date +%s) - T0) ))
  if [ $left -lt 30 ]; then echo "$f: NOT RUN (budget of $BUDGET s used up)" >> $OUT; continue; fi
  res=$(timeout $left python -m pytest $f -q -m gpu 2>&1 | tail -1)
  [ -z "$res" ] && res="cut off by the time budget"
  echo "$f: $res" >> $OUT
done
cat $OUT
