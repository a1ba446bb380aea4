#!/bin/bash
set -u
O=gpurun_out/c8; rm -rf $O; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$O/raw -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $R/$O/bench_stdout.log 2>&1
cd $R
python tools/calls/seq.py $O > $O/seq.txt
rm -rf $O/raw
cat $O/seq.txt
