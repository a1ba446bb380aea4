#!/bin/bash
set -u
O=gpurun_out/r13; rm -rf $O; mkdir -p $O
for m in 2048 8192 16384; do echo "== CID_QATTN_LNFOLD_MAX=$m" >> $O/levels.txt; CID_QATTN_LNFOLD_MAX=$m timeout 300 python tools/xattn_levels.py 2>&1 | grep -v "amdgpu\|family" | cut -c1-125 >> $O/levels.txt; done
cat $O/levels.txt
