#!/bin/bash
O=gpurun_out/r06e; mkdir -p $O
for v in "" g32nodrain g32nox g32both; do
  if [ -z "$v" ]; then L=""; else L="CID_LIBRARY=$PWD/consistentid_amd/libcid_$v.so"; fi
  echo "== ${v:-product}" >> $O/abl.txt
  env $L python tools/kbench.py --only gemm 2>/dev/null | grep -a "geglu L[012] " >> $O/abl.txt
done
cat $O/abl.txt
