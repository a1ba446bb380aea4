#!/bin/bash
O=gpurun_out/c11; mkdir -p $O
for i in 1 2; do
python tools/kbench.py --only norm 2>&1 | grep "stats\|layernorm" > $O/new$i.txt
CID_LIBRARY=$PWD/consistentid_amd/libcid_base.so python tools/kbench.py --only norm 2>&1 | grep "stats\|layernorm" > $O/old$i.txt
done
paste $O/new1.txt $O/old1.txt | cut -c1-75,120-200; paste $O/new2.txt $O/old2.txt | cut -c1-75,120-200
