#!/bin/bash
O=gpurun_out/c17; mkdir -p $O
for st in 0 2 4 6 8 10 12 16 20 0; do
  echo -n "stagger $st x 512 cycles: " >> $O/stagger.txt
  CID_X3_STAGGER=$st python tools/kbench.py --only xattn3 2>&1 | grep "xattn3" >> $O/stagger.txt
done
cat $O/stagger.txt
