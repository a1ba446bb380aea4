#!/bin/bash
O=gpurun_out/c6; mkdir -p $O
python tools/calls/convvar.py > $O/new.txt 2>&1
CID_CONV_H32=0 python tools/calls/convvar.py > $O/old.txt 2>&1
paste $O/new.txt $O/old.txt
