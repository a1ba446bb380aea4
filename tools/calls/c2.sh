#!/bin/bash
set -u
O=gpurun_out/c2; mkdir -p $O
timeout 300 ./tools/probes/conv_loop > $O/conv_loop.txt 2>&1




cat $O/conv_loop.txt
