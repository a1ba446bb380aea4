#!/bin/bash
O=gpurun_out/c13; mkdir -p $O
./tools/probes/conv_loop long > $O/long.txt 2>&1 &
PID=$!
sleep 1
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --json 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['card0']
print({k: v for k, v in c.items() if 'sclk clock speed' in k or 'Power' in k})"; done > $O/smi.txt 2>&1
wait $PID
cat $O/long.txt | head -8; cat $O/smi.txt
