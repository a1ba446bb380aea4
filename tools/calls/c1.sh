#!/bin/bash
# round 5, GPU call 1: conv-loop prototype, extended issue-rate probe, -fno-slp-vectorize A/B (xattn3 / attn / gemm)
set -u
O=gpurun_out/c1; mkdir -p $O
timeout 300 ./tools/probes/conv_loop > $O/conv_loop.txt 2>&1
timeout 120 ./tools/probes/issue_rates > $O/issue_rates.txt 2>&1
run() { local tag=$1; shift; env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
run shipped X=1
for v in nsx3 nsattn nsgemm; do run $v CID_LIBRARY=$PWD/consistentid_amd/libcid_$v.so; done
run shipped X=1
timeout 300 python tools/kbench.py --only gemm,attn,xattn > $O/kbench_shipped.txt 2>&1
CID_LIBRARY=$PWD/consistentid_amd/libcid_nsx3.so timeout 120 python tools/kbench.py --only xattn > $O/kbench_nsx3.txt 2>&1
CID_LIBRARY=$PWD/consistentid_amd/libcid_nsattn.so timeout 120 python tools/kbench.py --only attn > $O/kbench_nsattn.txt 2>&1
CID_LIBRARY=$PWD/consistentid_amd/libcid_nsgemm.so timeout 300 python tools/kbench.py --only gemm > $O/kbench_nsgemm.txt 2>&1
cat $O/conv_loop.txt; cat $O/ab.txt; grep -h "xattn3\|self-attn L0" $O/kbench_*.txt
