#!/bin/bash
# round 6, second call: the cheaper erf GELU (same-call A/B against the round-5 form: experiment build gelu5), its parity tests,
# and the issue-rate probe extended by one MFMA wave beside 1 / 2 / 3 VALU waves
set -u
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
./tools/probes/issue_rates > $O/issue_rates.txt 2>&1; tail -12 $O/issue_rates.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -k "geglu or layernorm_fold" -s 2>&1 | grep -a "parity\|passed\|failed" | tail -12
for i in 1 2; do
  python tools/kbench.py --only gemm 2>/dev/null | grep -a "geglu" > $O/kb_new_$i.txt
  CID_LIBRARY=$PWD/consistentid_amd/libcid_gelu5.so python tools/kbench.py --only gemm 2>/dev/null | grep -a "geglu" > $O/kb_old_$i.txt
done
paste -d'|' $O/kb_new_1.txt $O/kb_old_1.txt | cut -c1-75,120-200
run() { local tag=$1; shift; env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2; do
run new X=1
run old-gelu CID_LIBRARY=$PWD/consistentid_amd/libcid_gelu5.so
done
cat $O/ab.txt
