#!/bin/bash
# round-4 GPU call 1: instruction issue rates, self-attention ablations + counters, the round's pipeline changes, the new bench line
set -u
O=gpurun_out/r1; rm -rf $O; mkdir -p $O
timeout 120 ./tools/probes/issue_rates > $O/issue_rates.txt 2>&1
for v in "" attnabl1 attnabl16 attnabl17 attnabl6 attnabl7; do
  echo "== $v" >> $O/attn_abl.txt
  if [ -z "$v" ]; then timeout 120 python tools/kbench.py --only attn 2>&1 | grep "self-attn" >> $O/attn_abl.txt
  else CID_LIBRARY=$PWD/consistentid_amd/libcid_$v.so timeout 120 python tools/kbench.py --only attn 2>&1 | grep "self-attn L0" >> $O/attn_abl.txt; fi
done
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_controlnet.py tests/test_gpu_kernels.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json
bash tools/pmc_run.sh attn0 $O/pmc_attn > $O/pmc_attn.txt 2>&1
rm -rf $O/pmc_attn
cat $O/attn_abl.txt
