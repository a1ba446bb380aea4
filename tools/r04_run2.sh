#!/bin/bash
# round-4 GPU call 2: N-loop GEGLU + self-attention max hand-off: parity, kernel A/B, step A/B
set -u
O=gpurun_out/r2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
echo "== attention: shipped" > $O/ab.txt
timeout 120 python tools/kbench.py --only attn 2>&1 | grep "self-attn" >> $O/ab.txt
echo "== attention: max at the head of the step (round 3 form)" >> $O/ab.txt
CID_LIBRARY=$PWD/consistentid_amd/libcid_attnhead.so timeout 120 python tools/kbench.py --only attn 2>&1 | grep "self-attn" >> $O/ab.txt
echo "== gemm: shipped (N-loop GEGLU)" >> $O/ab.txt
timeout 200 python tools/kbench.py --only gemm 2>&1 | grep -i "geglu\|lin L0" >> $O/ab.txt
echo "== gemm: CID_GEGLU_NLOOP=1 (one tile per workgroup)" >> $O/ab.txt
CID_GEGLU_NLOOP=1 timeout 200 python tools/kbench.py --only gemm 2>&1 | grep -i "geglu" >> $O/ab.txt
for nl in 2 4 5 20; do echo "== gemm: CID_GEGLU_NLOOP=$nl" >> $O/ab.txt; CID_GEGLU_NLOOP=$nl timeout 200 python tools/kbench.py --only gemm 2>&1 | grep -i "geglu L" | grep -v fold >> $O/ab.txt; done
cat $O/ab.txt
python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary > $O/bench_new.json 2>/dev/null
CID_GEGLU_NLOOP=1 CID_LIBRARY=$PWD/consistentid_amd/libcid_attnhead.so python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/bench_old.json 2>/dev/null
python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/bench_new2.json 2>/dev/null
cut -c1-160 $O/bench_new.json $O/bench_old.json $O/bench_new2.json
