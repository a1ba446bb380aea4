#!/bin/bash
# round-4 GPU call 6: GroupNorm apply with the activations in flight across the fold, LayerNorm with one round trip (A/B vs the previous library)
set -u
O=gpurun_out/r6; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "norm or groupnorm or layernorm" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
echo "== new" > $O/kb.txt; timeout 200 python tools/kbench.py --only norm 2>&1 | grep -i "norm" >> $O/kb.txt
echo "== old" >> $O/kb.txt; CID_LIBRARY=$PWD/consistentid_amd/libcid_oldnorm.so timeout 200 python tools/kbench.py --only norm 2>&1 | grep -i "norm" >> $O/kb.txt
cat $O/kb.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/new /" >> $O/bench.txt
CID_LIBRARY=$PWD/consistentid_amd/libcid_oldnorm.so python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | cut -c1-110 | sed "s/^/old /" >> $O/bench.txt
done
cat $O/bench.txt
