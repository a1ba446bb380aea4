#!/bin/bash
# round 6, first call: kernel + UNet tests after the conv_in / conv_out / step_select / out2 changes, the two new full-size parity
# cases, and a short headline bench
set -u
O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -x -q -m gpu -p no:cacheprovider > $O/pytest_kernels.txt 2>&1
tail -3 $O/pytest_kernels.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -p no:cacheprovider -k "config3 or sdxl_reference_default or sd15_unet_forward_full_size" > $O/pytest_new.txt 2>&1
grep -a "^\[parity\]\|^\[drift\]\|passed\|failed\|Error" $O/pytest_new.txt | tail -12
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>$O/bench.err | tail -1 > $O/bench_$i.json
python -c "import json; d=json.load(open('$O/bench_$i.json')); print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms')"
done
