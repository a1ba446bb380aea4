#!/bin/bash
# round-4 GPU call 3: two CFG lanes on two streams (A/B), N-loop on the 256x128 GEGLU tile (SDXL), parity of both
set -u
O=gpurun_out/r3; rm -rf $O; mkdir -p $O
python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/bench_lanes1.json 2>/dev/null
CID_CFG_LANES=2 python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/bench_lanes2.json 2>$O/lanes2.err
python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline > $O/bench_lanes1b.json 2>/dev/null
CID_CFG_LANES=2 python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline --batch-per-gpu 8 > $O/bench_b8_lanes2.json 2>/dev/null
python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline --batch-per-gpu 8 > $O/bench_b8_lanes1.json 2>/dev/null
cut -c1-140 $O/bench_lanes1.json $O/bench_lanes2.json $O/bench_lanes1b.json $O/bench_b8_lanes2.json $O/bench_b8_lanes1.json; tail -3 $O/lanes2.err
python bench.py --family sdxl --no-cpu-baseline --no-torch-baseline --no-roofline > $O/bench_sdxl_nloop.json 2>/dev/null
CID_GEGLU_NLOOP=1 python bench.py --family sdxl --no-cpu-baseline --no-torch-baseline --no-roofline > $O/bench_sdxl_nonloop.json 2>/dev/null
cut -c1-140 $O/bench_sdxl_nloop.json $O/bench_sdxl_nonloop.json
CID_CFG_LANES=2 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config2 or config5" -s > $O/pytest_lanes2.txt 2>&1
tail -6 $O/pytest_lanes2.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "geglu or layernorm_fold" > $O/pytest_geglu.txt 2>&1
tail -2 $O/pytest_geglu.txt
