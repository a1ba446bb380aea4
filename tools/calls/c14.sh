#!/bin/bash
O=gpurun_out/c14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_controlnet.py tests/test_gpu_fullsize.py -x -q -m gpu -k "controlnet or cn or inpaint or residual" 2>&1 | tail -4 > $O/pytest.txt
python bench.py --family cn-inpaint --steps 2 --warmup 1 --no-torch-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cn-inpaint b8', d['value'], d['ms_per_step'])" >> $O/pytest.txt
cat $O/pytest.txt
