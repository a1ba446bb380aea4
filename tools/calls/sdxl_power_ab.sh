#!/bin/bash
O=gpurun_out/r06h; mkdir -p $O
for v in 1 0 1 0; do
CID_GEGLU_H32=$v python bench.py --family sdxl --steps 2 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | tail -1 > $O/sdxl_$v.json
python -c "
import json; d=json.load(open('$O/sdxl_$v.json')); g=d.get('gpu_state') or {}
print('GEGLU_H32=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', {k: g[k] for k in ('sclk_mhz','power_w','temp_c') if k in g})"
done
