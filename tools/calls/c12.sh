#!/bin/bash
O=gpurun_out/c12; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'])
print('gpu_state', d.get('gpu_state'))
r = d['roofline']
print('roofline frac', r['frac'], 'avg us', r['avg_launch_us'], 'frac_of_sustained', r.get('frac_of_sustained'), r.get('sustained'))
print('roofline gpu_state', r.get('gpu_state'))
print('step', r.get('step'))
print('secondary', {k: v['images_per_s'] for k, v in d.get('secondary', {}).items()})
print('torch', d.get('torch_fp16_baseline', {}).get('value'), 'cpu', d.get('cpu_baseline', {}).get('value'))
"
tail -3 $O/bench_default.err
