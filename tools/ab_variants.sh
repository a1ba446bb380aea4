#!/bin/bash
# Same-box A/B of variant libraries against the shipped one: the denoise loop (two alternating passes) and, with KBENCH=<groups>,
# the kbench rows that differ by more than 2 %.  Build the variants first (python -m consistentid_amd.build --variant NAME ...).
# usage (inside one gpurun call): [KBENCH=gemm] tools/ab_variants.sh <outdir> NAME [NAME ...]
set -u
O=$1; shift
mkdir -p $O
run() { local tag=$1; shift; env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for pass in 1 2; do
  run shipped X=1
  for v in "$@"; do run $v CID_LIBRARY=$PWD/consistentid_amd/libcid_$v.so; done
done
if [ -n "${KBENCH:-}" ]; then
  timeout 300 python tools/kbench.py --only $KBENCH > $O/kbench_shipped.txt 2>&1
  for v in "$@"; do
    CID_LIBRARY=$PWD/consistentid_amd/libcid_$v.so timeout 300 python tools/kbench.py --only $KBENCH > $O/kbench_$v.txt 2>&1
    python - $O/kbench_shipped.txt $O/kbench_$v.txt $v >> $O/ab.txt <<'PY'
import re, sys
def rows(p):
    out = {}
    for l in open(p):
        m = re.match(r"(.{40}) (\S+)\s+([0-9.]+)\s", l)
        if m: out[m.group(1).strip()] = float(m.group(3))
    return out
a, b = rows(sys.argv[1]), rows(sys.argv[2])
print(f"-- kbench rows of {sys.argv[3]} that differ from shipped by more than 2 % (us, shipped -> variant)")
for k, v in a.items():
    if k in b and abs(b[k] / v - 1) > 0.02: print(f"   {k:40s} {v:8.1f} -> {b[k]:8.1f}  ({b[k] / v - 1:+.1%})")
PY
  done
fi
cat $O/ab.txt
