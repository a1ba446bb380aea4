#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command; writes a compact per-kernel summary.
# usage: tools/profile_bench.sh <outdir> [bench args...]
set -u
OUT=$1; shift
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/raw -o bench -- python $R/bench.py "$@" > $R/$OUT/bench_stdout.log 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(lambda: [0, 0.0])
shapes = collections.defaultdict(lambda: [0, 0.0])      # (kernel, grid, LDS) -> launches of ONE shape inside the step
durs = collections.defaultdict(list)
for f in glob.glob(f"{out}/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = r["Kernel_Name"]
        rows[k][0] += 1; rows[k][1] += d
        g = "x".join(str(r.get(c, "?")) for c in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        sk = (k, g, r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")))
        shapes[sk][0] += 1; shapes[sk][1] += d
        durs[sk].append(d)
tot = sum(v[1] for v in rows.values())
with open(f"{out}/kernel_stats.csv", "w") as fo:
    fo.write("kernel,calls,total_us,avg_us,pct\n")
    for k, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        name = k.replace(",", ";")
        if len(name) > 140: name = name[:140] + "..."
        fo.write(f"\"{name}\",{n},{t:.1f},{t/n:.2f},{100*t/tot:.2f}\n")
with open(f"{out}/kernel_shapes.csv", "w") as fo:
    fo.write("kernel,grid_threads,lds_bytes,calls,total_us,avg_us,pct,p10_us,median_us,p90_us\n")
    for (k, g, l), (n, t) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
        name = k.replace(",", ";").replace("(anonymous namespace)::", "")
        if len(name) > 90: name = name[:90] + "..."
        ds = sorted(durs[(k, g, l)])
        if t / tot > 0.002: fo.write(f"\"{name}\",{g},{l},{n},{t:.1f},{t/n:.2f},{100*t/tot:.2f},{ds[len(ds)//10]:.2f},{ds[len(ds)//2]:.2f},{ds[(9*len(ds))//10]:.2f}\n")
print(open(f"{out}/kernel_stats.csv").read()[:3000])
PY
tail -2 $OUT/bench_stdout.log | cut -c1-400
