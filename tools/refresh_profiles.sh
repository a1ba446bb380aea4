#!/bin/bash
# One GPU call that regenerates every measured artefact kept under profiles/ (run through gpurun, then
# `python tools/collect_profiles.py gpurun_out/final r04` copies the summaries into profiles/ with their stamps).
# usage: tools/refresh_profiles.sh [core|extra|all]   (two shorter GPU calls instead of one long one: core, then extra)
# Variant libraries used by the extra stage (build them before the call):
#   python -m consistentid_amd.build --variant trace CID_X3_TRACE
#   for b in 1 16 6 7; do python -m consistentid_amd.build --variant attnabl$b CID_ATTN_ABL=$b; done
set -u
STAGE=${1:-all}
O=gpurun_out/final
if [ "$STAGE" != extra ]; then
rm -rf $O; mkdir -p $O
# PMC passes first: bench.py reports roofline.traffic only from a summary taken on the very kernel sources it runs
bash tools/pmc_run.sh xattn3 $O/pmc_xattn > $O/pmc_xattn.txt 2>&1
rm -rf $O/pmc_xattn/p*/
python tools/collect_profiles.py $O r04 --pmc-only
bash tools/profile_bench.sh $O/prof --no-cpu-baseline --no-torch-baseline --no-secondary > $O/prof.log 2>&1
rm -rf $O/prof/raw
python tools/collect_profiles.py $O r04 --stats-only
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/kbench.py > $O/kbench.txt 2>&1
python tools/xattn_levels.py 2>&1 | grep -v amdgpu.ids > $O/xattn_levels.txt
# same-box A/B of the round's structural changes: GEGLU N-loop, three-stage DMA ring, query projection with attention epilogue
ab() { env "$@" python bench.py --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-52s %.4f images/s  %.2f ms/generation' % ('$*', d['value'], d['ms_per_step']))" >> $O/ab.txt; }
for i in 1 2; do ab X=shipped; ab CID_GEGLU_NLOOP=1; ab CID_GEMM_NBUF=2; ab CID_QATTN=0; ab CID_GEGLU_NLOOP=1 CID_GEMM_NBUF=2 CID_QATTN=0; done
tail -1 $O/bench_default.json | cut -c1-600; cat $O/ab.txt
fi
[ "$STAGE" = core ] && exit 0
mkdir -p $O
./tools/probes/issue_rates > $O/issue_rates.txt 2>&1
bash tools/pmc_run.sh attn0 $O/pmc_attn > $O/pmc_selfattn.txt 2>&1
rm -rf $O/pmc_attn
for v in "" attnabl1 attnabl16 attnabl6 attnabl7; do
  echo "== ${v:-shipped}" >> $O/attn_ablation.txt
  if [ -z "$v" ]; then python tools/kbench.py --only attn 2>&1 | grep "self-attn" >> $O/attn_ablation.txt
  else CID_LIBRARY=$PWD/consistentid_amd/libcid_$v.so python tools/kbench.py --only attn 2>&1 | grep "self-attn L0" >> $O/attn_ablation.txt; fi
done
bash tools/pmc_run.sh conv0 $O/pmc_conv0 > $O/pmc_conv0.txt 2>&1
rm -rf $O/pmc_conv0/
CID_LIBRARY=consistentid_amd/libcid_trace.so python tools/x2_trace.py --gen 3 > $O/xattn_trace.txt 2>&1
bash tools/profile_bench.sh $O/prof_sdxl --family sdxl --no-cpu-baseline --no-torch-baseline --no-roofline > $O/prof_sdxl.log 2>&1
rm -rf $O/prof_sdxl/raw
tail -5 $O/attn_ablation.txt
# the full GPU suite takes ten minutes: SKIP_PYTEST=1 leaves it to a call of its own
[ -n "${SKIP_PYTEST:-}" ] || { timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt; }
