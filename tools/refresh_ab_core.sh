#!/bin/bash
# One short GPU call for a codegen-only library change: same-box A/B of the previous library against the rebuilt one, then
# the evidence bench.py and the docs read (in-step kernel table, default bench line, kbench), most important first so that
# a call cut short still leaves a consistent prefix.  Before the call:
#   cp consistentid_amd/libcid.so consistentid_amd/libcid_prev.so      (at the previous commit's sources)
#   python -m consistentid_amd.build                                    (the new sources)
# afterwards: python tools/collect_profiles.py gpurun_out/abcore r04
set -u
O=gpurun_out/abcore
rm -rf $O; mkdir -p $O
T0=$(date +%s); t() { echo "[$(( $(date +%s) - T0 )) s] $*" >> $O/timeline.txt; }
ab() { local tag=$1; shift; env "$@" timeout 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-secondary --no-roofline 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.4f images/s  %.2f ms/generation' % ('$tag', d['value'], d['ms_per_step']))" >> $O/ab.txt; t "ab $tag"; }
PREV=$PWD/consistentid_amd/libcid_prev.so
# parity of the rebuilt library on the kernels it touches (3x3 convolutions, GEMM tiles, cross-attention epilogue GEMMs)
timeout 240 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "test_gemm_conv3x3 or test_gemm_linear or test_gemm_layernorm_fold or (test_id_cross_attention and not _v)" > $O/pytest_kernels.txt 2>&1; tail -1 $O/pytest_kernels.txt; t "pytest kernels"
ab previous CID_LIBRARY=$PREV; ab rebuilt X=1; ab rebuilt X=1; ab previous CID_LIBRARY=$PREV
cat $O/ab.txt
timeout 300 bash tools/profile_bench.sh $O/prof --no-cpu-baseline --no-torch-baseline --no-secondary > $O/prof.log 2>&1; t "rocprof stats"
rm -rf $O/prof/raw
python tools/collect_profiles.py $O r04 --stats-only > /dev/null
timeout 420 python bench.py > $O/bench_default.json 2> $O/bench_default.err; t "bench default"
tail -1 $O/bench_default.json | cut -c1-400
timeout 200 python tools/kbench.py > $O/kbench.txt 2>&1; t "kbench (rebuilt)"
bash tools/pmc_run.sh xattn3 $O/pmc_xattn > $O/pmc_xattn.txt 2>&1; rm -rf $O/pmc_xattn/p*/; t "pmc xattn3"
cat $O/timeline.txt
