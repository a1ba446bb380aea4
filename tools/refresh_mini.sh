set -u
O=gpurun_out/final2
rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_vae.py -q -m gpu -k "sdxl" 2>&1 | tail -3
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/pmc_run.sh xattn3 $O/pmc_xattn > $O/pmc_xattn.txt 2>&1
rm -rf $O/pmc_xattn/p*/
CID_LIBRARY=consistentid_amd/libcid_trace.so python tools/x2_trace.py --gen 3 > $O/xattn_trace.txt 2>&1
tail -1 $O/bench_default.json | cut -c1-300
tail -24 $O/xattn_trace.txt
