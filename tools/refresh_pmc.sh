# shortest refresh after a kernel-source change: PMC passes of the roofline kernel (so that bench.py reports
# roofline.traffic for the new digest) + one bench line without the CPU / stock-PyTorch comparators
set -u
O=gpurun_out/final3
rm -rf $O; mkdir -p $O
bash tools/pmc_run.sh xattn3 $O/pmc_xattn > $O/pmc_xattn.txt 2>&1
rm -rf $O/pmc_xattn/p*/
python tools/collect_profiles.py $O r02 --pmc-only
python bench.py --no-cpu-baseline --no-torch-baseline > $O/bench_default_final.json 2> $O/bench.err
tail -1 $O/bench_default_final.json | cut -c1-900
