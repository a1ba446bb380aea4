# one GPU call: third-generation fused cross-attention -- parity, launch time beside generation 2, phase trace
# usage: bash tools/x3_ab.sh [variant ...]   (libcid_<variant>.so built with consistentid_amd.build --variant)
set -u
O=gpurun_out/x3ab
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cross_attention_v3" -x 2>&1 | tail -4 | tee $O/pytest.txt
timeout 200 python tools/kbench.py --only xattn2 2>&1 | tee $O/kbench.txt | grep "id-xattn"
for lib in "$@"; do
  echo "== variant $lib"
  case $lib in
    *trace*) CID_LIBRARY=consistentid_amd/libcid_$lib.so timeout 200 python tools/x2_trace.py --gen 3 2>&1 | grep -v amdgpu.ids | tee $O/x3_$lib.txt ;;
    *) CID_LIBRARY=consistentid_amd/libcid_$lib.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cross_attention_v3" -x 2>&1 | tail -2
       CID_LIBRARY=consistentid_amd/libcid_$lib.so timeout 200 python tools/kbench.py --only xattn2 2>&1 | grep "id-xattn3" | tee -a $O/kbench_$lib.txt ;;
  esac
done
