mkdir -p gpurun_out/p3
for v in "" "CID_GEGLU_TILE=1" "CID_GEGLU_TILE=2" "CID_GEMM_TILE=1" "CID_GEMM_TILE=2" "CID_GEMM_TILE=3"; do
  echo "== $v" >> gpurun_out/p3/tile_ab.txt
  env $v python tools/kbench.py --only gemm 2>&1 | grep -E "^(lin|geglu L|qkv L|conv3 L0 320->320 |conv3 L1 640->640 )" | grep -v "LN fold\|gn-stats" >> gpurun_out/p3/tile_ab.txt
done
cat gpurun_out/p3/tile_ab.txt
