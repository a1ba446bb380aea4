cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweepx
run() { name=$1; shift; env "$@" python tools/kbench.py --only gemm --family sdxl --b2 4 > gpurun_out/sweepx/$name.txt 2>&1; }
run default X=1
run t1 CID_GEMM_TILE=1
run t2 CID_GEMM_TILE=2
run t3 CID_GEMM_TILE=3
run t1sk2 CID_GEMM_TILE=1 CID_GEMM_SK=2
run t1sk4 CID_GEMM_TILE=1 CID_GEMM_SK=4
run t2sk2 CID_GEMM_TILE=2 CID_GEMM_SK=2
run t3sk2 CID_GEMM_TILE=3 CID_GEMM_SK=2
run g1 CID_GEGLU_TILE=1
run g2 CID_GEGLU_TILE=2
