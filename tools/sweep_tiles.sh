cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep
run() { name=$1; shift; env "$@" python tools/kbench.py --only gemm > gpurun_out/sweep/$name.txt 2>&1; }
run default X=1
run t2 CID_GEMM_TILE=2
run t3 CID_GEMM_TILE=3
run t1sk2 CID_GEMM_TILE=1 CID_GEMM_SK=2
run t1sk4 CID_GEMM_TILE=1 CID_GEMM_SK=4
run t1sk5 CID_GEMM_TILE=1 CID_GEMM_SK=5
run t1sk10 CID_GEMM_TILE=1 CID_GEMM_SK=10
run t2sk2 CID_GEMM_TILE=2 CID_GEMM_SK=2
run t2sk4 CID_GEMM_TILE=2 CID_GEMM_SK=4
run t3sk2 CID_GEMM_TILE=3 CID_GEMM_SK=2
run t3sk4 CID_GEMM_TILE=3 CID_GEMM_SK=4
