cd $GRAFT_REPO_ROOT
for t in 0 1 2 4; do echo "== ASR_GEMM_TILE=$t"; ASR_GEMM_TILE=$t timeout -k 10 120 python tests/tools/bench_beside.py 2>&1 | grep -v amdgpu.ids; done
