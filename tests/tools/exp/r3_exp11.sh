cd $GRAFT_REPO_ROOT
for t in 1 2 4; do echo "== ASR_GEMM_TILE=$t"; ASR_GEMM_TILE=$t timeout -k 10 120 python tests/tools/gemm_k_sweep.py 7968 1024 2>&1 | grep -v amdgpu.ids; done
echo "== default tiles"; timeout -k 10 120 python tests/bench_gemm.py 2>&1 | grep -v amdgpu.ids
