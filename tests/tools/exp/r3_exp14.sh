cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_headline_parity_gpu.py -x -q -m gpu --durations=8 > gpurun_out/r3_exp14.log 2>&1; tail -40 gpurun_out/r3_exp14.log
