cd $GRAFT_REPO_ROOT
for m in saved coef both; do echo "== forward writes: $m"; BENCH_SWEEP_OUT=$m ASR_SWEEP_DBG=0 timeout -k 10 120 python tests/tools/bench_sweep.py --shapes las_small,deepspeech --iters 30 2>&1 | grep "sweep=1" | cut -c1-120; done
