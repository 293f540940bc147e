set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_rnn_gpu.py tests/test_real_configs_gpu.py -x -q -m gpu -k "persistent or las_small or through_persistent" > gpurun_out/r3_exp5_tests.log 2>&1 || { tail -30 gpurun_out/r3_exp5_tests.log; exit 1; }
tail -3 gpurun_out/r3_exp5_tests.log
b() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 40 $EXTRA > gpurun_out/r3_exp5_$name.json 2> gpurun_out/r3_exp5_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3_exp5_$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['value'], d['config'].get('remeasured'))
except Exception as e: print('$name', 'ERR', e)
PY
}
b ov0 ASR_OVERLAP=0
b ov0_pr1 ASR_OVERLAP=0 ASR_SWEEP_BWD_PROBE=1
b ov1 ASR_OVERLAP=1
b ov1_pr1 ASR_OVERLAP=1 ASR_SWEEP_BWD_PROBE=1
b ov1_pr2 ASR_OVERLAP=1 ASR_SWEEP_BWD_PROBE=2
b ov1_pr4 ASR_OVERLAP=1 ASR_SWEEP_BWD_PROBE=4
b ov1_pr8 ASR_OVERLAP=1 ASR_SWEEP_BWD_PROBE=8
echo done
