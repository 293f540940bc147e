set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_rnn_gpu.py tests/test_real_configs_gpu.py tests/test_trainstep_gpu.py tests/test_dp_gpu.py tests/test_las_gpu.py tests/test_ds2_gpu.py -x -q -m gpu > gpurun_out/r3_exp3_tests.log 2>&1 || { tail -30 gpurun_out/r3_exp3_tests.log; exit 1; }
tail -3 gpurun_out/r3_exp3_tests.log
b() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 40 $EXTRA > gpurun_out/r3_exp3_$name.json 2> gpurun_out/r3_exp3_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3_exp3_$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['value'], d['config'].get('remeasured'))
except Exception as e: print('$name', 'ERR', e)
PY
}
b ov0 ASR_OVERLAP=0
b ov1_p3 ASR_OVERLAP=1 ASR_SWEEP_PRIO=3
b ov1_p0 ASR_OVERLAP=1 ASR_SWEEP_PRIO=0
b ov0_p0 ASR_OVERLAP=0 ASR_SWEEP_PRIO=0
EXTRA=--no-graph b ov1_p3_nograph ASR_OVERLAP=1 ASR_SWEEP_PRIO=3
EXTRA="--workload deepspeech" b ds_ov0 ASR_OVERLAP=0
EXTRA="--workload deepspeech" b ds_ov1 ASR_OVERLAP=1
EXTRA="--workload las_large --steps 8" b large_ov0 ASR_OVERLAP=0
EXTRA="--workload las_large --steps 8" b large_ov1 ASR_OVERLAP=1
echo done
