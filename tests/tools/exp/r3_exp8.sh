cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 120 python tests/tools/bench_sweep.py --shapes las_small,deepspeech --iters 20 2>&1 | grep -v amdgpu.ids
b() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 40 $EXTRA > gpurun_out/r3_exp8_$name.json 2> gpurun_out/r3_exp8_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3_exp8_$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['value'], d['config'].get('remeasured'), d['config'].get('sweep_errors'))
except Exception as e: print('$name', 'ERR', e)
PY
}
b ov0 ASR_OVERLAP=0
b ov1 ASR_OVERLAP=1
b ov1_pr1 ASR_OVERLAP=1 ASR_SWEEP_BWD_PROBE=1
b ov1_pr4 ASR_OVERLAP=1 ASR_SWEEP_BWD_PROBE=4
b ov1_p0 ASR_OVERLAP=1 ASR_SWEEP_PRIO=0
EXTRA="--workload deepspeech" b ds_ov0 ASR_OVERLAP=0
EXTRA="--workload deepspeech" b ds_ov1 ASR_OVERLAP=1
