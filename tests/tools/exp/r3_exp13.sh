cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export ASR_OVERLAP=0
b() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 40 $EXTRA > gpurun_out/r3_exp13_$name.json 2> gpurun_out/r3_exp13_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3_exp13_$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['value'], d['config'].get('remeasured'), d['config'].get('sweep_errors'))
except Exception as e: print('$name', 'ERR', e)
PY
}
b small_e102 ASR_GEMM_E64=1.02
b small_e095 ASR_GEMM_E64=0.95
b small_e110 ASR_GEMM_E64=1.10
b small_e080 ASR_GEMM_E64=0.80
EXTRA="--workload deepspeech" b ds_e102 ASR_GEMM_E64=1.02
EXTRA="--workload deepspeech" b ds_e080 ASR_GEMM_E64=0.80
EXTRA="--workload las_large --steps 8" b large_e102 ASR_GEMM_E64=1.02
EXTRA="--workload las_large --steps 8" b large_e080 ASR_GEMM_E64=0.80
