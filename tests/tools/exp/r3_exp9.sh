cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_rnn_gpu.py -x -q -m gpu -k "persistent" 2>&1 | tail -2
b() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 40 $EXTRA > gpurun_out/r3_exp9_$name.json 2> gpurun_out/r3_exp9_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3_exp9_$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['value'], d['config'].get('remeasured'), d['config'].get('sweep_errors'))
except Exception as e: print('$name', 'ERR', e)
PY
}
b ov0 ASR_OVERLAP=0
b ov1 ASR_OVERLAP=1
for t in 2 3 4; do
b ov0_t$t ASR_OVERLAP=0 ASR_GEMM_TILE=$t
b ov1_t$t ASR_OVERLAP=1 ASR_GEMM_TILE=$t
done
