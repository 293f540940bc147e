cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_trainstep_gpu.py tests/test_ds2_gpu.py tests/test_dp_gpu.py -x -q -m gpu > gpurun_out/r3_exp15_tests.log 2>&1 || { tail -40 gpurun_out/r3_exp15_tests.log; exit 1; }
tail -3 gpurun_out/r3_exp15_tests.log
timeout -k 10 120 python tests/tools/bench_sweep.py --shapes las_small,deepspeech --iters 20 2>&1 | grep -v amdgpu.ids
for w in las_small deepspeech; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 40 --workload $w > gpurun_out/r3_exp15_$w.json 2> gpurun_out/r3_exp15_$w.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3_exp15_$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['value'], d['config'].get('remeasured'), d['config'].get('sweep_errors'))
PY
done
