cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_rnn_gpu.py tests/test_real_configs_gpu.py tests/test_ds2_gpu.py tests/test_las_gpu.py tests/test_layers_gpu.py -x -q -m gpu > gpurun_out/r3_exp21_tests.log 2>&1 || { tail -40 gpurun_out/r3_exp21_tests.log; exit 1; }
tail -3 gpurun_out/r3_exp21_tests.log
for w in las_small deepspeech; do for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --no-extra-workloads --steps 60 --workload $w > gpurun_out/r3_exp21.json 2> gpurun_out/r3_exp21.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3_exp21.json').read().strip().splitlines()[-1]); print('$w run $i', d['ms_per_step'], d['config'].get('sweep_errors'))
PY
done; done
