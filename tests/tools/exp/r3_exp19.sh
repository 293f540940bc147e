cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_real_configs_gpu.py tests/test_headline_parity_gpu.py tests/test_las_gpu.py -x -q -m gpu -k "decoder or las_small or whole_model" > gpurun_out/r3_exp19_tests.log 2>&1 || { tail -40 gpurun_out/r3_exp19_tests.log; exit 1; }
tail -3 gpurun_out/r3_exp19_tests.log
timeout -k 10 120 python tests/tools/bench_decoder_sweep.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 60 > gpurun_out/r3_exp19.json 2> gpurun_out/r3_exp19.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3_exp19.json').read().strip().splitlines()[-1]); print('las_small run $i', d['ms_per_step'], d['config'].get('remeasured'), d['config'].get('sweep_errors'))
PY
done
