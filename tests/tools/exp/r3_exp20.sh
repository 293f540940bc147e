cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_real_configs_gpu.py tests/test_headline_parity_gpu.py -x -q -m gpu -k "15_second or decoder_sweep_equals" --durations=5 > gpurun_out/r3_exp20_tests.log 2>&1 || { tail -40 gpurun_out/r3_exp20_tests.log; exit 1; }
tail -12 gpurun_out/r3_exp20_tests.log
timeout -k 10 120 python tests/tools/bench_decoder_sweep.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline --no-kernel-rooflines --no-extra-workloads > gpurun_out/r3_exp20_bench.json 2> gpurun_out/r3_exp20_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3_exp20_bench.json').read().strip().splitlines()[-1]); print('las_small', d['ms_per_step'], d['config'].get('sweep_errors'))
PY
