cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do for w in deepspeech las_small; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 60 --workload $w > gpurun_out/r3_exp18.json 2> gpurun_out/r3_exp18.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3_exp18.json').read().strip().splitlines()[-1]); print('$w run $i', d['ms_per_step'], d['config'].get('remeasured'), d['config'].get('sweep_errors'))
PY
done; done
