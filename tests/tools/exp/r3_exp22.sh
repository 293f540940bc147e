cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for s in 1 0 1 0; do
ASR_SIDE_STREAM=$s timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --no-extra-workloads --steps 60 --workload deepspeech > gpurun_out/r3_exp22.json 2> gpurun_out/r3_exp22.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3_exp22.json').read().strip().splitlines()[-1]); print('deepspeech side=$s', d['ms_per_step'], d['config'].get('sweep_errors'))
PY
done
