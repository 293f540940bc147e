set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export ASR_OVERLAP=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_tr6 -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-rooflines > gpurun_out/r3_tr6.log 2>&1
f=$(find gpurun_out/r3_tr6 -name "*kernel_stats.csv" | head -1)
python tests/tools/per_step.py $f 25 30 > gpurun_out/r3_perstep6_ov0.txt
rm -rf gpurun_out/r3_tr6
export ASR_OVERLAP=1
for i in 1 2 3 4 5 6; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 60 > gpurun_out/r3_exp6_$i.json 2> gpurun_out/r3_exp6_$i.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3_exp6_$i.json').read().strip().splitlines()[-1]); print('run $i', d['ms_per_step'], d['config'].get('remeasured'), d['config'].get('sweep_errors'))
PY
done
echo done
