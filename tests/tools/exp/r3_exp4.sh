set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for ov in 0 1; do
export ASR_OVERLAP=$ov
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_tr_ov$ov -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --no-graph > gpurun_out/r3_tr_ov$ov.log 2>&1
f=$(find gpurun_out/r3_tr_ov$ov -name "*kernel_trace.csv" | head -1)
python tests/tools/timeline.py $f > gpurun_out/r3_timeline_ov$ov.txt
f=$(find gpurun_out/r3_tr_ov$ov -name "*kernel_stats.csv" | head -1)
python tests/tools/per_step.py $f 9 25 > gpurun_out/r3_perstep_ov$ov.txt
rm -rf gpurun_out/r3_tr_ov$ov
done
echo done
