cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export ASR_OVERLAP=0
for t in 0 2 4; do
export ASR_GEMM_TILE=$t
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_tr12 -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-rooflines > gpurun_out/r3_tr12.log 2>&1
f=$(find gpurun_out/r3_tr12 -name "*kernel_stats.csv" | head -1)
python tests/tools/per_step.py $f 25 60 > gpurun_out/r3_perstep12_t$t.txt
rm -rf gpurun_out/r3_tr12
done
