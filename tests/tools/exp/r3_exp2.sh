set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export ASR_SIDE_STREAM=enc,dec
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --no-graph --steps 40 > gpurun_out/r3_exp2_nograph_side.json 2> gpurun_out/r3_exp2_nograph_side.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_trace_side -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --no-graph > gpurun_out/r3_trace_side.log 2>&1
f=$(find gpurun_out/r3_trace_side -name "*kernel_trace.csv" | head -1)
python tests/tools/timeline.py $f > gpurun_out/r3_timeline_side.txt
rm -rf gpurun_out/r3_trace_side
unset ASR_SIDE_STREAM
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_trace_base -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --no-graph > gpurun_out/r3_trace_base.log 2>&1
f=$(find gpurun_out/r3_trace_base -name "*kernel_trace.csv" | head -1)
python tests/tools/timeline.py $f > gpurun_out/r3_timeline_base.txt
rm -rf gpurun_out/r3_trace_base
echo done
