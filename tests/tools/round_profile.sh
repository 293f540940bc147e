cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
timeout -k 10 300 python bench.py > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err && \
timeout -k 10 300 python bench.py --workload deepspeech --no-cpu-baseline > gpurun_out/bench_ds.json 2> gpurun_out/bench_ds.err && \
timeout -k 10 300 python bench.py --workload las_large --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_large_bf16.json 2> gpurun_out/bench_large.err && \
timeout -k 10 300 python bench.py --workload las_large --precision f32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_large_f32.json 2>> gpurun_out/bench_large.err && \
timeout -k 10 300 python bench.py --precision bf16 --no-cpu-baseline > gpurun_out/bench_small_bf16.json 2>> gpurun_out/bench_small.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -o small -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_small.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ds -o ds -- python3 bench.py --workload deepspeech --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_ds.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_largebf -o large -- python3 bench.py --workload las_large --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/prof_large.log 2>&1; \
for f in gpurun_out/bench_*.json; do echo $f; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $f | head -3 | tr '\n' ' '; echo; done
