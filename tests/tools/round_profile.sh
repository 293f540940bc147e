# rocprofv3 kernel-trace summaries of the three bench workloads + the PMC passes of the headline (one gpurun call):
#   gpurun --timeout 1100 -- 'bash tests/tools/round_profile.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -o small -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-rooflines > gpurun_out/prof_small.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ds -o ds -- python3 bench.py --workload deepspeech --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-rooflines > gpurun_out/prof_ds.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_largebf -o large -- python3 bench.py --workload las_large --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-rooflines > gpurun_out/prof_large.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --no-graph > gpurun_out/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --no-graph > gpurun_out/pmc_write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_mfma -o m -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --no-graph > gpurun_out/pmc_mfma.log 2>&1; \
find gpurun_out/prof_small gpurun_out/prof_ds gpurun_out/prof_largebf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma -name "*.csv" | head -40; tail -2 gpurun_out/prof_*.log gpurun_out/pmc_*.log | cut -c 1-300
