# rocprofv3 kernel-trace summaries of the three bench workloads + the PMC passes of the headline (one gpurun call):
#   gpurun --timeout 1150 -- 'bash tests/tools/round_profile.sh r03'
# Writes gpurun_out/<round>_*.{csv,txt}; copy the ones to be judged into profiles/.
R=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && \
B="--no-cpu-baseline --no-kernel-rooflines --no-extra-workloads --no-dp-path" && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -o small -- python3 bench.py --steps 20 --warmup 5 $B > gpurun_out/prof_small.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ds -o ds -- python3 bench.py --workload deepspeech --steps 20 --warmup 5 $B > gpurun_out/prof_ds.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_largebf -o large -- python3 bench.py --workload las_large --steps 5 --warmup 3 $B > gpurun_out/prof_large.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 3 $B --no-graph > gpurun_out/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 3 --warmup 3 $B --no-graph > gpurun_out/pmc_write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_mfma -o m -- python3 bench.py --steps 3 --warmup 3 $B --no-graph > gpurun_out/pmc_mfma.log 2>&1
rc=$?
for w in small:25:las_small ds:25:deepspeech large:8:las_large_bf16; do
  IFS=: read tag steps name <<< "$w"
  f=$(find gpurun_out/prof_$tag* -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f gpurun_out/${R}_${name}_kernel_stats.csv && python tests/tools/per_step.py $f $steps 60 > gpurun_out/${R}_${name}_per_step.txt
done
{ f=$(find gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tests/tools/pmc_summary.py $f 6
  f=$(find gpurun_out/pmc_write -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tests/tools/pmc_summary.py $f 6; } > gpurun_out/${R}_las_small_pmc_hbm_traffic.txt 2>&1
f=$(find gpurun_out/pmc_mfma -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tests/tools/pmc_mfma.py $f 6 > gpurun_out/${R}_las_small_pmc_mfma_util.txt 2>&1
rm -rf gpurun_out/prof_small gpurun_out/prof_ds gpurun_out/prof_largebf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
tail -2 gpurun_out/prof_*.log gpurun_out/pmc_*.log | cut -c 1-300
head -5 gpurun_out/${R}_*_per_step.txt gpurun_out/${R}_las_small_pmc_*.txt
exit $rc
