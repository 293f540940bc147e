# per-step kernel profile of a bench workload: bash tests/tools/exp/r3_prof.sh NAME [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
name=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o t -- python3 bench.py --steps ${PROF_STEPS:-20} --warmup 5 --no-cpu-baseline --no-kernel-rooflines "$@" > gpurun_out/prof_$name.log 2>&1
f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
python tests/tools/per_step.py $f $((${PROF_STEPS:-20}+5)) 60 > gpurun_out/perstep_$name.txt
cp $f gpurun_out/kernel_stats_$name.csv
rm -rf gpurun_out/prof_$name
head -16 gpurun_out/perstep_$name.txt | cut -c1-150
