set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for s in none enc dec enc,dec; do
  if [ $s = none ]; then unset ASR_SIDE_STREAM; else export ASR_SIDE_STREAM=$s; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --steps 40 > gpurun_out/r3_exp1_$s.json 2> gpurun_out/r3_exp1_$s.err || exit 1
done
unset ASR_SIDE_STREAM
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-rooflines --no-graph --steps 40 > gpurun_out/r3_exp1_nograph.json 2> gpurun_out/r3_exp1_nograph.err
echo done
