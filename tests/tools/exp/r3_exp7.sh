cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for dbg in 0 16; do for prio in 0 3; do
echo "== dbg $dbg prio $prio"
ASR_SWEEP_DBG=$dbg ASR_SWEEP_PRIO=$prio timeout -k 10 120 python tests/tools/bench_sweep.py --shapes las_small,deepspeech --iters 20 2>&1 | grep -v amdgpu.ids
done; done
