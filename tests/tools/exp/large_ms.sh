# ms per las_large training step under one environment variable's values:  bash tests/tools/exp/large_ms.sh ASR_RNN_WIDE_CFG 0 1 2
name=$1; shift
for v in "$@"; do
  echo -n "$name=$v: "
  env $name=$v python bench.py --workload las_large --steps 4 --warmup 3 --no-extra-workloads --no-dp-path --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'])"
done
