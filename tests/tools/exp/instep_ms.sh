# ms per las_small training step under one environment variable's values:  bash tests/tools/exp/instep_ms.sh ASR_SWEEP_DELAY 40 50 60
name=$1; shift
for v in "$@"; do
  echo -n "$name=$v: "
  env $name=$v python bench.py --no-extra-workloads --no-dp-path --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'])"
done
