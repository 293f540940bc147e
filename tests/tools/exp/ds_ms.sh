# ms per deepspeech training step under one environment variable's values:  bash tests/tools/exp/ds_ms.sh ASR_DS2_CONV_BESIDE 0 1
name=$1; shift
for v in "$@"; do
  echo -n "$name=$v: "
  env $name=$v python bench.py --workload deepspeech --no-extra-workloads --no-dp-path --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'])"
done
