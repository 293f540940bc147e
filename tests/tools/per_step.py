"""Per-training-step view of a rocprofv3 kernel_stats.csv: python tests/tools/per_step.py FILE PROFILED_STEPS [rows]
(PROFILED_STEPS = warm-up + timed steps of the profiled bench.py command; the per-kernel roofline timing loops of bench.py
add a few dozen calls to the kernels they time, visible as fractional calls/step)."""
import csv
import sys

steps = float(sys.argv[2])
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot / steps / 1e6:.3f} ms over {steps:.0f} profiled steps")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{r['Name'][:64]:64s} calls/step {float(r['Calls']) / steps:7.1f} avg_us {float(r['AverageNs']) / 1e3:8.1f} "
          f"ms/step {float(r['TotalDurationNs']) / steps / 1e6:7.3f} {float(r['Percentage']):5.1f}%")
