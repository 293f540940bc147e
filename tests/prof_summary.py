"""Summarise a rocprofv3 kernel_stats.csv per training step: python tests/prof_summary.py FILE STEPS"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:64]:64s} calls/step {int(r['Calls'])/steps:7.1f} avg_us {float(r['AverageNs'])/1e3:8.1f} "
          f"ms/step {float(r['TotalDurationNs'])/steps/1e6:7.3f} {float(r['TotalDurationNs'])/tot*100:5.1f}%")
print("total ms/step", tot / steps / 1e6)
