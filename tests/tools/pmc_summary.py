"""Sum a rocprofv3 --pmc counter_collection.csv per kernel: python tests/tools/pmc_summary.py FILE [STEPS]
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 bytes (rocprofv3 'kilobytes')."""
import collections
import csv
import sys

steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot, cnt = collections.Counter(), collections.Counter()
name = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:70]
    tot[k] += float(r["Counter_Value"])
    cnt[k] += 1
    name = r["Counter_Name"]
allv = sum(tot.values())
print(f"{name}: {len(tot)} kernels, {sum(cnt.values())} dispatches, total {allv / 1024:.1f} MB over the run, {allv / 1024 / steps:.1f} MB per step")
for k, v in tot.most_common(14):
    print(f"{k:70s} n/step {cnt[k] / steps:7.1f}  MB/step {v / 1024 / steps:9.2f}")
