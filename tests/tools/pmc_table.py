"""Per (kernel, grid) table of rocprofv3 --pmc counters: python tests/tools/pmc_table.py counter_collection.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(key, r["Counter_Name"])] += 1
names = sorted({c for v in acc.values() for c in v})
print("kernel,grid," + ",".join(names))
for key, v in acc.items():
    print(f"{key[0]},{key[1]}," + ",".join(f"{v[c] / max(n[(key, c)], 1):.4g}" for c in names))
