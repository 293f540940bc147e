"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES counter_collection.csv:
python tests/tools/pmc_mfma.py FILE STEPS.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the
chip (MI355X_MICROARCH.md), so utilisation = busy cycles / (kernel duration x 2.4 GHz x 1024)."""
import collections
import csv
import sys

steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
busy, dur, cnt = collections.Counter(), collections.Counter(), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
        continue
    k = r["Kernel_Name"].split("(")[0][:64]
    busy[k] += float(r["Counter_Value"])
    dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    cnt[k] += 1
cap = lambda ns: ns * 2.4 * 1024
tb, td = sum(busy.values()), sum(dur.values())
print(f"whole run: MFMA busy {tb / steps / 1e9:.3f} G SIMD-cycles per step, kernel time {td / steps / 1e6:.2f} ms per step (serialised by the "
      f"counter collection), MFMA utilisation over kernel time {tb / cap(td) * 100:.1f} %")
print(f"{'kernel':64s} {'n/step':>7s} {'ms/step':>8s} {'MFMA util %':>11s}")
for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:22]:
    print(f"{k:64s} {cnt[k] / steps:7.1f} {v / steps / 1e6:8.3f} {busy[k] / cap(v) * 100:11.1f}")
