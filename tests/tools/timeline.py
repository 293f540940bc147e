"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: python tests/tools/timeline.py TRACE.csv [marker] [nth-from-end]
Prints every kernel of the step (a step starts at a `marker` kernel, default logmel_kernel) with its start offset, duration, queue
and how much of it ran while another kernel was also running - to see what actually overlaps."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "logmel_kernel"
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda k: k[0])
starts = [i for i, k in enumerate(ks) if marker in k[2]]
i0, i1 = starts[-nth], starts[-nth + 1] if nth > 1 else len(ks)
step = ks[i0:i1]
t0 = step[0][0]
print(f"step of {len(step)} kernels, {(max(k[1] for k in step) - t0) / 1e6:.3f} ms from first start to last end")
busy = sum(k[1] - k[0] for k in step)
print(f"sum of kernel durations {busy / 1e6:.3f} ms")
for n, (s, e, name, q) in enumerate(step):
    ov = 0
    for m, (s2, e2, _, _) in enumerate(step):
        if m != n:
            ov += max(0, min(e, e2) - max(s, s2))
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{q:>3}  overlapped {ov / 1e3:8.1f} us  {name[:90]}")
