"""Kernel timeline of ONE step out of a rocprofv3 --kernel-trace csv: name, start offset, duration, gap to the previous
kernel's end (any stream).  usage: step_timeline.py <dir> [index of the step counted from the end, default 3]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# a step starts at the prep kernel of the LRT forward
starts = [i for i, r in enumerate(rows) if "kl_kernel<" in r[2] or "prep_kernel" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-k - 1], starts[-k]
t0 = rows[a][0]
prev_end = t0
tot_gap = 0
for s, e, n, q in rows[a:b]:
    gap = (s - prev_end) / 1e3
    tot_gap += max(gap, 0)
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  q{q}  {n[:90]}")
    prev_end = max(prev_end, e)
print(f"step {(rows[b][0] - t0) / 1e3:.1f} us, sum of positive gaps {tot_gap:.1f} us, kernels {b - a}")
