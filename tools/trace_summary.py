#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace CSV of bench.py to the per-kernel totals of ONE steady-state step
(the window between the last two k_heat_peaks launches of the timed loop) and print/save a small CSV.

    python tools/trace_summary.py <dir with *_kernel_trace.csv> <out.csv> [--delete-trace] [--anchor=KERNEL_SUBSTRING]

(rocprofv3 of ROCm 7 writes a database by default: pass --output-format csv to it.)
"""
import collections
import csv
import glob
import os
import sys

d, out = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
anchor = ([a.split("=", 1)[1] for a in sys.argv if a.startswith("--anchor=")] or ["k_heat_peaks"])[0]
hp = [i for i, r in enumerate(rows) if anchor in r[2]]
# A steady-state step of bench.py is a graph replay: the windows between consecutive K_A launches that hold the MOST COMMON
# kernel count among the big ones (warm-up passes tune and hold many more kernels; pp_time_kernels / verify windows hold few).
sizes = [hp[i + 1] - hp[i] for i in range(len(hp) - 1)]
big = [n for n in sizes if n > 100]
common = collections.Counter(big).most_common(1)[0][0]
last = max(i for i, n in enumerate(sizes) if n == common)
a, b = hp[last], hp[last + 1]
seg = rows[a + 1:b + 1]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in seg:
    agg[k][0] += e - s
    agg[k][1] += 1
busy = sum(v[0] for v in agg.values())
wall = seg[-1][1] - seg[0][0]
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls_per_step", "total_us_per_step", "avg_us", "percent_of_busy"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        w.writerow([k[:160], v[1], round(v[0] / 1e3, 2), round(v[0] / v[1] / 1e3, 2), round(100.0 * v[0] / busy, 2)])
    w.writerow(["# step wall us", "", round(wall / 1e3, 1), "", ""])
    w.writerow(["# step busy us", "", round(busy / 1e3, 1), "", ""])
    w.writerow(["# kernels per step", len(seg), "", "", ""])
print("step wall ms", wall / 1e6, "busy ms", busy / 1e6, "kernels", len(seg))
if "--delete-trace" in sys.argv:
    os.remove(f)
