"""Map a rocprofv3 --kernel-trace CSV of tools/layer_table.py (or bench.py) onto the layer plan.

usage: trace_layers.py <kernel_trace.csv> [frames_to_skip]
Prints per-kernel-name totals and, for the last frames, the per-dispatch durations in launch order.
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = defaultdict(lambda: [0, 0])
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"].split("(")[0]
    tot[k][0] += d
    tot[k][1] += 1
print("%-70s %8s %10s %9s" % ("kernel", "calls", "total_us", "avg_us"))
for k, (d, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%-70s %8d %10.1f %9.2f" % (k[:70], n, d / 1e3, d / n / 1e3))
# frame segmentation: a frame starts at pyramid_kernel
starts = [i for i, r in enumerate(rows) if "pyramid_kernel" in r["Kernel_Name"]]
if len(starts) >= 3:
    a, b = starts[-2], starts[-1]
    fr = rows[a:b]
    t0 = int(fr[0]["Start_Timestamp"])
    print("\nlast full frame: %d dispatches, span %.1f us, busy %.1f us" % (
        len(fr), (int(fr[-1]["End_Timestamp"]) - t0) / 1e3,
        sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fr) / 1e3))
    for r in fr:
        print("%9.1f %8.2f  %s  grid=%s wg=%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][-40:],
              r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
