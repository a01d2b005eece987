#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of the DEFAULT bench command runs every engine kernel at several sizes (the headline steps, the nested
configurations, the end-to-end legs' ranges), so `--stats` averages mix them.  This groups the trace's dispatches by (kernel, grid size):
    python tools/summarize_default_profile.py <kernel_trace.csv> > profiles/rNN_bench_default_n1_kernel_groups.csv
The group of the headline kernel with the full-size grid is what `roofline.kernel_ms` of the same run's line must agree with."""
import collections
import csv
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "", 1).split("(")[0]     # (k_reduce is not a template: no leading "void")
    if not name.startswith("k_"):
        continue
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * int(r.get("Grid_Size_Y", 1) or 1)
    rows[(name, grid, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_threads", "workgroup", "calls", "avg_ms", "min_ms", "max_ms"])
for (name, grid, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, grid, wg, len(v), f"{sum(v) / len(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])
