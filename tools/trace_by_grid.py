"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid): calls, total and mean duration.
    python3 tools/trace_by_grid.py <dir with *_kernel_trace.csv> [min_total_us]"""
import csv, glob, os, sys
from collections import defaultdict
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = csv.DictReader(open(files[-1]))
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    name = name[:name.index("(")] if "(" in name else name
    g = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1),
         int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1))
    a = agg[(name, g)]
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
print("total %.1f us over %d dispatches" % (tot, sum(a[0] for a in agg.values())))
byname = defaultdict(float)
for (n, g), a in agg.items(): byname[n] += a[1]
for n, t in sorted(byname.items(), key=lambda x: -x[1]): print("%-60s %10.1f us %5.1f %%" % (n[:60], t, 100 * t / tot))
print()
for (n, g), a in sorted(agg.items(), key=lambda x: -x[1][1]):
    if a[1] >= lim: print("%-44s grid %-18s calls %5d total %10.1f us mean %9.2f us" % (n[:44], g, a[0], a[1], a[1] / a[0]))
if len(sys.argv) > 3:      # every duration of the kernels whose name contains argv[3], in dispatch order
    rows = csv.DictReader(open(files[-1]))
    d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if sys.argv[3] in r["Kernel_Name"]]
    print(" ".join("%.1f" % t for _, t in sorted(d)[:60]))
