"""Busy time and idle gaps of one stream from a rocprofv3 --kernel-trace CSV: how much of a solve's wall time the GPU spends between
kernels (host round trips, launch latency) rather than in them.   python tools/trace_gaps.py <kernel_trace.csv> [first_kernel_substr]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
s = [int(r["Start_Timestamp"]) for r in rows]; e = [int(r["End_Timestamp"]) for r in rows]
busy = sum(b - a for a, b in zip(s, e))
span = e[-1] - s[0]
gaps = [s[i + 1] - e[i] for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
print("kernels %d, span %.3f s, busy %.3f s (%.1f %%), gaps %.3f s" % (len(rows), span / 1e9, busy / 1e9, 100.0 * busy / span, sum(pos) / 1e9))
hist = defaultdict(lambda: [0, 0])
for g in pos:
    k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"
    hist[k][0] += 1; hist[k][1] += g
for k in ("<2us", "<5us", "<10us", "<20us", "<100us", ">=100us"):
    print("  gaps %-7s %8d  total %.3f s" % (k, hist[k][0], hist[k][1] / 1e9))
# which kernel precedes the large gaps
by = defaultdict(lambda: [0, 0])
for i, g in enumerate(gaps):
    if g >= 5000:
        n = rows[i]["Kernel_Name"]; n = n[n.find("k_"):n.find("(", n.find("k_"))] if "k_" in n else n[:40]
        by[n][0] += 1; by[n][1] += g
for n, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
    print("  gap >= 5 us behind %-34s %7d  %.3f s" % (n, v[0], v[1] / 1e9))
