#!/usr/bin/env python
"""Per one-pass kernel form: median duration of the pass, of the re-rank launch that follows it and of the launch in front of it, from the per-dispatch
CSV scripts/rocpd_summary.py writes (rocprofv3 --kernel-trace):   python one_pass_dispatch_medians.py <..._dispatches.csv>"""
import collections
import csv
import statistics as st
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
stats = collections.defaultdict(list)
for i in range(1, len(rows) - 1):
    r, nxt, prev = rows[i], rows[i + 1], rows[i - 1]
    if "stream8" in r["name"] and "rerank_kernel" in nxt["name"]:
        stats[r["name"][:70]].append((float(r["duration"]), float(nxt["duration"]), prev["name"].split("(")[0][:28], float(prev["duration"])))
print("%-72s %6s %9s %11s   %s" % ("pass kernel", "calls", "pass us", "re-rank us", "launch in front (us)"))
for k, v in sorted(stats.items()):
    print("%-72s %6d %9.1f %11.1f   %s (%.1f)" % (k, len(v), st.median(x[0] for x in v), st.median(x[1] for x in v), v[len(v) // 2][2], st.median(x[3] for x in v)))
