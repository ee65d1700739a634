#!/usr/bin/env python
"""GPU-side anatomy of single-query searches from a rocprofv3 kernel trace summarised by scripts/rocpd_summary.py:
groups the dispatches into calls (a gap of more than 100 us starts a new call), prints for the median of the last calls the
number of launches, the sum of kernel durations and the span first start -> last end.
    python scripts/lab/single_query_timeline.py <..._dispatches.csv>"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
calls, cur = [], []
prev_end = None
for r in rows:
    st, du = float(r["start"]), float(r["duration"])
    if prev_end is not None and st - prev_end > 100.0:
        calls.append(cur)
        cur = []
    cur.append((r["name"], st, du))
    prev_end = st + du
calls.append(cur)
tail = [c for c in calls[-30:] if len(c) > 5]
spans = [c[-1][1] + c[-1][2] - c[0][1] for c in tail]
sums = [sum(x[2] for x in c) for c in tail]
print("calls analysed", len(tail), "launches per call", int(np.median([len(c) for c in tail])),
      "kernel time sum us %.1f" % np.median(sums), "GPU span us %.1f" % np.median(spans))
c = tail[len(tail) // 2]
t0 = c[0][1]
for name, st, du in c:
    print("  +%7.1f us  %6.1f us  %s" % (st - t0, du, name[:70]))
