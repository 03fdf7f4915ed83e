#!/usr/bin/env python3
"""Batch-1 inference timeline from a rocprofv3 --kernel-trace results.db of tools/infer_graph.py: the kernels of the LAST hipGraph replay in launch order
with their durations and the idle gap in front of each.  Usage: python tools/infer_trace.py <dir-or-db>"""
import glob, os, sqlite3, sys
src = sys.argv[1]
db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True))[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# a replay starts with the layout conversion of the input image
starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r[0]]
a = starts[-2]; b = starts[-1]
ks = rows[a:b]
t0 = ks[0][1]
tot = 0
print("%-72s %9s %9s" % ("kernel", "dur us", "gap us"))
prev_end = ks[0][1]
for name, s, e in ks:
    print("%-72s %9.1f %9.1f" % (name.replace("void uegan::", "")[:72], (e - s) / 1e3, (s - prev_end) / 1e3))
    tot += e - s
    prev_end = max(prev_end, e)
print("launches %d  kernel sum %.1f us  wall %.1f us (first start -> last end)  next replay starts %.1f us after the first" % (len(ks), tot / 1e3, (prev_end - t0) / 1e3, (rows[b][1] - t0) / 1e3))
