#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace results.db (rocpd sqlite) into a per-kernel stats table (text).
Usage: python tools/rocprof_summary.py <dir-or-db> <out.txt> [title]"""
import glob
import os
import re
import sqlite3
import sys

src, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else ""
db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True))[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write("# %s\n# source: rocprofv3 --kernel-trace --stats (rocpd db %s)\n# total kernel time %.3f ms over %d dispatches\n" % (
        title, os.path.basename(db), tot / 1e6, sum(r[1] for r in rows)))
    f.write("%-100s %8s %12s %7s %12s %12s %12s\n" % ("kernel", "calls", "total_ms", "pct", "avg_us", "min_us", "max_us"))
    for n, cnt, t, avg, mn, mx in rows:
        n = re.sub(r"\(.*", "", n)
        f.write("%-100s %8d %12.3f %6.2f%% %12.2f %12.2f %12.2f\n" % (n[:100], cnt, t / 1e6, 100.0 * t / tot, avg / 1e3, mn / 1e3, mx / 1e3))
print("wrote", out)
