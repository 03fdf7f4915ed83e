#!/usr/bin/env python3
"""GPU idle time of the training step from a rocprofv3 --kernel-trace results.db (rocpd sqlite): per step (a step ends with the generator's
Adam launch -- the second `adam_kernel` of the step), wall time, time with at least one kernel running (union over both streams), idle time
= wall - busy, the number of launches, and the sum of kernel durations.  Usage: python tools/gap_analysis.py <dir-or-db> [out.json]"""
import glob
import json
import os
import sqlite3
import sys

src = sys.argv[1]
db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True))[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
ends = adam[1::2]                      # every second Adam launch closes a step
steps = []
for a, b in zip(ends[:-1], ends[1:]):
    ks = rows[a + 1:b + 1]
    t0, t1 = rows[a][2], max(k[2] for k in ks)
    iv = sorted((max(k[1], t0), k[2]) for k in ks)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps = sorted(((iv[i + 1][0] - max(x[1] for x in iv[:i + 1])) for i in range(len(iv) - 1)), reverse=True)
    steps.append({"wall_ms": (t1 - t0) / 1e6, "busy_ms": busy / 1e6, "idle_ms": (t1 - t0 - busy) / 1e6, "launches": len(ks),
                  "kernel_sum_ms": sum(k[2] - k[1] for k in ks) / 1e6, "largest_gaps_us": [round(g / 1e3, 1) for g in gaps[:5] if g > 0]})
out = {"steps": steps}
if steps:
    n = len(steps)
    out["mean"] = {k: round(sum(s[k] for s in steps) / n, 3) for k in ("wall_ms", "busy_ms", "idle_ms", "launches", "kernel_sum_ms")}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
