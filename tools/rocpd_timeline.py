#!/usr/bin/env python
"""Print the kernel timeline of ONE steady-state step from a rocprofv3 rocpd database: every kernel between two
consecutive launches of an anchor kernel (default: the rasterizer's k_preprocess), with start offset, duration and
the idle gap since the previous kernel ended.  Shows where the step time that is not kernel time goes.
usage: python tools/rocpd_timeline.py <results.db> [anchor-substring] [out.md] [--agg]   (--agg: one row per kernel name)"""
import sqlite3
import sys

AGG = "--agg" in sys.argv
if AGG: sys.argv.remove("--agg")
db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_preprocess<"
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0] and "bwd" not in r[0]]
if len(idx) < 3:
    sys.exit(f"anchor {anchor!r} seen {len(idx)} times")
k = len(idx) // 2 if len(idx) > 40 else len(idx) - 3      # a steady-state step from the middle of the timed region (the tail of a bench run is A/B legs and read-backs)
a, b = idx[k], idx[k + 1]
t0, prev_end, busy = rows[a][1], rows[a][1], 0
lines = ["| # | kernel | start us | dur us | gap before us |", "|---|---|---|---|---|"]
for i, (n, s, e) in enumerate(rows[a:b]):
    short = n.split("(")[0][-70:]
    lines.append(f"| {i} | `{short}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {max(0, s - prev_end) / 1e3:.1f} |")
    busy += e - s
    prev_end = max(prev_end, e)
span = rows[b][1] - t0
if AGG:
    import collections
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s_, e in rows[a:b]:
        agg[n][0] += 1; agg[n][1] += e - s_
    lines = ["| kernel | calls/step | ms/step | % of span |", "|---|---|---|---|"]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:80]:
        lines.append(f"| `{n[:110]}` | {c} | {t / 1e6:.3f} | {100 * t / span:.1f} |")
lines.append(f"\nstep span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle {100 * (1 - busy / span):.1f} %")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
