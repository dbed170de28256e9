#!/usr/bin/env python
"""Where is the GPU idle inside a step?  Between two launches of an anchor kernel: total idle, a histogram of gap
sizes and the N largest gaps with the kernels on either side.
usage: python tools/rocpd_gaps.py <results.db> [anchor-substring] [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_preprocess<"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nm = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {nm}, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0] and "bwd" not in r[0]]
a, b = idx[-2], idx[-1]
seg = rows[a:b + 1]
span = seg[-1][1] - seg[0][1]
gaps, prev_end, busy = [], seg[0][2], seg[0][2] - seg[0][1]
for i in range(1, len(seg) - 1):
    n, s, e = seg[i]
    g = s - prev_end
    if g > 0: gaps.append((g, i))
    busy += max(0, e - max(s, prev_end)); prev_end = max(prev_end, e)
tot = sum(g for g, _ in gaps)
print(f"step span {span / 1e6:.2f} ms, kernels {len(seg) - 1}, idle {tot / 1e6:.2f} ms ({100 * tot / span:.1f} %)")
for lo, hi in ((0, 5), (5, 20), (20, 100), (100, 1000), (1000, 1e9)):
    sel = [g for g, _ in gaps if lo * 1e3 <= g < hi * 1e3]
    print(f"  gaps {lo:>5}-{hi:<6} us: {len(sel):5d}  total {sum(sel) / 1e6:7.2f} ms")
print(f"largest {N} gaps:")
for g, i in sorted(gaps, reverse=True)[:N]:
    short = lambda n: n.split("(")[0][-60:]
    print(f"  {g / 1e3:9.1f} us at +{(seg[i][1] - seg[0][1]) / 1e6:8.2f} ms  after `{short(seg[i - 1][0])}` before `{short(seg[i][0])}`")
