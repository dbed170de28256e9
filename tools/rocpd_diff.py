#!/usr/bin/env python
"""steady-state per-step kernel table = (profile with B steps - profile with A steps) / (B - A).
usage: rocpd_diff.py <dbA> <stepsA> <dbB> <stepsB> [out.md]"""
import sqlite3, sys
def load(path):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    nm = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    return {r[0]: (r[1], r[2]) for r in cur.execute(f"select {nm}, count(*), sum(end-start) from kernels group by {nm}")}
a, na, b, nb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
rows = []
for k, (cb, tb) in b.items():
    ca, ta = a.get(k, (0, 0))
    d = (tb - ta) / (nb - na)
    if d > 0: rows.append((k, (cb - ca) / (nb - na), d))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
lines = [f"steady-state GPU kernel time per step: {tot / 1e6:.2f} ms", "", "| kernel | calls/step | ms/step | % |", "|---|---|---|---|"]
for k, c, d in rows[:40]:
    lines.append(f"| `{k[:100]}` | {c:.1f} | {d / 1e6:.3f} | {100 * d / tot:.1f} |")
print("\n".join(lines))
if len(sys.argv) > 5: open(sys.argv[5], "w").write("\n".join(lines) + "\n")
