#!/usr/bin/env python
"""One step of a multi-stream run from a rocprofv3 rocpd database: per stream / queue the busy time, launch count and span, then the full
timeline with the stream of every kernel (start offset, duration, gap since the previous kernel on the SAME stream).
usage: python tools/rocpd_streams.py <results.db> [anchor-substring]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_preprocess<"
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nm = "name" if "name" in cols else [c for c in cols if "name" in c][0]
sid = next((c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols), None)
print("# columns:", cols)
rows = cur.execute(f"select {nm}, start, end, {sid or 0} from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0] and "bwd" not in r[0]]
a, b = idx[-3], idx[-2]
seg = rows[a + 1:b + 1]            # from behind one rasterizer call to the end of the next: encoder of the step + its rasterizer start
# the encoder starts after the previous step's last rasterizer kernel: skip the kernels of that rasterizer call
k0 = 0
for i, r in enumerate(seg):
    if r[0].startswith("gsr::") or "gsr::" in r[0][:12]: k0 = i + 1
    else:
        if i > 20: break
seg = seg[k0:]
t0 = seg[0][1]; span = seg[-1][1] - t0
per = collections.defaultdict(lambda: [0, 0, None, None])
for n, s, e, q in seg:
    p = per[q]; p[0] += 1; p[1] += e - s; p[2] = s if p[2] is None else p[2]; p[3] = e
print(f"# span {span / 1e3:.1f} us, kernels {len(seg)}")
for q, (c, t, s, e) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"# stream {q}: {c} kernels, busy {t / 1e3:.1f} us, from +{(s - t0) / 1e3:.1f} to +{(e - t0) / 1e3:.1f} us")
last = {}
for n, s, e, q in seg:
    g = s - last.get(q, s); last[q] = e
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {g / 1e3:6.1f}  q{q}  {n.split('(')[0][-80:]}")
