#!/usr/bin/env python
"""Per-kernel mean of rocprofv3 --pmc counters (csv output).  usage: pmc_summary.py <dir>... [--json out.json]"""
import csv, sys, json, collections
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if out_json: dirs.remove(out_json)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    per_dispatch = collections.defaultdict(float)
    names = {}
    for row in csv.DictReader(open(f"{d}/p_counter_collection.csv")):
        k = (row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[k] += float(row["Counter_Value"])
        names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for (did, cname), val in per_dispatch.items():
        kn = names[did]
        if "gsr::" not in kn and "styl3r" not in kn: continue
        kn = kn.split("(")[0].replace("void ", "")
        agg[kn][cname].append(val)
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
for k, cs in sorted(res.items()):
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:28s} {v:16.1f}")
if out_json: json.dump(res, open(out_json, "w"), indent=1)
