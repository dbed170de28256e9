#!/bin/bash
# usage (GPU box, repo root): bash tools/probes/fetch_gather_calib.sh <out.md>
set -u
OUT=${1:-gpurun_out/fetch_gather_calib.md}; export TMPDIR=/tmp
hipcc -w --offload-arch=gfx950 -O3 tools/probes/fetch_gather_calib.hip -o /tmp/fgc || exit 1
/tmp/fgc > /tmp/fgc_plain.txt
rm -rf /tmp/fgc_pmc /tmp/fgc_pmc_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fgc_pmc -o p -- /tmp/fgc > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/fgc_pmc_w -o p -- /tmp/fgc > /dev/null 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1], "w")
out.write("# FETCH_SIZE on a 48-byte gather vs a coalesced stream (2^26 records, 3.2 GB, each read once)\n\n")
out.write(open("/tmp/fgc_plain.txt").read().replace("\n", "  \n") + "\n")
n = 1 << 26
for d in ("/tmp/fgc_pmc", "/tmp/fgc_pmc_w"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/p_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        if "fill" in k: continue
        for c, vals in sorted(cs.items()):
            v = sum(vals) / len(vals)
            extra = f" = {v * 1024 / n:.1f} B per record raw (KiB units), x2 = {2 * v * 1024 / n:.1f}" if c == "FETCH_SIZE" else f" = {v / n:.3f} per record"
            out.write(f"* `{k}` {c}: {v:.4g}{extra}\n")
out.close()
print(open(sys.argv[1]).read())
PY
