#!/bin/bash
# PMC passes over the attention lab: matrix-pipe busy, VALU / LDS instruction counts, waits -- per attention kernel.
# usage: tools/probes/attn_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1
CMD="python tools/probes/attn_lab.py"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/ap_${TAG}_a -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/ap_${TAG}_b -o p -- $CMD > /dev/null 2>&1
python - <<'P' $TAG
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for part in "ab":
    f = glob.glob(f"gpurun_out/ap_{tag}_{part}/**/p_counter_collection.csv", recursive=True)
    if not f: print("no csv", part); continue
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "attn" not in k or "tail" in k or "delta" in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
print("| kernel (all launches of tools/probes/attn_lab.py) | MFMA busy | VALU insts per MFMA | LDS insts per MFMA | LDS conflict cycles / LDS active | waiting for an instruction |")
print("|---|---|---|---|---|---|")
for k, d in sorted(acc.items()):
    g = lambda c: d[c] / max(cnt[(k, c)], 1)
    simd = 1024 * g("GRBM_GUI_ACTIVE") / 8
    print(f"| `{k}` | {g('SQ_VALU_MFMA_BUSY_CYCLES') / simd:.3f} | {g('SQ_INSTS_VALU') / max(g('SQ_INSTS_MFMA'), 1):.1f} | {g('SQ_INSTS_LDS') / max(g('SQ_INSTS_MFMA'), 1):.2f} | "
          f"{g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f} | {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.2f} |")
P
rm -rf gpurun_out/ap_${TAG}_a gpurun_out/ap_${TAG}_b
