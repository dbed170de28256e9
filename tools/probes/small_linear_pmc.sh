#!/bin/bash
# PMC passes over the small-M Linear kernel (csrc/vit_gemm_sm.hip) on three C2 shapes, f16x3: matrix-pipe busy, VALU / LDS / memory instruction counts, L2 hits /
# misses, HBM bytes -- per launch.  Separate passes per counter group, kernel trace only (no other trace domain).  usage (GPU box): bash tools/probes/small_linear_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r06sm}; O=gpurun_out
CMD="env LAB_EAGER=1 RULE_ONLY=1 SHAPES=enc_fc1,enc_proj,dec_proj python tools/probes/small_linear_lab.py f16x3"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_a -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_${TAG}_b -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_c -o p -- $CMD > /dev/null 2>&1
python - $TAG <<'P' > $O/${TAG}_small_linear_pmc.md
import csv, glob, sys, collections
tag = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for part in "abc":
    f = glob.glob(f"gpurun_out/pmc_{tag}_{part}/**/p_counter_collection.csv", recursive=True)
    if not f: print("no csv for pass", part); continue
    per = collections.defaultdict(float); meta = {}
    for r in csv.DictReader(open(f[0])):
        if "k_linear_sm" not in r["Kernel_Name"]: continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
    for (did, c), v in per.items():
        rows[meta[did]][c].append(v)
print("| kernel (tile blocks, waves, products) | grid threads | counter | mean per launch |\n|---|---|---|---|")
for k, cs in sorted(rows.items()):
    for c, v in sorted(cs.items()):
        print(f"| `{k[0]}` | {k[1]} | {c} | {sum(v) / len(v):.0f} |")
P
cat $O/${TAG}_small_linear_pmc.md | head -70
