#!/bin/bash
# PMC passes over the GEMM lab (qkv shape only): LDS conflicts, MFMA busy, waits -- per kernel.  usage: tools/probes/gemm_pmc.sh <cfgs> <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CFGS=$1; TAG=$2
CMD="env RES=0 python tools/probes/gemm_lab.py $CFGS enc_qkv"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/gp_${TAG}_a -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/gp_${TAG}_b -o p -- $CMD > /dev/null 2>&1
python - <<'P' $TAG
import csv, glob, sys, collections
tag = sys.argv[1]
for part in "ab":
    f = glob.glob(f"gpurun_out/gp_{tag}_{part}/**/p_counter_collection.csv", recursive=True)
    if not f: print("no csv", part); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        if "linear_x6" not in k: continue
        if r.get("Grid_Size") and False: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k)
        for c, v in d.items(): print(f"    {c:28s} {v / cnt[(k, c)]:16.0f} per launch")
P
