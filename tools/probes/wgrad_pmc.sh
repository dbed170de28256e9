#!/bin/bash
# PMC passes over the weight-gradient microbench (counters only, kernel-trace domain): usage tools/probes/wgrad_pmc.sh [PRODUCTS]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PRODUCTS=${1:-3} ONLY=enc_fc1
CMD="python tools/probes/wgrad_lab.py"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/wgpmc_$i -o p -- $CMD > /dev/null 2>&1
done
python - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob("/tmp/wgpmc_*"):
    per = collections.defaultdict(float); names = {}
    for f in glob.glob(d + "/**/p_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"]); names[row["Dispatch_Id"]] = (row["Kernel_Name"], row.get("Grid_Size", ""))
    for (did, c), v in per.items():
        kn, gs = names[did]
        if "k_wgrad_x6" in kn: agg[kn.split("(")[0][-30:] + " grid " + gs][c].append(v)
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
