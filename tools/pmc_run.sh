#!/bin/bash
# PMC passes over a short bench run (counters in their own runs: no --stats / trace domains besides kernel-trace), then the
# per-kernel summary and the stamped profiles/pmc_latest.json that bench.py reads.
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <tag>      -> gpurun_out/<tag>_pmc.{txt,json}, gpurun_out/pmc_latest.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-infer-leg --no-dropin-leg --no-stage-legs"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${TAG}_sq -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${TAG}_lds -o p -- $CMD > /dev/null 2>&1
DIRS=$(for k in sq fetch write lds; do find gpurun_out/pmc_${TAG}_$k -name "p_counter_collection.csv" -printf "%h\n" | head -1; done)
python tools/pmc_summary.py $DIRS --json gpurun_out/${TAG}_pmc.json > gpurun_out/${TAG}_pmc.txt
PAIRS=$(python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-infer-leg --no-dropin-leg --no-stage-legs | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['pairs_R'])")
python tools/pmc_latest.py gpurun_out/${TAG}_pmc.json profiles/${TAG}_pmc.json $PAIRS gpurun_out/pmc_latest.json
rm -rf gpurun_out/pmc_${TAG}_sq gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_lds
