#!/bin/bash
# SMI clock / power samples while the raster-only bench line runs (is the VALU-bound composite pass at full clock?): usage (GPU box, repo root)
# bash tools/probes/power_trace_raster.sh <out-prefix>
OUT=${1:-gpurun_out/r05_power_raster}
( while true; do echo "t=$(date +%s.%N | cut -c1-13) $(rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')"; sleep 0.2; done ) > ${OUT}_smi.txt 2>&1 &
SAMPLER=$!
python bench.py --no-cpu-baseline --no-train-leg --no-infer-leg --no-stage-legs --no-dropin-leg --steps 3000 --prewarm-seconds 3 2> /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
kill $SAMPLER
awk 'NR % 4 == 0' ${OUT}_smi.txt | cut -c1-120 | tail -25
