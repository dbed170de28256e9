#!/bin/bash
# SMI clock / power samples beside tools/probes/power_soak.py: usage (GPU box, repo root) bash tools/probes/power_trace.sh <out-prefix>
OUT=${1:-gpurun_out/r03_power}
( while true; do echo "t=$(date +%s.%N | cut -c1-13) $(rocm-smi --showpower --showclocks --showtemp --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')"; sleep 0.25; done ) > ${OUT}_smi.txt 2>&1 &
SAMPLER=$!
python tools/probes/power_soak.py 6 > ${OUT}_soak.jsonl 2>&1
kill $SAMPLER
rocm-smi --showpower --showclocks --csv 2>&1 | head -3 > ${OUT}_smi_header.txt
wc -l ${OUT}_smi.txt ${OUT}_soak.jsonl
head -3 ${OUT}_smi.txt | cut -c1-400
