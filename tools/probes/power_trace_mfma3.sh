#!/bin/bash
# SMI clock / power samples beside tools/probes/power_soak_mfma3.py: usage (GPU box, repo root) bash tools/probes/power_trace_mfma3.sh <out-prefix>
OUT=${1:-gpurun_out/r05_power_mfma3}
( while true; do echo "t=$(date +%s.%N | cut -c1-13) $(rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')"; sleep 0.25; done ) > ${OUT}_smi.txt 2>&1 &
SAMPLER=$!
python tools/probes/power_soak_mfma3.py 4 > ${OUT}_soak.jsonl 2> ${OUT}_soak.err
kill $SAMPLER
cat ${OUT}_soak.jsonl; tail -2 ${OUT}_soak.err
rocm-smi --showpower --showclocks --csv 2>&1 | head -2 | cut -c1-300
python - ${OUT} <<'PY'
import sys, json, re
pre = sys.argv[1]
soak = [json.loads(l) for l in open(pre + "_soak.jsonl")]
smi = []
for l in open(pre + "_smi.txt"):
    m = re.match(r"t=([\d.]+) (.*)", l)
    if m: smi.append((float(m.group(1)), m.group(2)))
t0 = None
for i, s in enumerate(soak):
    lo = soak[i - 1]["t"] if i else s["t"] - 4.0
    rows = [r for t, r in smi if lo + 1.0 <= t <= s["t"]]
    print(s["kernel"], s["TF_median"], "TF |", rows[len(rows) // 2][:160] if rows else "no smi sample")
PY
