#!/bin/bash
# VERDICT r05 #1, the premise: how much faster would the f16x3 GEMM-shaped kernels run if their fp32 operands arrived already split?
# Builds libvit_nosplit.so (-DVIT_EXP_NOSPLIT) HERE (no GPU needed), then on the GPU box: bash tools/exp_nosplit.sh run <tag>
set -u
if [ "${1:-}" = "build" ]; then
  VIT_LIB_NAME=libvit_nosplit.so VIT_HIPCC_EXTRA="-DVIT_EXP_NOSPLIT" python -c "from styl3r_amd import vit_ops; print(vit_ops.build_library(verbose=False))"
  exit 0
fi
TAG=${2:-r06}; O=gpurun_out; mkdir -p $O; rm -f $O/${TAG}_nosplit_lab.jsonl
for rep in 1 2; do
  python tools/probes/nosplit_lab.py 2>/dev/null | grep '"lib"' >> $O/${TAG}_nosplit_lab.jsonl
  VIT_LIB_NAME=libvit_nosplit.so python tools/probes/nosplit_lab.py 2>/dev/null | grep '"lib"' >> $O/${TAG}_nosplit_lab.jsonl
done
python - <<PY
import json, collections
t = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open("$O/${TAG}_nosplit_lab.jsonl"):
    r = json.loads(l); t[(r["kernel"], r["shape"])][r["lib"]].append(r["us"])
print("| kernel | shape | product us | no-split us | speed-up |\n|---|---|---|---|---|")
for (k, sh), v in t.items():
    a, b = min(v["product"]), min(v["libvit_nosplit.so"])
    print(f"| {k} | {sh} | {a:.1f} | {b:.1f} | {a / b:.3f} |")
PY
