#!/bin/bash
# wgrad A/B (GPU box): parity tests that touch the weight gradient, then tools/probes/wgrad_lab.py per mode with the old and the new kernel
set -u; TAG=${1:-wg}; O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_vit.py -m gpu -q -x -k "linear or wgrad or Linear or fused or ragged or f16x3" 2>&1 | tail -4
for MODE in f16x3 bf16x6; do
  for OLD in 1 ""; do
    VIT_WGRAD_OLD=$OLD MODE=$MODE python tools/probes/wgrad_lab.py 2>/dev/null | grep shape >> $O/${TAG}_wgrad_lab.jsonl
  done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$O/${TAG}_wgrad_lab.jsonl")]
t = collections.defaultdict(dict)
for r in rows: t[(r["mode"], r["shape"])][r["kernel"]] = (r["us"], r["TF"])
for k, v in t.items(): print(k, v, "speedup %.2f" % (v["old"][0] / v["x6t"][0]) if "old" in v and "x6t" in v else "")
PY
