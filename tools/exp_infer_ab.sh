#!/bin/bash
# same-box A/B of the C2 inference leg (bench.py `infer`), round-5 tree vs working tree.  usage: bash tools/exp_infer_ab.sh <tag>
set -u; TAG=${1:-r06}; O=$PWD/gpurun_out; mkdir -p $O; ROOT=$PWD; rm -f $O/${TAG}_infer_ab.jsonl
for rep in 1 2; do for T in r05:$ROOT/ab_r05 r06:$ROOT; do
  cd ${T#*:}
  timeout 900 python bench.py --steps 5 --warmup 2 --prewarm-seconds 0.2 --no-cpu-baseline --no-train-leg --no-stage-legs --no-dropin-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'tree': '${T%%:*}', 'infer': d.get('infer')}))" >> $O/${TAG}_infer_ab.jsonl
  cd $ROOT
done; done
python - <<PY
import json
for l in open("$O/${TAG}_infer_ab.jsonl"):
    d = json.loads(l); i = d["infer"]
    print(d["tree"], {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if "ms" in kk}) for k, v in i.items() if k in ("total_ms", "encoder_ms", "rasterizer_ms", "stream_graphs_total_ms", "modes", "f16x3", "bf16x3", "arithmetic", "align_pose")})
PY
