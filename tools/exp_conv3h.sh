#!/bin/bash
# A/B of the halo convolution's fragment prefetch (round 5): convolution parity tests on the product library, then tools/probes/conv3_halo.py
# with the product and with a prebuilt older library (VIT_LIB_NAME).  usage (GPU box): bash tools/exp_conv3h.sh <tag> [old lib name]
set -u
TAG=${1:-c3h}; OLD=${2:-libvit_old.so}; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_vit.py -m gpu -x -q -k "conv or halo" > $O/${TAG}_conv_tests.log 2>&1
tail -3 $O/${TAG}_conv_tests.log
python tools/probes/conv3_halo.py 20 > $O/${TAG}_new.jsonl 2> $O/${TAG}_new.err
VIT_LIB_NAME=$OLD python tools/probes/conv3_halo.py 20 > $O/${TAG}_old.jsonl 2> $O/${TAG}_old.err
python - $O/${TAG}_new.jsonl $O/${TAG}_old.jsonl <<'PY'
import json, sys
new = [json.loads(l) for l in open(sys.argv[1])]; old = [json.loads(l) for l in open(sys.argv[2])]
for n, o in zip(new, old):
    print(n["layer"], {m: (n[m]["us"], o[m]["us"], round(o[m]["us"] / n[m]["us"], 3), n[m]["TF"]) for m in ("bf16x6", "bf16x3", "f16x3")})
PY
