#!/bin/bash
# A/B of the composite-backward kernels on the headline workload (GPU box): parity tests of all, then the raster-only bench line with each.
# usage: bash tools/exp_k6.sh <tag> [K6=lib pairs, e.g. rows16=libgsr_x_noatom.so ...]  -> gpurun_out/<tag>_*
set -u
TAG=${1:-k6}; shift; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_rasterizer.py -m gpu -x -q > $O/${TAG}_raster_tests.log 2>&1
tail -3 $O/${TAG}_raster_tests.log
B="--no-cpu-baseline --no-train-leg --no-infer-leg --no-stage-legs"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], "views/s", d["ms_per_step"], "ms/step", {k: v["avg_ms"] for k, v in d["roofline"]["stages"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for K in tile rows16 rows8; do
  GSR_K6=$K timeout 600 python bench.py $B > $O/${TAG}_bench_$K.json 2> $O/${TAG}_bench_$K.err
  show $K $O/${TAG}_bench_$K.json
done
for KL in "$@"; do
  K=${KL%%=*}; LIB=${KL##*=}
  GSR_K6=$K GSR_LIB_NAME=$LIB timeout 600 python bench.py $B > $O/${TAG}_bench_${K}_$LIB.json 2> $O/${TAG}_bench_${K}_$LIB.err
  show "$K/$LIB" $O/${TAG}_bench_${K}_$LIB.json
done
