#!/bin/bash
# Round-5 composite-kernel clean-up A/B (GPU box): rasterizer parity tests on the product library, then the raster-only bench line with every
# prebuilt variant library named on the command line (GSR_LIB_NAME / GSR_HIPCC_EXTRA builds made in the build container, e.g. an older commit's
# sources, -DGSR_K6_R10 = ten-value reduction, -DGSR_EXP_E = exp() on the unscaled conic, -DGSR_K5_BRANCHY = the exec-mask-branch forward).  usage: bash tools/exp_k6r5.sh <tag> [lib names ...] -> gpurun_out/<tag>_*
set -u
TAG=${1:-k6r5}; shift; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_rasterizer.py tests/test_host_boundary.py -m gpu -x -q > $O/${TAG}_raster_tests.log 2>&1
tail -3 $O/${TAG}_raster_tests.log
B="--no-cpu-baseline --no-train-leg --no-infer-leg --no-stage-legs --no-dropin-leg"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], "views/s", d["ms_per_step"], "ms/step", {k: v["avg_ms"] for k, v in d["roofline"]["stages"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
for L in libgsr_hip.so "$@"; do
  GSR_LIB_NAME=$L timeout 600 python bench.py $B > $O/${TAG}_bench_${L}_$rep.json 2> $O/${TAG}_bench_${L}_$rep.err
  show "$L#$rep" $O/${TAG}_bench_${L}_$rep.json
done
done
