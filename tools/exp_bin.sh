#!/bin/bash
# A/B of the K1 / K3 binning (LDS histograms vs wave-aggregated global atomics, GSR_BIN=ballot) on the headline and the larger workloads (GPU box)
set -u
TAG=${1:-bin}; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_rasterizer.py tests/test_host_boundary.py -m gpu -x -q > $O/${TAG}_raster_tests.log 2>&1
tail -3 $O/${TAG}_raster_tests.log
GSR_BIN=ballot timeout 1500 python -m pytest tests/test_gpu_rasterizer.py -m gpu -x -q > $O/${TAG}_raster_tests_ballot.log 2>&1
tail -1 $O/${TAG}_raster_tests_ballot.log
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
for M in lds ballot; do
  GSR_BIN=$M timeout 600 python bench.py $B > $O/${TAG}_bench_${M}_$rep.json 2> $O/${TAG}_bench_${M}_$rep.err
  show "$M#$rep" $O/${TAG}_bench_${M}_$rep.json
done
done
for M in lds ballot; do
  GSR_BIN=$M timeout 600 python bench.py --ctx 4 $B > $O/${TAG}_bench_c4_${M}.json 2> $O/${TAG}_bench_c4_${M}.err
  show "ctx4 $M" $O/${TAG}_bench_c4_${M}.json
  GSR_BIN=$M timeout 600 python bench.py --ctx 4 --res 512 --sh-degree 4 --scenes 3 $B > $O/${TAG}_bench_c5_${M}.json 2> $O/${TAG}_bench_c5_${M}.err
  show "c5 $M" $O/${TAG}_bench_c5_${M}.json
done
