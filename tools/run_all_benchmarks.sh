#!/bin/bash
# Reproduces every number quoted in DESIGN.md section 6 / 8 on one MI355X (run from the repo root, ~6 minutes).
# usage: bash tools/run_all_benchmarks.sh [outdir]      (default: gpurun_out/benchmarks)
set -u
OUT=${1:-gpurun_out/benchmarks}; mkdir -p "$OUT"
run() { echo "== $*"; "$@" 2>&1 | tail -1 | tee -a "$OUT/all.jsonl" | cut -c1-260; }
run python bench.py                                                              # headline M1 (+ roofline, CPU baseline)
run python bench.py --grid 128 --no-cpu-baseline --no-train-leg --no-infer-leg                                 # 16 384 Gaussians / scene
run python bench.py --ctx 2 --no-cpu-baseline --no-train-leg --no-infer-leg                                    # 131 072
run python bench.py --ctx 4 --no-cpu-baseline --no-train-leg --no-infer-leg                                    # 262 144
run python bench.py --ctx 4 --res 512 --sh-degree 4 --scenes 3 --no-cpu-baseline --no-train-leg --no-infer-leg # C5 stress shapes
run python bench.py --grid 512 --ctx 4 --res 512 --scenes 2 --no-cpu-baseline --no-train-leg --no-infer-leg    # 1 048 576
for MODE in f16x3 bf16x6 bf16x3; do       # M2, C3 (f16x3: the headline arithmetic; bf16x6: the library default; bf16x3: the reference's TF32 class)
  run python tools/bench_train.py --config c3 --scenes 10 --steps 10 --warmup 3 --linear-mode $MODE
done
run python tools/bench_train.py --config c3 --scenes 8 --steps 10 --warmup 3 --linear-mode f16x3
for MODE in f16x3 bf16x6 bf16x3; do       # style stage at the reference's batch; 512^2 / sh 4 stress step (1 and 3 scenes: the reference's per-GPU batch)
  run python tools/bench_train.py --config c4 --scenes 6 --steps 10 --warmup 3 --linear-mode $MODE
  run python tools/bench_train.py --config c5 --scenes 1 --steps 10 --warmup 3 --linear-mode $MODE
done
run python tools/bench_train.py --config c5 --scenes 3 --steps 5 --warmup 2 --linear-mode f16x3
run env STYL3R_DP_MODE=rs_ag python tools/bench_train.py --config c3 --scenes 10 --steps 10 --warmup 3 --linear-mode f16x3   # the rs_ag exchange mode's data layout on one GPU (flat parameter buckets, owned-range AdamW; no process group: no collectives)
run python tools/bench_infer.py                                                  # C2 inference
run python tools/bench_infer.py --streams                                        # C2 inference, style branch + decoder 2 + heads on side streams
run python tools/bench_infer.py --stream-graphs                                  # C2 inference, one hipGraph per stream segment
run python tools/bench_vit.py                                                    # kernel microbenchmarks
echo "results: $OUT/all.jsonl"
