#!/bin/bash
# Same-box A/B of the round-5 tree (ab_r05/: `git archive 3b4bea7`, built here, git-ignored) against the working tree, alternating:
# the C3 train step (f16x3, 10 scenes) and the raster headline.  usage (GPU box): bash tools/exp_r05_ab.sh <tag> [reps]
set -u; TAG=${1:-r06}; REPS=${2:-3}; O=$PWD/gpurun_out; mkdir -p $O; rm -f $O/${TAG}_r05_ab.jsonl
ROOT=$PWD
one() {  # $1 = label, $2 = dir
  cd $2
  timeout 900 python tools/bench_train.py --scenes 10 --steps 12 --warmup 3 --linear-mode f16x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'tree': '$1', 'leg': 'c3_train_f16x3', 'ms_per_step': d['ms_per_step']}))" >> $O/${TAG}_r05_ab.jsonl
  timeout 600 python bench.py --no-cpu-baseline --no-train-leg --no-infer-leg --no-stage-legs --no-dropin-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'tree': '$1', 'leg': 'raster_headline', 'ms_per_step': d['ms_per_step'], 'views_per_s': d['value']}))" >> $O/${TAG}_r05_ab.jsonl
  cd $ROOT
}
for rep in $(seq 1 $REPS); do one r05 $ROOT/ab_r05; one r06 $ROOT; done
cat $O/${TAG}_r05_ab.jsonl
