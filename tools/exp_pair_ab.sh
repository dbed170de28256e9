#!/bin/bash
# same-box A/B of the round-6 weight-image path: VIT_PAIR_SPLIT=0 (one launch per image, rounds 2-5) vs 1 (pair kernels), C3 step, alternating
set -u; TAG=${1:-r06}; O=gpurun_out; mkdir -p $O; rm -f $O/${TAG}_pair_ab.jsonl
for rep in 1 2 3; do
  for P in 0 1; do
    VIT_PAIR_SPLIT=$P timeout 600 python tools/bench_train.py --scenes 10 --steps 16 --warmup 3 --linear-mode f16x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'pair': $P, 'ms_per_step': d['ms_per_step']}))" >> $O/${TAG}_pair_ab.jsonl
  done
done
cat $O/${TAG}_pair_ab.jsonl
