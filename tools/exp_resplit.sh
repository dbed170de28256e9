#!/bin/bash
# A/B of the batched weight re-split on one box: per-kernel tables of the C3 step with and without it
set -u; TAG=${1:-rs}; O=gpurun_out; export TMPDIR=/tmp
for V in 1 0; do
  STYL3R_RESPLIT=$V python tools/bench_train.py --config c3 --scenes 10 --steps 8 --warmup 3 --linear-mode f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resplit=$V', d['ms_per_step'])"
  STYL3R_RESPLIT=$V rocprofv3 --kernel-trace -d /tmp/prof_rs$V -- python tools/bench_train.py --config c3 --scenes 10 --steps 3 --warmup 2 --linear-mode f16x3 > /dev/null 2>&1
  python tools/rocpd_timeline.py $(find /tmp/prof_rs$V -name "*.db" | head -1) "k_preprocess<" $O/${TAG}_train_step_c3_resplit$V.md --agg > /dev/null
  grep -E "k_split|step span" $O/${TAG}_train_step_c3_resplit$V.md | cut -c1-150
done
