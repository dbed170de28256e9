#!/bin/bash
# round 6: C3 train step (10 scenes) -- wall time without the profiler, then the per-kernel table and the idle-gap report of one step.
# usage (GPU box): bash tools/r06_train_profile.sh <tag> [mode=f16x3] [extra bench_train args]
set -u
TAG=${1:-r06t}; MODE=${2:-f16x3}; shift; shift || true
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/bench_train.py --scenes 10 --steps 6 --warmup 3 --linear-mode $MODE "$@" > $O/${TAG}_train_c3_$MODE.json 2> $O/${TAG}_train_c3_$MODE.err
cat $O/${TAG}_train_c3_$MODE.json | cut -c1-400
rm -rf /tmp/prof_c3; timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -- python tools/bench_train.py --scenes 10 --steps 3 --warmup 2 --linear-mode $MODE "$@" > /dev/null 2>&1
DB=$(find /tmp/prof_c3 -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB "k_preprocess<" $O/${TAG}_train_step_c3_$MODE.md --agg > /dev/null
python tools/rocpd_gaps.py $DB "k_preprocess<" 30 > $O/${TAG}_train_gaps_c3_$MODE.txt
head -8 $O/${TAG}_train_gaps_c3_$MODE.txt; tail -1 $O/${TAG}_train_step_c3_$MODE.md
echo done
