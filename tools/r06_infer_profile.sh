#!/bin/bash
# round 6: C2 inference (batch 1) -- wall time per launch form, then the kernel trace of one step with the stream of every kernel.
# usage (GPU box): bash tools/r06_infer_profile.sh <tag> [mode=f16x3]
set -u
TAG=${1:-r06i}; MODE=${2:-f16x3}
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
export VIT_LINEAR_MODE=$MODE VIT_ATTENTION=$MODE
for F in "" "--streams" "--stream-graphs"; do
  timeout 600 python tools/bench_infer.py --steps 30 --warmup 5 $F 2>$O/${TAG}_infer.err | tail -1 | cut -c1-330
done > $O/${TAG}_infer_$MODE.jsonl
cat $O/${TAG}_infer_$MODE.jsonl
for F in streams stream-graphs; do
  rm -rf /tmp/prof_c2; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -- python tools/bench_infer.py --steps 6 --warmup 3 --$F > /dev/null 2>&1
  DB=$(find /tmp/prof_c2 -name "*.db" | head -1)
  python tools/rocpd_timeline.py $DB "k_preprocess<" $O/${TAG}_c2_${F}_$MODE.md --agg > /dev/null
  python tools/rocpd_streams.py $DB "k_preprocess<" > $O/${TAG}_c2_${F}_${MODE}_streams.txt
  tail -1 $O/${TAG}_c2_${F}_$MODE.md; head -12 $O/${TAG}_c2_${F}_${MODE}_streams.txt
done
echo done
