#!/bin/bash
# the full GPU suite N times (no -x): flaky bars show up here before they show up on the driver's box
N=${1:-2}; TAG=${2:-stress}; O=gpurun_out
for i in $(seq 1 $N); do
  python -m pytest tests -m gpu -q -p no:cacheprovider > $O/${TAG}_suite_$i.log 2>&1
  grep -E "^FAILED|passed|failed" $O/${TAG}_suite_$i.log | tail -5
done
