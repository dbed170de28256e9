#!/bin/bash
# usage (GPU box): tools/exp_run.sh <libname>...   -- stage times of prebuilt kernel-experiment libraries (GSR_LIB_NAME builds)
for L in "$@"; do
  GSR_LIB_NAME=$L python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-leg 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'views/s', d['ms_per_step'], 'ms/step', {k:v['avg_ms'] for k,v in d['roofline']['stages'].items()})"
done
