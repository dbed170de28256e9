#!/bin/bash
# usage: tools/exp_bench.sh "<hipcc extra flags>" <libname>   -- builds a kernel-experiment variant and prints stage times
export GSR_HIPCC_EXTRA="$1" GSR_LIB_NAME="$2"
python -c "from styl3r_amd import _lib; _lib.build_library(force=True)" && \
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], 'views/s', d['ms_per_step'], 'ms/step'); print({k:v['avg_ms'] for k,v in d['roofline']['stages'].items()})"
