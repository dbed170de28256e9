#!/bin/bash
# product-major MFMA order (round 5) vs the accumulator-major order of rounds 1 - 4 (libvit_x_old.so): wgrad lab + the C3 train step, alternating
set -u; TAG=${1:-mo}; O=gpurun_out
python -m pytest tests/test_gpu_vit.py -m gpu -q -x 2>&1 | tail -2
for L in libvit_x_old.so libvit_hip.so; do
  VIT_LIB_NAME=$L MODE=f16x3 python tools/probes/wgrad_lab.py 2>/dev/null | grep shape | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$L', d['shape'], d['us'], d['TF'])"
done
for L in libvit_x_old.so libvit_hip.so libvit_x_old.so libvit_hip.so; do
  VIT_LIB_NAME=$L python tools/bench_train.py --config c3 --scenes 10 --steps 8 --warmup 3 --linear-mode f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'])"
done
