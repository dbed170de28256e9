#!/bin/bash
# repeated e2e runs: how often Gaussians are set aside, whether each one is identified, and what else trips.  usage: exp_e2e_flips.sh [-k expression] [runs]
K=${1:-"c3 and (f16x3 or bf16x6)"}; N=${2:-8}
for i in $(seq 1 $N); do
  python -m pytest tests/test_e2e_parity.py -m gpu -q -s -k "$K" 2>&1 | grep -E "set aside|passed|failed|^E  " | cut -c1-700
done
