#!/bin/bash
# repeated e2e runs: how often Gaussians are set aside, whether each one is identified, and what else trips
for i in 1 2 3 4 5 6 7 8; do
  python -m pytest tests/test_e2e_parity.py -m gpu -q -s -k "c3 and (f16x3 or bf16x6)" 2>&1 | grep -E "set aside|passed|failed|^E  " | cut -c1-700
done
