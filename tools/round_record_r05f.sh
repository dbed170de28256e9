#!/bin/bash
# Round-5 final record after the composite-kernel clean-up (the encoder kernels did not change since r05z): raster record (kernel stats,
# timeline, PMC stamped with the running build, the two driver-form bench lines), e2e parity tables, the C3 / C4 train-step tables in the
# headline arithmetic, the rasterizer size sweep.  usage (GPU box): bash tools/round_record_r05f.sh <tag>  -> gpurun_out/<tag>_*
set -u
TAG=${1:-r05f}; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
bash tools/round_record_raster.sh $TAG
python -m pytest tests/test_e2e_parity.py -m gpu -q -s 2>&1 | grep "^\s*\[c\|^\s*\[full\|^.\s*\[\|b=2,v=4\|passed\|failed" > $O/${TAG}_e2e_parity_tables.txt
for CFG in "c3 10" "c4 6"; do
  set -- $CFG
  rocprofv3 --kernel-trace -d /tmp/prof_$1 -- python tools/bench_train.py --config $1 --scenes $2 --steps 3 --warmup 2 --linear-mode f16x3 > /dev/null 2>&1
  python tools/rocpd_timeline.py $(find /tmp/prof_$1 -name "*.db" | head -1) "k_preprocess<" $O/${TAG}_train_step_$1_b$2_f16x3.md --agg > /dev/null
done
B="--no-cpu-baseline --no-train-leg --no-infer-leg"
{ python bench.py --grid 128 $B | tail -1; python bench.py --ctx 2 $B | tail -1; python bench.py --ctx 4 $B | tail -1;
  python bench.py --ctx 4 --res 512 --sh-degree 4 --scenes 3 $B | tail -1; python bench.py --grid 512 --ctx 4 --res 512 --scenes 2 $B | tail -1;
  python tools/bench_train.py --config c3 --scenes 10 --steps 10 --warmup 3 --linear-mode f16x3 | tail -1;
  python tools/bench_train.py --config c4 --scenes 6 --steps 10 --warmup 3 --linear-mode f16x3 | tail -1;
  python tools/bench_train.py --config c5 --scenes 1 --steps 10 --warmup 3 --linear-mode f16x3 | tail -1; } > $O/${TAG}_sweep.jsonl 2> $O/${TAG}_sweep.err
echo record done
