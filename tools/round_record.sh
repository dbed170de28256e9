#!/bin/bash
# One call that regenerates the round's record on an MI355X box: driver-shaped bench line (direct and under a one-rank
# torch.distributed.run = RCCL path), rocprofv3 kernel stats of the same command, the PMC passes (pmc_latest.json), the per-step kernel
# table of the C3 train step in both arithmetic modes, the e2e parity tables, every benchmark of DESIGN.md 6 / 8.
# usage (GPU box, repo root): bash tools/round_record.sh <tag>        -> gpurun_out/<tag>_*
set -u
TAG=${1:-r05}; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-infer-leg > $O/${TAG}_bench_torchrun1.json 2> $O/${TAG}_bench_torchrun1.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python bench.py --no-cpu-baseline --no-train-leg --no-infer-leg --no-dropin-leg > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_bench -name "*.db" | head -1) $O/${TAG}_kernel_stats.md > /dev/null
python tools/rocpd_timeline.py $(find /tmp/prof_bench -name "*.db" | head -1) "k_build_views" $O/${TAG}_step_timeline.md > /dev/null
bash tools/pmc_run.sh ${TAG}
for MODE in f16x3 bf16x3 bf16x6; do
  rocprofv3 --kernel-trace -d /tmp/prof_c3_$MODE -- python tools/bench_train.py --config c3 --scenes 10 --steps 3 --warmup 2 --linear-mode $MODE > /dev/null 2>&1
  python tools/rocpd_timeline.py $(find /tmp/prof_c3_$MODE -name "*.db" | head -1) "k_preprocess<" $O/${TAG}_train_step_c3_b10_$MODE.md --agg > /dev/null
done
# the other two train configurations: per-kernel tables of the C4 style-stage step (6 scenes) and the C5 stress step, headline arithmetic
rocprofv3 --kernel-trace -d /tmp/prof_c4 -- python tools/bench_train.py --config c4 --scenes 6 --steps 3 --warmup 2 --linear-mode f16x3 > /dev/null 2>&1
python tools/rocpd_timeline.py $(find /tmp/prof_c4 -name "*.db" | head -1) "k_preprocess<" $O/${TAG}_train_step_c4_b6_f16x3.md --agg > /dev/null
rocprofv3 --kernel-trace -d /tmp/prof_c5 -- python tools/bench_train.py --config c5 --scenes 1 --steps 3 --warmup 2 --linear-mode f16x3 > /dev/null 2>&1
python tools/rocpd_timeline.py $(find /tmp/prof_c5 -name "*.db" | head -1) "k_preprocess<" $O/${TAG}_train_step_c5_b1_f16x3.md --agg > /dev/null
python -m pytest tests/test_e2e_parity.py -m gpu -q -s 2>&1 | grep "^\s*\[c\|^\s*\[full\|^.\s*\[\|b=2,v=4\|passed\|failed" > $O/${TAG}_e2e_parity_tables.txt
bash tools/run_all_benchmarks.sh $O/${TAG}_benchmarks > $O/${TAG}_benchmarks.log 2>&1
echo done
