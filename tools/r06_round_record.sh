#!/bin/bash
# round 6 record (encoder side + the driver-form lines): bench.py as the driver runs it, the same under a one-rank torch.distributed.run (RCCL
# path), per-kernel tables of the C3 / C4 / C5 train steps, observed parity distances, the e2e parity tables, every benchmark of DESIGN 6 / 8.
# (raster kernel stats / timeline / PMC: tools/r06_raster_record.sh)   usage (GPU box): bash tools/r06_round_record.sh <tag>
set -u; TAG=${1:-r06z}; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-infer-leg --no-stage-legs > $O/${TAG}_bench_torchrun1.json 2> $O/${TAG}_bench_torchrun1.err
for CFG in "c3 10" "c4 6" "c5 1"; do set -- $CFG
  rm -rf /tmp/prof_t; rocprofv3 --kernel-trace -d /tmp/prof_t -- python tools/bench_train.py --config $1 --scenes $2 --steps 3 --warmup 2 --linear-mode f16x3 > /dev/null 2>&1
  python tools/rocpd_timeline.py $(find /tmp/prof_t -name "*.db" | head -1) "k_preprocess<" $O/${TAG}_train_step_$1_b$2_f16x3.md --agg > /dev/null
done
PARITY_VERBOSE=1 python -m pytest tests/test_encoder.py tests/test_encoder_mid.py tests/test_backbone_variants.py tests/test_head_variants.py tests/test_gpu_adapter.py tests/test_gpu_vit.py -m gpu -q -s 2>&1 | grep "\[parity\]\|attention error\|passed\|failed" > $O/${TAG}_parity_observed.txt
python -m pytest tests/test_e2e_parity.py -m gpu -q -s 2>&1 | grep "^\s*\[c\|^\s*\[full\|^.\s*\[\|b=2,v=4\|passed\|failed" > $O/${TAG}_e2e_parity_tables.txt
bash tools/run_all_benchmarks.sh $O/${TAG}_benchmarks > $O/${TAG}_benchmarks.log 2>&1
echo done
