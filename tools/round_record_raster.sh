#!/bin/bash
# the raster part of the round record alone (kernel stats, step timeline, PMC passes stamped with the running build), then the two
# driver-form bench lines WITH the fresh pmc_latest.json in place.  usage (GPU box): bash tools/round_record_raster.sh <tag>
set -u
TAG=${1:-r05}; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python bench.py --no-cpu-baseline --no-train-leg --no-infer-leg --no-dropin-leg > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_bench -name "*.db" | head -1) $O/${TAG}_kernel_stats.md > /dev/null
python tools/rocpd_timeline.py $(find /tmp/prof_bench -name "*.db" | head -1) "k_build_views" $O/${TAG}_step_timeline.md > /dev/null
bash tools/pmc_run.sh ${TAG}
cp $O/pmc_latest.json profiles/pmc_latest.json
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-infer-leg > $O/${TAG}_bench_torchrun1.json 2> $O/${TAG}_bench_torchrun1.err
echo done
