#!/bin/bash
# round 6: rasterizer suite + short bench + step timeline on the GPU box.  usage: bash tools/r06_raster_check.sh <tag>
set -u
TAG=${1:-r06a}; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_rasterizer.py tests/test_gpu_properties.py tests/test_gpu_losses.py tests/test_pose_align.py tests/test_host_boundary.py -m gpu -x -q > $O/${TAG}_raster_tests.log 2>&1
tail -5 $O/${TAG}_raster_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-train-leg --no-infer-leg --no-stage-legs > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 1500 $O/${TAG}_bench.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python bench.py --no-cpu-baseline --no-train-leg --no-infer-leg --no-dropin-leg --no-stage-legs > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_bench -name "*.db" | head -1) $O/${TAG}_kernel_stats.md > /dev/null
python tools/rocpd_timeline.py $(find /tmp/prof_bench -name "*.db" | head -1) "k_build_views" $O/${TAG}_step_timeline.md
echo done
