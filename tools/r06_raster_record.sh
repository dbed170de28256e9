#!/bin/bash
# round 6 raster record: kernel stats + step timeline of the headline command, the PMC passes stamped with the running build (pmc_latest.json),
# then the headline line WITH the fresh counters in place.  usage (GPU box): bash tools/r06_raster_record.sh <tag>
set -u; TAG=${1:-r06}; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python bench.py --no-cpu-baseline --no-train-leg --no-infer-leg --no-dropin-leg --no-stage-legs > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/prof_bench -name "*.db" | head -1) $O/${TAG}_kernel_stats.md > /dev/null
python tools/rocpd_timeline.py $(find /tmp/prof_bench -name "*.db" | head -1) "k_build_views" $O/${TAG}_step_timeline.md > /dev/null
bash tools/pmc_run.sh ${TAG}
cp $O/pmc_latest.json profiles/pmc_latest.json
python bench.py --no-train-leg --no-infer-leg --no-stage-legs > $O/${TAG}_bench_raster.json 2> $O/${TAG}_bench_raster.err
tail -c 600 $O/${TAG}_bench_raster.json; echo; echo done
