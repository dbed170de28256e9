cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/prof_c4 -- python tools/bench_train.py --config c4 --scenes 6 --steps 3 --warmup 2 --linear-mode bf16x3 > /dev/null 2>&1
python tools/rocpd_timeline.py $(find /tmp/prof_c4 -name "*.db" | head -1) "k_preprocess<" gpurun_out/r03_train_step_c4_b6_bf16x3.md --agg > /dev/null
head -40 gpurun_out/r03_train_step_c4_b6_bf16x3.md | cut -c1-150; tail -1 gpurun_out/r03_train_step_c4_b6_bf16x3.md
