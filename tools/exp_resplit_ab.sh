for V in 0 1 0 1 0 1; do
  STYL3R_RESPLIT=$V python tools/bench_train.py --config c3 --scenes 10 --steps 8 --warmup 3 --linear-mode f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resplit=$V', d['ms_per_step'])"
done
