set -u; O=gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_backbone_variants.py -m gpu -q 2>&1 | tail -5
PARITY_VERBOSE=1 python -m pytest tests/test_encoder.py tests/test_encoder_mid.py tests/test_backbone_variants.py -m gpu -q -s 2>&1 | grep -E "parity\]|passed|failed" > $O/r05k_parity_encoder.txt; tail -3 $O/r05k_parity_encoder.txt
bash tools/probes/fetch_gather_calib.sh $O/r05k_fetch_calib.md > /dev/null 2>&1; cat $O/r05k_fetch_calib.md
python bench.py --no-train-leg --no-infer-leg --no-cpu-baseline > $O/r05k_bench_raster.json 2> $O/r05k_bench_raster.err; python -c "
import json; d=json.loads(open('$O/r05k_bench_raster.json').read().strip().splitlines()[-1]); print(d['value'], d.get('dropin_per_view'))"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-infer-leg --no-stage-legs --no-cpu-baseline --train-steps 4 > $O/r05k_bench_torchrun1.json 2> $O/r05k_bench_torchrun1.err; python -c "
import json; d=json.loads(open('$O/r05k_bench_torchrun1.json').read().strip().splitlines()[-1]); t=d['train_step']; print(t.get('ms_per_step'), t.get('comm'), t.get('dp_modes'), t.get('comm_error'))"
