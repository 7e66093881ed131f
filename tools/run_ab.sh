cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "titsias or cfg5 or checkpoints" 2>&1 | tail -3 > gpurun_out/ab/ab.log
for i in 1 2; do timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-shard-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', round(d['ms_per_step'],2), 'ms frac', round(d['roofline']['frac'],3))"; done >> gpurun_out/ab/ab.log 2>&1
cat gpurun_out/ab/ab.log
