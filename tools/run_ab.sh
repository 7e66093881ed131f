cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for i in 1 2 3; do
for v in 0 1; do
MOGP_ACC_STREAM=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('acc_stream=$v', round(d['ms_per_step'],3), 'ms', 'potrf', round(d['stages_ms_per_eval']['potrf'],2))"
done; done > gpurun_out/ab/ab.log 2>&1
cat gpurun_out/ab/ab.log
