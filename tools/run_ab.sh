cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for c in cfg3 cfg4; do for v in 4 6 8; do
MOGP_OUTER=$v timeout 300 python bench.py --config $c --no-cpu-baseline --no-shard-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c outer=$v', round(d['ms_per_step'],2), 'ms frac', round(d['roofline']['frac'],3), {k:round(v,1) for k,v in d.get('stages_ms_per_eval',{}).items() if k in ('potrf','trtri','lauum')})"
done; done > gpurun_out/ab/ab.log 2>&1
cat gpurun_out/ab/ab.log
