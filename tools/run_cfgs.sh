cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cfgs
for c in cfg2 cfg3 cfg4 cfg5; do
timeout 300 python bench.py --config $c --no-cpu-baseline --no-shard-probe 2>gpurun_out/cfgs/$c.err | tail -1 > gpurun_out/cfgs/$c.json
python -c "
import json; d=json.loads(open('gpurun_out/cfgs/$c.json').read()); print('$c', round(d['ms_per_step'],2),'ms  frac',round(d['roofline']['frac'],3), {k:round(v,2) for k,v in d.get('stages_ms_per_eval',{}).items()})"
done
