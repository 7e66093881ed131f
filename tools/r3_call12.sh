#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/c12; mkdir -p $O
AB=MOGP_PAIR_ACC:0,1 timeout 600 python tools/chain_check.py 600,1100,2048,4097,6000,8192 > $O/pa_check.txt 2>&1; cat $O/pa_check.txt | tail -7
for v in 0 1; do MOGP_PAIR_ACC=$v timeout 300 python bench.py --no-cpu-baseline --no-configs > $O/bench_p$v.json 2> $O/bench_p$v.err; python -c "
import json; d=json.loads(open('$O/bench_p$v.json').read().strip().splitlines()[-1]); print('MOGP_PAIR_ACC=$v', round(d['value'],2), 'evals/s', round(d['ms_per_step'],3), 'ms')"; done
for v in 0 1; do MOGP_SWEEP_MASKED=$v timeout 600 python bench.py --config cfg3 --mode sharded --steps 3 --warmup 1 > $O/cfg3_sh_m$v.json 2> $O/cfg3_sh_m$v.err; python -c "
import json; d=json.loads(open('$O/cfg3_sh_m$v.json').read().strip().splitlines()[-1]); print('cfg3 sharded 1 rank MOGP_SWEEP_MASKED=$v', round(d['ms_per_step'],1), 'ms')"; done
timeout 900 python bench.py --shard-probe --probes cfg3,cfg2 --no-cpu-baseline --no-configs > $O/bench_probe.json 2> $O/bench_probe.err; python -c "
import json; d=json.loads(open('$O/bench_probe.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('sharded'), indent=1)[:2500])"
