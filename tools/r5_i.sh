#!/bin/bash
# round 5: the moment kernel keeping its sums in registers across the tiles of a channel pair
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "lml or full_size or device_raw or cfg3 or only_the_needed or dataflow or sharing_one_gpu" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
b() { timeout 300 python bench.py --config $1 --steps $2 --warmup 3 --no-cpu-baseline --no-configs --sustained 0 2>>$O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('stages_ms_per_eval',{}); print('$3', round(d['ms_per_step'],3), 'moments stage', s.get('moments'), 'kernel', s.get('moment_kernel'), 'frac', d.get('moments_hbm',{}).get('frac'))"; }
b cfg2 60 "cfg2"
b cfg2 60 "cfg2"
b cfg3 3 "cfg3"
