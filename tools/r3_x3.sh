#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2; do timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_eval']; print('cfg2', round(d['ms_per_step'],3), 'moment_kernel', round(s['moment_kernel'],4), 'gram_kernel', round(s['gram_kernel'],4))"; done
timeout 300 python bench.py --config cfg5 --steps 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', round(d['ms_per_step'],3))"
timeout 300 python bench.py --config cfg3 --steps 3 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_eval']; print('cfg3', round(d['ms_per_step'],3), 'moment_kernel', round(s['moment_kernel'],4))"
