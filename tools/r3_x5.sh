#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
AB=MOGP_FULL_INVERSE:1,0 timeout 900 python tools/chain_check.py 600,2048,4097,6000,8192 2>&1 | grep -E "^N=|tile map|^   "
for v in 1 0 1 0; do MOGP_FULL_INVERSE=$v timeout 300 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 MOGP_FULL_INVERSE=$v', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'evals/s')"; done
for v in 1 0; do MOGP_FULL_INVERSE=$v timeout 300 python bench.py --config cfg3 --steps 3 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 MOGP_FULL_INVERSE=$v', round(d['ms_per_step'],3), 'ms')"; done
