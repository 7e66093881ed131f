#!/bin/bash
# A/B of one environment switch on the GPU box: parity of the two settings (tools/chain_check.py), then bench.py under each.
# usage (through gpurun): bash tools/r3_ab.sh VAR v0 v1 [sizes]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; VAR=$1; A=$2; B=$3; SIZES=${4:-600,1100,2048,4097,6000,8192}
AB=$VAR:$A,$B timeout 900 python tools/chain_check.py $SIZES 2>&1 | grep -E "^N=|tile map|^   "
for v in $A $B; do env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['value'],2), 'evals/s', round(d['ms_per_step'],3), 'ms')"; done
