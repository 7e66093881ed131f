#!/bin/bash
# experiment batch: stream-K GEMM launches inside the real schedules.  usage (gpurun): bash tools/r3_x2.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { cfg=$1; shift; env "$@" timeout 300 python bench.py --config $cfg --steps ${STEPS:-20} --no-cpu-baseline --no-configs 2>&1 | grep -E "^\{|timed out" | tail -2 | python -c "
import json,sys; L=sys.stdin.read().strip().splitlines(); d=json.loads(L[-1]); print('FALLBACK ' if len(L)>1 else '', '$cfg $*', round(d['ms_per_step'],3), 'ms')"; }
run cfg2 MOGP_SK=0
run cfg2 MOGP_SK=1
run cfg2 MOGP_SK=1 MOGP_SK_FILL=35
run cfg2 MOGP_SK=2
run cfg2 MOGP_SK=2 MOGP_SK_MIN=16
run cfg2 MOGP_SK=3
run cfg2 MOGP_SK=4
STEPS=5
run cfg5 MOGP_SK=0
run cfg5 MOGP_SK=1
run cfg5 MOGP_SK=1 MOGP_SK_FILL=80
run cfg5 MOGP_SK=4
run cfg4 MOGP_SK=0
run cfg4 MOGP_SK=1
run cfg4 MOGP_SK=4
