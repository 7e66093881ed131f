#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (boxes differ by a few percent): tools/ab/libA.so, tools/ab/libB.so, alternated.
# usage (gpurun): bash tools/ab_libs.sh <config> <steps> <rounds>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; CFG=${1:-cfg2}; STEPS=${2:-20}; R=${3:-3}
for r in $(seq $R); do for v in A B; do cp tools/ab/lib$v.so mogptk_amd/csrc/libmogp_hip.so
  timeout 300 python bench.py --config $CFG --steps $STEPS --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG lib$v', round(d['ms_per_step'],3))"; done; done
