#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/cfg5_err.py 2>&1 | grep -E "induction|weight|scale|loss rel|checksum|bitwise|Error|Exception"
run() { env "$@" python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-shard-probe 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2))"; }
run A=1
run MOGP_SYRK_KS=15
run MOGP_RV_COLS=0
run MOGP_SYRK_KS=15 MOGP_RV_COLS=0
run MOGP_SYRK_KS=-8
run A=1
