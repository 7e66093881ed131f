#!/bin/bash
# round 5: diagonal tiles inside the strip Gram kernel; K_uu's factorisation chain on the private stream (sparse bound)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "gram or lml or full_size or titsias or cfg5 or snelson or hensman or predict or dataflow or device_raw" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
b() { timeout 300 python bench.py --config $1 --steps $2 --warmup 3 --no-cpu-baseline --no-configs --sustained 0 2>>$O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', round(d['ms_per_step'],3), d.get('stages_ms_per_eval',{}).get('gram'), d.get('gram_hbm',{}).get('frac'))"; }
b cfg2 60 "cfg2"
b cfg2 60 "cfg2"
for r in 1 2 3; do
  MOGP_POTRF_PRIVATE=0 b cfg5 8 "cfg5 chain of K_uu on the model's stream"
  MOGP_POTRF_PRIVATE=1 b cfg5 8 "cfg5 chain of K_uu on the private stream"
done
python tools/cfg5_err.py 2>&1 | tail -4
