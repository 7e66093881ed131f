#!/bin/bash
# A/B of the substitution-leaf forms (MOGP_TRSM_LEAF) at configs[4]: results bit for bit, then time; then the dataflow schedule against POTRF/TRTRI/LAUUM above 80 tile rows
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-leaf}; mkdir -p $O
for f in 0 7 8; do
  echo "== form $f"; MOGP_TRSM_LEAF=$f timeout 300 python tools/cfg5_err.py 2>&1 | grep -E "checksum|loss rel|Z|bitwise|Exception|Error" | cut -c1-160
done 2>&1 | tee $O/leaf_check.txt
for f in 0 7 8 0 8; do
  MOGP_TRSM_LEAF=$f timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-shard-probe 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('form $f', round(d['ms_per_step'],2), 'ms')"
done 2>&1 | tee $O/leaf_time.txt
for p in; do MOGP_GRAD_PATH=$p SIZES=3414,4096,4779,5462 timeout 600 python tools/grad_path_sizes.py 2>&1 | tail -1; done | tee $O/sizes.txt
