#!/bin/bash
# round 3, GPU call 3: does confining the inverse's streams to a subset of the CUs (the Cholesky's trailing updates pace the chain) help?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/c3; mkdir -p $O
export MOGP_LOOKAHEAD=1
for n in 0 160 120 80 48; do
  MOGP_INV_CUS=$n timeout 300 python bench.py --no-cpu-baseline > $O/bench_inv$n.json 2> $O/bench_inv$n.err
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/c3/bench_inv%s.json" % n).read().strip().splitlines()[-1])
    print("MOGP_INV_CUS=%s: %.2f evals/s  %.3f ms  potrf %.2f" % (n, d["value"], d["ms_per_step"], d["stages_ms_per_eval"]["potrf"]))
except Exception as e:
    print("MOGP_INV_CUS=%s: no line (%r)" % (n, e))
PY
done
cd /tmp; MOGP_INV_CUS=80 timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py $O/kt > $O/timeline.txt 2>&1; head -30 $O/timeline.txt
