#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/c5; mkdir -p $O
for v in 0 1 2; do MOGP_SPLIT3=$v timeout 300 python bench.py --no-cpu-baseline > $O/bench_s$v.json 2> $O/bench_s$v.err
python - $v <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/c5/bench_s%s.json" % t).read().strip().splitlines()[-1])
    print("MOGP_SPLIT3=%s: %.2f evals/s  %.3f ms  potrf %.2f" % (t, d["value"], d["ms_per_step"], d["stages_ms_per_eval"]["potrf"]))
except Exception as e:
    print("MOGP_SPLIT3=%s: no line (%r)" % (t, e))
PY
done
cd /tmp
for v in 1 2; do MOGP_SPLIT3=$v timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt$v -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt$v.log 2>&1; done
cd $GRAFT_REPO_ROOT; for v in 1 2; do python tools/timeline.py $O/kt$v > $O/timeline$v.txt 2>&1; grep -A3 "busy us" $O/timeline$v.txt; grep k_chain $O/timeline$v.txt | head -2; done
