#!/bin/bash
# round 3, GPU call 2: two-block look-ahead schedule (potri.hip) A/B against the one-block schedule; bench; trace; GPU tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/c2; mkdir -p $O
AB=MOGP_LOOKAHEAD:1,2 timeout 900 python tools/chain_check.py 300,600,900,1100,1500,2048,3000,4097,6000,8192 > $O/la_check.txt 2>&1; echo "rc=$?" >> $O/la_check.txt
tail -12 $O/la_check.txt
for la in 1 2; do MOGP_LOOKAHEAD=$la timeout 300 python bench.py --no-cpu-baseline > $O/bench_la$la.json 2> $O/bench_la$la.err; done
python - <<'PY'
import json
for t in ("1", "2"):
    try:
        d = json.loads(open("gpurun_out/c2/bench_la%s.json" % t).read().strip().splitlines()[-1])
        print("MOGP_LOOKAHEAD=%s: %.2f evals/s  %.3f ms  stages %s" % (t, d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d.get("stages_ms_per_eval", {}).items()}))
    except Exception as e:
        print("MOGP_LOOKAHEAD=%s: no line (%r)" % (t, e))
PY
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py $O/kt --list > $O/timeline.txt 2>&1; head -36 $O/timeline.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
