#!/bin/bash
# round 3, GPU call 1: persistent chain kernel -- A/B against the launch-per-step chain, bench A/B, kernel trace + timeline, GPU test suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/c1; mkdir -p $O
timeout 900 python tools/chain_check.py > $O/chain_check.txt 2>&1; echo "rc=$?" >> $O/chain_check.txt
tail -15 $O/chain_check.txt
MOGP_CHAIN=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_chain0.json 2> $O/bench_chain0.err
MOGP_CHAIN=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_chain1.json 2> $O/bench_chain1.err
python - <<'PY'
import json
for t in ("0", "1"):
    try:
        d = json.loads(open("gpurun_out/c1/bench_chain%s.json" % t).read().strip().splitlines()[-1])
        print("MOGP_CHAIN=%s: %.2f evals/s  %.3f ms  stages %s" % (t, d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d.get("stages_ms_per_eval", {}).items()}))
    except Exception as e:
        print("MOGP_CHAIN=%s: no line (%r)" % (t, e))
PY
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py $O/kt --list > $O/timeline.txt 2>&1; head -40 $O/timeline.txt
rm -rf $O/kt/*/*.db 2>/dev/null
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
