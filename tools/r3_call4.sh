#!/bin/bash
# round 3, GPU call 4: three-way split of the trailing update (remainder on its own stream) A/B; bench; trace; GPU tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/c4; mkdir -p $O
AB=MOGP_SPLIT3:0,1 timeout 900 python tools/chain_check.py 600,900,1100,1500,2048,3000,4097,6000,8192 > $O/s3_check.txt 2>&1; echo "rc=$?" >> $O/s3_check.txt
tail -12 $O/s3_check.txt
for v in 0 1; do MOGP_SPLIT3=$v timeout 300 python bench.py --no-cpu-baseline > $O/bench_s$v.json 2> $O/bench_s$v.err; done
python - <<'PY'
import json
for t in ("0", "1"):
    try:
        d = json.loads(open("gpurun_out/c4/bench_s%s.json" % t).read().strip().splitlines()[-1])
        print("MOGP_SPLIT3=%s: %.2f evals/s  %.3f ms  stages %s" % (t, d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d.get("stages_ms_per_eval", {}).items()}))
    except Exception as e:
        print("MOGP_SPLIT3=%s: no line (%r)" % (t, e))
PY
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py $O/kt > $O/timeline.txt 2>&1; head -30 $O/timeline.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
