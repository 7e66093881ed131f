#!/bin/bash
# A/B of the headline bench line under environment variants: usage (gpurun): bash tools/ab_bench.sh "VAR=val VAR2=val" "..." ...   (one quoted string per variant; "" = defaults)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  out=$(env $v python bench.py --no-cpu-baseline --no-configs --no-shard-probe 2>/dev/null | tail -1)
  python - "$v" <<PY
import json, sys
d = json.loads('''$out''')
s = d.get("stages_ms_per_eval", {})
print("%-52s %7.2f evals/s  median %.3f ms  mean %.3f  sustained %.2f  kernel %.3f ms  gram %.0f us  moments %.0f us  timeouts %s" % (sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"], d["mean_ms_per_step"], d["sustained"]["value"], s.get("gemm_kernel", 0), 1e3 * s.get("gram_kernel", 0), 1e3 * s.get("moment_kernel", 0), d["config"]["dataflow_timeouts"]))
PY
done
