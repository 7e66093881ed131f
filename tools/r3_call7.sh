#!/bin/bash
# round 3, GPU call 7: new bench.py line (configs, direct cpu baseline, Adam in the step), chain kernel inside the sweep (sharded path), probes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/c7; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "schedule or sharded or rccl or cfg3" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c7/bench_default.json").read().strip().splitlines()[-1])
print("cfg2 %.2f evals/s %.3f ms | configs %s | cpu %s" % (d["value"], d["ms_per_step"], {k: (round(v.get("ms_per_step", -1), 2), round(v.get("frac", -1), 3)) if "error" not in v else v for k, v in d.get("configs", {}).items()}, d.get("cpu_baseline")))
PY
for mode in 0 1; do MOGP_CHAIN=$mode timeout 600 python bench.py --config cfg3 --mode sharded --steps 3 --warmup 1 > $O/cfg3_sharded_chain$mode.json 2> $O/cfg3_sharded_chain$mode.err; python -c "
import json; d=json.loads(open('$O/cfg3_sharded_chain$mode.json').read().strip().splitlines()[-1]); print('cfg3 sharded 1 rank MOGP_CHAIN=$mode', round(d['ms_per_step'],1), 'ms', d['config'].get('rccl_ranks'))"; done
timeout 900 python bench.py --shard-probe --probes cfg3,cfg2 --no-cpu-baseline --no-configs > $O/bench_probe.json 2> $O/bench_probe.err; python -c "
import json; d=json.loads(open('$O/bench_probe.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('sharded'), indent=1)[:1800])"
