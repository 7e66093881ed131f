#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-g16}; mkdir -p $O
for f in 0 1; do MOGP_MOM_X=$f timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe 2> $O/bench_m$f.err | tail -1 > $O/bench_m$f.json; done
for f in 0 1; do echo momx=$f; python - <<PY
import json; d=json.load(open("$O/bench_m$f.json")); print(d["ms_per_step"], d["stages_ms_per_eval"]["moment_kernel"], d["moments_hbm"]["frac"])
PY
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lml_and_gradient or raw_outputs_against_numpy or full_size or edge_cases or adam or lbfgs or dataflow" 2>&1 | tail -4
