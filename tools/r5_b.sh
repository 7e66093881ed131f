#!/bin/bash
# round 5, second GPU call: k_flow2 with the replacement claim (refill) -- parity, A/B over refill_from, time stamps; the replay mode
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5b; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "dataflow or full_size or first_evaluation or native_library or cfg4" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
b() { timeout 200 python bench.py --config $1 --steps $2 --warmup 5 --no-cpu-baseline --no-configs --sustained 0 2>>$O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', round(d['ms_per_step'],3), d['config'].get('dataflow_kernel'), d['config'].get('fell_back'))"; }
for r in 1 2 3; do
  MOGP_FLOW_PIPE=0 b cfg2 60 "cfg2 r4-kernel          "
  MOGP_FLOW_PIPE=1 MOGP_FLOW_REFILL_FROM=99 b cfg2 60 "cfg2 pipelined no refill"
  MOGP_FLOW_PIPE=1 MOGP_FLOW_REFILL_FROM=4 b cfg2 60 "cfg2 pipelined refill>=4"
  MOGP_FLOW_PIPE=1 MOGP_FLOW_REFILL_FROM=2 b cfg2 60 "cfg2 pipelined refill>=2"
  MOGP_FLOW_PIPE=1 MOGP_FLOW_REFILL_FROM=1 b cfg2 60 "cfg2 pipelined refill>=1"
done
for r in 1 2; do
  MOGP_FLOW_PIPE=0 b cfg4 8 "cfg4 r4-kernel"
  MOGP_FLOW_PIPE=1 b cfg4 8 "cfg4 pipelined refill>=4"
done
(MOGP_FLOW_PIPE=1 timeout 150 python tools/flow_trace.py 8192) > $O/cfg2_timeline_pipe.txt 2>&1
head -4 $O/cfg2_timeline_pipe.txt; tail -4 $O/cfg2_timeline_pipe.txt
(timeout 300 python tools/flow_replay.py) > $O/replay.txt 2>&1; tail -12 $O/replay.txt
