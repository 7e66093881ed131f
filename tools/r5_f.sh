#!/bin/bash
# round 5: XCD-aware claims of the dataflow kernel (one head per XCD + locality interleave) against one head per queue: parity, time, HBM-side traffic
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5f; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "dataflow or full_size or first_evaluation or native_library or cfg4 or concurrent" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
b() { timeout 200 python bench.py --config $1 --steps $2 --warmup 5 --no-cpu-baseline --no-configs --sustained 0 2>>$O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', round(d['ms_per_step'],3), d['config'].get('dataflow_kernel'), d['config'].get('fell_back'))"; }
for r in 1 2 3; do
  MOGP_FLOW_XCD=0 b cfg2 60 "cfg2 one head per queue"
  MOGP_FLOW_XCD=1 b cfg2 60 "cfg2 one head per XCD  "
done
for r in 1 2; do
  MOGP_FLOW_XCD=0 b cfg4 8 "cfg4 one head per queue"
  MOGP_FLOW_XCD=1 b cfg4 8 "cfg4 one head per XCD  "
done
cd /tmp
for x in 0 1; do for cnt in FETCH_SIZE WRITE_SIZE; do
  MOGP_FLOW_XCD=$x FLOW_REPLAY_SERIAL=1 timeout -k 5 400 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc${x}_$cnt -o p -- python $GRAFT_REPO_ROOT/tools/flow_replay.py 8192 3 > $O/pmc${x}_$cnt.log 2>&1
  grep -E "alone" $O/pmc${x}_$cnt.log | head -2
done
cd $GRAFT_REPO_ROOT
python tools/pmc_flow.py "$(find $O/pmc${x}_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc${x}_WRITE_SIZE -name '*counter_collection.csv' | head -1)" 8192 $O/pmc_traffic_xcd$x.json > $O/pmc_flow_xcd$x.txt 2>&1
echo "---- MOGP_FLOW_XCD=$x"; cat $O/pmc_flow_xcd$x.txt
cd /tmp; done
rm -rf $O/pmc0_FETCH_SIZE $O/pmc0_WRITE_SIZE $O/pmc1_FETCH_SIZE $O/pmc1_WRITE_SIZE
(cd $GRAFT_REPO_ROOT; MOGP_FLOW_XCD=1 timeout 150 python tools/flow_trace.py 8192) > $O/cfg2_timeline.txt 2>&1; head -3 $O/cfg2_timeline.txt; tail -3 $O/cfg2_timeline.txt
