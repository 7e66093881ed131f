#!/bin/bash
# round 5, third GPU call: the whole GPU suite (with the cfg3 golden), slot-cached descriptors alone (k_flow2 without its short look), the dataflow
# kernel's HBM traffic through the replay mode under rocprofv3 --pmc
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O
MOGP_FLOW_PIPE=0 timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
b() { timeout 200 python bench.py --config $1 --steps $2 --warmup 5 --no-cpu-baseline --no-configs --sustained 0 2>>$O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', round(d['ms_per_step'],3), d['config'].get('dataflow_kernel'), d['config'].get('fell_back'))"; }
for r in 1 2 3; do
  MOGP_FLOW_PIPE=0 b cfg2 60 "cfg2 r4-kernel               "
  MOGP_FLOW_PIPE=1 MOGP_FLOW_FAST=0 b cfg2 60 "cfg2 slots only (no short look)"
done
cd /tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  MOGP_FLOW_PIPE=0 FLOW_REPLAY_SERIAL=1 timeout -k 5 400 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_$cnt -o p -- python $GRAFT_REPO_ROOT/tools/flow_replay.py 8192 3 > $O/pmc_$cnt.log 2>&1
  tail -6 $O/pmc_$cnt.log
done
cd $GRAFT_REPO_ROOT
python tools/pmc_flow.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv") 8192 $O/pmc_traffic.json > $O/pmc_flow.txt 2>&1
cat $O/pmc_flow.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
