#!/bin/bash
# round 5, fourth GPU call: the sharded evaluation's new exchange (ranks sharing this GPU), the dataflow kernel's HBM traffic (replay under --pmc),
# configs[2] forced onto the dataflow schedule (256 tile rows), the 1-rank sharded overhead
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "sharded or rccl or exchange_variants or concurrent_processes" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
b() { timeout 400 python bench.py --config $1 --steps $2 --warmup 3 --no-cpu-baseline --no-configs --sustained 0 $4 2>>$O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', round(d['ms_per_step'],3), d['config'].get('dataflow_kernel'), d['config'].get('fell_back'))"; }
b cfg2 60 "cfg2 default"
b cfg2 60 "cfg2 default"
b cfg3 3 "cfg3 default (phases)"
MOGP_GRAD_PATH=fused b cfg3 3 "cfg3 as dataflow (256 tile rows)"
b cfg3 3 "cfg3 sharded, 1-rank RCCL group" "--mode sharded"
cd /tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  FLOW_REPLAY_SERIAL=1 timeout -k 5 400 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_$cnt -o p -- python $GRAFT_REPO_ROOT/tools/flow_replay.py 8192 3 > $O/pmc_$cnt.log 2>&1
  grep -E "replay|alone|gradient" $O/pmc_$cnt.log | head -5
done
cd $GRAFT_REPO_ROOT
find $O -name "*counter_collection.csv" | head
python tools/pmc_flow.py "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" 8192 $O/pmc_traffic.json > $O/pmc_flow.txt 2>&1
cat $O/pmc_flow.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
