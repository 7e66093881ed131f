#!/bin/bash
# round 5: the dataflow kernel's HBM traffic (replay under rocprofv3 --pmc), second attempt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O
cd /tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  FLOW_REPLAY_SERIAL=1 timeout -k 5 400 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_$cnt -o p -- python $GRAFT_REPO_ROOT/tools/flow_replay.py 8192 3 > $O/pmc_$cnt.log 2>&1
  echo "rc $?"; grep -E "replay|alone|gradient|SIGSEGV" $O/pmc_$cnt.log | head -5
done
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" | head
python tools/pmc_flow.py "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" 8192 $O/pmc_traffic.json > $O/pmc_flow.txt 2>&1
cat $O/pmc_flow.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
