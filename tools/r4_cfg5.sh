#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-c5}; mkdir -p $O
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_cfg5 -o p -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 4 --warmup 2 --no-cpu-baseline --no-configs --no-shard-probe > $O/kt_cfg5.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/eval_timeline.py $O/kt_cfg5 100 > $O/cfg5_timeline.txt 2>&1
rm -rf $O/kt_cfg5
tail -2 $O/kt_cfg5.log | cut -c1-300; cat $O/cfg5_timeline.txt | head -150
