#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/x4; mkdir -p $O
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-configs > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/ktrace.py $O/kt --csv $O/kernel_stats.csv 2>/dev/null | head -30
