#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4b}; mkdir -p $O; cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_cfg2 -o p -- python $GRAFT_REPO_ROOT/bench.py --config cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe > $O/kt_cfg2.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/ktrace.py $O/kt_cfg2 --csv $O/cfg2_kernel_stats.csv | head -8; rm -rf $O/kt_cfg2; tail -c 600 $O/kt_cfg2.log
