#!/bin/bash
# kernel trace of bench.py (cfg2) -> timeline + windowed GEMM rate.  usage (gpurun): bash tools/r3_trace.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=${1:-trace}; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py $O/kt > $O/timeline.txt 2>&1; python tools/gemm_rate.py $O/kt >> $O/timeline.txt 2>&1; python tools/ktrace.py $O/kt --csv $O/kernel_stats.csv > /dev/null 2>&1
grep -E "span|periods|window" $O/timeline.txt
