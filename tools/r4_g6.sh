#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-g6}; mkdir -p $O
(timeout 150 python tools/flow_trace.py 8192) > $O/trace_8192.txt 2>&1
for nap in 4 2 1 0; do MOGP_FLOW_NAP=$nap timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe 2> $O/bench_nap$nap.err | tail -1 > $O/bench_nap$nap.json; done
head -8 $O/trace_8192.txt; grep -A12 "^queues" $O/trace_8192.txt; tail -3 $O/trace_8192.txt; for nap in 4 2 1 0; do echo nap $nap; cut -c100-240 $O/bench_nap$nap.json; done
