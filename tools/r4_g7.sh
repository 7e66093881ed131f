#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-g7}; mkdir -p $O
(FLOW_TRACE_SAVE=$O/trace_raw.npz timeout 150 python tools/flow_trace.py 8192) > $O/trace_8192.txt 2>&1
for nap in 4 5 6; do MOGP_FLOW_NAP=$nap timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe 2> $O/bench_nap$nap.err | tail -1 > $O/bench_nap$nap.json; done
for nap in 4 5 6; do echo nap $nap; cut -c100-240 $O/bench_nap$nap.json; done
