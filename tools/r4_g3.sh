#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-g3}; mkdir -p $O
(AB=MOGP_FLOW:0,1 MOGP_FLOW_MIN=2 timeout 400 python tools/chain_check.py ${2:-1500,4097,8192}) > $O/flow_check.txt 2>&1
(MOGP_FLOW_MIN=2 timeout 150 python tools/flow_trace.py 8192) > $O/trace_8192.txt 2>&1
for f in 1; do MOGP_FLOW=$f timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe 2> $O/bench_f$f.err | tail -1 > $O/bench_f$f.json; done
tail -20 $O/flow_check.txt; head -48 $O/trace_8192.txt; tail -25 $O/trace_8192.txt; cat $O/bench_f1.json | cut -c1-330
