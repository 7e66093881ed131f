#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-g24}; mkdir -p $O
for r in 1 2; do for f in 0 3 4; do MOGP_FLOW_NHI=$f timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe 2> $O/bench_h$f.err | tail -1 > $O/bench_h${f}_$r.json; done; done
for r in 1 2; do for f in 0 3 4; do echo nhi=$f; cut -c100-240 $O/bench_h${f}_$r.json; done; done
(MOGP_FLOW_NHI=4 timeout 150 python tools/flow_trace.py 8192) > $O/trace.txt 2>&1; head -6 $O/trace.txt; sed -n 8,26p $O/trace.txt; tail -2 $O/trace.txt
