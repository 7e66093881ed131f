#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-g19}; mkdir -p $O
for r in 1 2; do for f in 1 0; do MOGP_FLOW_REFILL=$f timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe 2> $O/bench_r$f.err | tail -1 > $O/bench_r${f}_$r.json; done; done
for r in 1 2; do for f in 1 0; do echo refill=$f; cut -c100-240 $O/bench_r${f}_$r.json; done; done
(MOGP_FLOW_REFILL=0 timeout 150 python tools/flow_trace.py 8192) > $O/trace_norefill.txt 2>&1; head -4 $O/trace_norefill.txt; tail -2 $O/trace_norefill.txt
