#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-g20}; mkdir -p $O
for r in 1 2; do for f in 0 1; do MOGP_FLOW_CLAIM1=$f timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe 2> $O/bench_c$f.err | tail -1 > $O/bench_c${f}_$r.json; done; done
for r in 1 2; do for f in 0 1; do echo claim1=$f; cut -c100-240 $O/bench_c${f}_$r.json; done; done
(MOGP_FLOW_CLAIM1=1 timeout 150 python tools/flow_trace.py 8192) > $O/trace_claim1.txt 2>&1; head -4 $O/trace_claim1.txt; tail -2 $O/trace_claim1.txt
