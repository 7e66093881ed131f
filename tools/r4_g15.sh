#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-g15}; mkdir -p $O
(AB=MOGP_FLOW_TAIL:0,1 MOGP_FLOW_MIN=2 timeout 400 python tools/chain_check.py 1700,8192) > $O/tail_check.txt 2>&1
for r in 1 2; do for f in 0 1; do MOGP_FLOW_TAIL=$f timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe 2> $O/bench_t$f.err | tail -1 > $O/bench_t${f}_$r.json; done; done
cat $O/tail_check.txt | tail -3; for r in 1 2; do for f in 0 1; do echo tail=$f; cut -c100-240 $O/bench_t${f}_$r.json; done; done
