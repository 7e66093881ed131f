#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-g14}; mkdir -p $O
(AB=MOGP_GRAM_SPLIT:0,1 MOGP_FLOW_MIN=2 timeout 400 python tools/chain_check.py 900,1700,4097,8192) > $O/split_check.txt 2>&1
for r in 1 2; do for f in 0 1; do MOGP_GRAM_SPLIT=$f timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-shard-probe 2> $O/bench_s$f.err | tail -1 > $O/bench_s${f}_$r.json; done; done
cat $O/split_check.txt | tail -5; for r in 1 2; do for f in 0 1; do echo split=$f; cut -c100-240 $O/bench_s${f}_$r.json; done; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dataflow or full_size or schedule" 2>&1 | tail -3
