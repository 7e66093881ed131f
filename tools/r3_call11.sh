#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/c11; mkdir -p $O
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Imogptk_amd/csrc -Iinclude tools/micro/gemm_rank.hip -o /tmp/gemm_rank 2> $O/build.err
for v in 0 1; do echo "== MOGP_GEMM16=$v"; MOGP_GEMM16=$v timeout 120 /tmp/gemm_rank; done > $O/gemm_rank.txt 2>&1
grep -E "==|32x64|K=4096|mt 64|mt 32" $O/gemm_rank.txt
AB=MOGP_GEMM16:0,1 timeout 600 python tools/chain_check.py 2048,8192 > $O/g16_check.txt 2>&1; cat $O/g16_check.txt | tail -3
for v in 0 1; do MOGP_GEMM16=$v timeout 300 python bench.py --no-cpu-baseline --no-configs > $O/bench_g$v.json 2> $O/bench_g$v.err; python -c "
import json; d=json.loads(open('$O/bench_g$v.json').read().strip().splitlines()[-1]); print('MOGP_GEMM16=$v', round(d['value'],2), 'evals/s', round(d['ms_per_step'],3), 'ms')"; done
