#!/bin/bash
# round 3, GPU call 10: eight-wave GEMM tile (MOGP_GEMM8=1) -- micro rates, parity against the four-wave kernel, bench A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/c10; mkdir -p $O
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Imogptk_amd/csrc -Iinclude tools/micro/gemm_rank.hip -o /tmp/gemm_rank 2> $O/build.err
for v in 0 1; do echo "== MOGP_GEMM8=$v"; MOGP_GEMM8=$v timeout 120 /tmp/gemm_rank; done > $O/gemm_rank.txt 2>&1
cat $O/gemm_rank.txt
AB=MOGP_GEMM8:0,1 timeout 600 python tools/chain_check.py 900,2048,4097,8192 > $O/g8_check.txt 2>&1; cat $O/g8_check.txt | tail -5
for v in 0 1; do MOGP_GEMM8=$v timeout 300 python bench.py --no-cpu-baseline --no-configs > $O/bench_g$v.json 2> $O/bench_g$v.err; python -c "
import json; d=json.loads(open('$O/bench_g$v.json').read().strip().splitlines()[-1]); print('MOGP_GEMM8=$v', round(d['value'],2), 'evals/s', round(d['ms_per_step'],3), 'ms')"; done
for c in cfg3 cfg4 cfg5; do for v in 0 1; do MOGP_GEMM8=$v timeout 300 python bench.py --config $c --no-cpu-baseline --steps 3 > $O/b_${c}_g$v.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/b_${c}_g$v.json').read().strip().splitlines()[-1]); print('$c MOGP_GEMM8=$v', round(d['ms_per_step'],2), 'ms')"; done; done
