#!/bin/bash
# round 5, call r: panels of the K_uu factorisation refined once against L_kk -- residual of the factor, dELBO/dZ against the truth, time
O=gpurun_out/r5s; mkdir -p $O
for rp in 1 0; do
  echo "== MOGP_REFINE_PANELS=$rp" >> $O/refine.txt
  MOGP_REFINE_PANELS=$rp timeout 300 python tools/titsias_chol_residual.py 2>&1 | grep -v "^\[\|^ \[" >> $O/refine.txt
  MOGP_REFINE_PANELS=$rp timeout 300 python tools/cfg5_err.py 2>&1 | grep -v "^MultiOutput\|^Gaussian" >> $O/refine.txt
  MOGP_REFINE_PANELS=$rp timeout 300 python bench.py --config cfg5 --steps 8 --warmup 3 --no-cpu-baseline --no-configs --no-shard-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 ms_per_step', d['ms_per_step'])" >> $O/refine.txt
done
cat $O/refine.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "titsias or snelson or hensman or svgp or sparse or cfg5" 2>&1 | tail -5
