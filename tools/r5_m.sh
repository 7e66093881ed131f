#!/bin/bash
# round 5, call m: dELBO/dZ at configs[4] against the truth with the M x N solve in the backward part (MOGP_TITSIAS_GB=0) and with the round-4 order
O=gpurun_out/r5m; mkdir -p $O
for gb in 1 0; do
  echo "== MOGP_TITSIAS_GB=$gb" >> $O/gb.txt
  MOGP_TITSIAS_GB=$gb timeout 300 python tools/cfg5_err.py 2>&1 | grep -v "^MultiOutput\|^Gaussian" >> $O/gb.txt
  MOGP_TITSIAS_GB=$gb timeout 300 python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-shard-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 ms_per_step', d['ms_per_step'])" >> $O/gb.txt
done
cat $O/gb.txt
