#!/bin/bash
# round 5, call l: does the summation tree of Q = v v^T (split-K slices) move dELBO/dZ relative to the 80-bit truth at configs[4]?
O=gpurun_out/r5l; mkdir -p $O
for ks in 0 32 64 128 -64; do
  echo "== MOGP_SYRK_KS=$ks" >> $O/ks.txt
  MOGP_SYRK_KS=$ks timeout 300 python tools/cfg5_err.py 2>&1 | grep -v "^MultiOutput\|^Gaussian" >> $O/ks.txt
done
cat $O/ks.txt
