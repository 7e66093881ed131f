#!/bin/bash
O=gpurun_out/r5r; mkdir -p $O
timeout 600 python tools/titsias_chol_residual.py > $O/chol_residual.txt 2>&1
cat $O/chol_residual.txt
