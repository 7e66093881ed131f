#!/bin/bash
O=gpurun_out/r5t; mkdir -p $O
timeout 900 python tools/exact_illcond.py 2048 > $O/exact_illcond.txt 2>&1
timeout 900 python tools/exact_illcond.py 8192 >> $O/exact_illcond.txt 2>&1
cat $O/exact_illcond.txt
