#!/bin/bash
O=gpurun_out/r5q; mkdir -p $O
timeout 1500 python tools/titsias_stage_errors.py > $O/stage_errors.txt 2>&1
cat $O/stage_errors.txt
