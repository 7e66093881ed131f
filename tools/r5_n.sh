#!/bin/bash
O=gpurun_out/r5n; mkdir -p $O
timeout 600 python tools/gram_accuracy.py 8192 > $O/gram_accuracy.txt 2>&1
cat $O/gram_accuracy.txt
