#!/bin/bash
O=gpurun_out/r5o; mkdir -p $O
GRAM_FROM_DEVICE=1 timeout 1200 python tools/titsias_n100k_numpy.py Hrge > $O/device_gram_numpy_algebra.txt 2>&1
cat $O/device_gram_numpy_algebra.txt
