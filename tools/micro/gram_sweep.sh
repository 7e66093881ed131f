cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/gram
for env in "MOGP_GRAM_STRIP2=1" "MOGP_GRAM_STRIP2=0" "MOGP_GRAM_STRIP2=1" "MOGP_GRAM_STRIP2=1 MOGP_STRIP_RUN=16" "MOGP_GRAM_STRIP2=1 MOGP_STRIP_RUN=12"; do
  env $env python tools/tile_kernels_time.py 8192 4 3 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$env', 'gram %.1f us (min %.1f) frac %.3f | moments %.1f us' % (d['gram_us'], d['gram_min_us'], d['gram_frac_hbm'], d['moments_us']))"
done > gpurun_out/gram/sweep3.txt
cat gpurun_out/gram/sweep3.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gram or lml or golden or titsias" 2>&1 | tail -3
