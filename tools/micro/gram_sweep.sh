cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/gram
for i in 1 2; do
  python tools/tile_kernels_time.py 8192 4 3 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gram %.1f us (min %.1f) frac %.3f | moments %.1f us (min %.1f) frac %.3f' % (d['gram_us'], d['gram_min_us'], d['gram_frac_hbm'], d['moments_us'], d['moments_min_us'], d['moments_frac_hbm']))"
done > gpurun_out/gram/sweep4.txt
cat gpurun_out/gram/sweep4.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lml or golden or raw_outputs or dataflow" 2>&1 | tail -3
