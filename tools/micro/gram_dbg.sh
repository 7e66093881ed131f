cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/gram
for env in "MOGP_GRAM_DBG=0" "MOGP_GRAM_DBG=1" "MOGP_GRAM_DBG=2" "MOGP_GRAM_DBG=4" "MOGP_GRAM_DBG=3" "MOGP_GRAM_DBG=6" "MOGP_GRAM_DBG=7"; do
  env $env python tools/tile_kernels_time.py 8192 4 3 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$env', 'gram %.1f us (min %.1f)' % (d['gram_us'], d['gram_min_us']))"
done > gpurun_out/gram/dbg.txt
cat gpurun_out/gram/dbg.txt
