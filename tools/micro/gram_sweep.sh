cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/gram
for env in "MOGP_STRIP_GRADE=0" "MOGP_STRIP_GRADE=1" "MOGP_STRIP_GRADE=2" "MOGP_STRIP_GRADE=0 MOGP_STRIP_RUN=1" "MOGP_STRIP_GRADE=0 MOGP_STRIP_RUN=2" "MOGP_STRIP_GRADE=0 MOGP_STRIP_RUN=8" "MOGP_STRIP_GRADE=1 MOGP_STRIP_RUN=8" "MOGP_STRIP_GRADE=1 MOGP_GRAM_NC=2" "MOGP_STRIP_GRADE=0 MOGP_GRAM_NC=2"; do
  env $env python tools/tile_kernels_time.py 8192 4 3 30 2>/dev/null | tail -1
done > gpurun_out/gram/sweep1.txt
cut -c1-400 gpurun_out/gram/sweep1.txt
