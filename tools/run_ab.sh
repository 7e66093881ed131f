cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_cfg5 -o p -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-shard-probe > $O/kt_cfg5.log 2>&1
echo done
