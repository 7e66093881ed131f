#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-leaf2}; mkdir -p $O
cd /tmp
for f in ${FORMS:-5}; do
MOGP_TRSM_LEAF=$f timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$f -o p -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 4 --warmup 2 --no-cpu-baseline --no-configs --no-shard-probe > $O/kt_$f.log 2>&1
echo "== form $f"; KTRACE_TOP=14 python $GRAFT_REPO_ROOT/tools/ktrace.py $O/kt_$f
rm -rf $O/kt_$f
done
