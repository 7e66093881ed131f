#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/shard4; mkdir -p $O
for i in 1 2 3; do
MASTER_ADDR=127.0.0.1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port $((29800+i)) tools/shard_check.py --points 3000 --backend gloo > $O/out$i.txt 2> $O/err$i.txt
echo run $i rc=$?
grep -i "mogp\|error\|definite\|timed" $O/err$i.txt | grep -v "all-gather callback\|ChildFailed\|elastic\|Traceback" | head -8
done
