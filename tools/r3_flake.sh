#!/bin/bash
# how often does the shared-GPU sharded check fail under a switch?  usage (gpurun): bash tools/r3_flake.sh "VAR=val" [ranks] [n]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=${2:-2}; N=${3:-6}; bad=0
for i in $(seq $N); do
  env $1 MASTER_ADDR=127.0.0.1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$R --master-addr 127.0.0.1 --master-port $((29700 + i)) tools/shard_check.py --points 3000 --backend gloo > /tmp/fl.out 2> /tmp/fl.err
  rc=$?
  python - <<'PY'
import json
try:
    r = json.loads([l for l in open('/tmp/fl.out').read().splitlines() if l.startswith('{')][-1])
    if max(r["rel_loss"], r["titsias"]["rel_loss"], r["hensman"]["rel_loss"], r["snelson"]["rel_loss"]) > 1e-8: print("  PER RANK [l0 l1 tl0 tl1 hl0 hl1 nl0 nl1]:", r.get("per_rank"))
    print("  rel_loss %.1e rel_grad %.1e | titsias %.1e %.1e | hensman %.1e | snelson %.1e %.1e" % (r["rel_loss"], r["rel_grad"], r["titsias"]["rel_loss"], r["titsias"]["rel_grad"], r["hensman"]["rel_loss"], r["snelson"]["rel_loss"], r["snelson"]["rel_grad"]))
except Exception as e:
    print("  no result:", [l for l in open('/tmp/fl.err').read().splitlines() if 'Error' in l or 'mogp' in l][-3:])
PY
done
