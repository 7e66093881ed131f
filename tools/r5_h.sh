#!/bin/bash
# round 5: K_uu's chain on the private stream A/B (configs[4]); stall counters of the moment and Gram tile kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
b() { timeout 300 python bench.py --config $1 --steps $2 --warmup 3 --no-cpu-baseline --no-configs --sustained 0 2>>$O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', round(d['ms_per_step'],3))"; }
for r in 1 2 3; do
  MOGP_POTRF_PRIVATE=0 b cfg5 8 "cfg5 K_uu chain on the model stream  "
  MOGP_POTRF_PRIVATE=1 b cfg5 8 "cfg5 K_uu chain on the private stream"
done
MOGP_POTRF_PRIVATE=0 python tools/cfg5_err.py 2>&1 | tail -2
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/counters_list.txt | sort -u | tr '\n' ' ' | head -c 3000; echo
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  MOGP_FLOW=0 timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-shard-probe --sustained 0 > $O/pmc_$tag.log 2>&1
  python - $O/pmc_$tag <<'PY'
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", "")
        if k.startswith("k_moments") or k.startswith("k_gram"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-24s launches %3d  " % (k[:24], len(next(iter(v.values())))) + "  ".join("%s=%.4g" % (c, sum(x) / len(x)) for c, x in sorted(v.items())))
PY
  rm -rf $O/pmc_$tag
done
