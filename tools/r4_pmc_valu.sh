#!/bin/bash
# where the waves of the Gram / moment kernels spend their cycles (stream schedule: --pmc serialises dispatches)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcv}; mkdir -p $O
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU"; do
  i=$((i+1))
  MOGP_FLOW=0 timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc$i.log 2>&1
done
python - $O <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", "")
        if any(s in k for s in ("k_moments_x", "k_gram_strip", "k_gram<", "k_moments<")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k)
    for c, x in sorted(v.items()):
        print("    %-28s %12.4g per launch (%d launches)" % (c, sum(x) / len(x), len(x)))
PY
rm -rf $O/p1 $O/p2 $O/p3
