#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/gramt; mkdir -p $O; cd /tmp
MOGP_FLOW=0 timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-configs --no-shard-probe > $O/kt.log 2>&1
KTRACE_TOP=30 python $GRAFT_REPO_ROOT/tools/ktrace.py $O/kt | grep -E "gram|moments"
MOGP_FLOW=0 timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/p -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc.log 2>&1
python - $O <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", "")
        if "k_gram_strip" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: "%.3g" % (sum(x) / len(x)) for c, x in sorted(v.items())})
PY
rm -rf $O/kt $O/p
