#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_try; mkdir -p $O
cd /tmp
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_F -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc_F.log 2>&1
echo rc=$?
tail -3 $O/pmc_F.log | cut -c1-600
python - <<PY
import csv, glob, collections
f = glob.glob("$O/pmc_F/**/*counter_collection.csv", recursive=True)
print(f)
if f:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", "")
        acc[n][0] += 1; acc[n][1] += float(r["Counter_Value"])
    for n, (c, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
        print("%-40s launches %5d  FETCH_SIZE KiB total %.4g  per launch %.4g" % (n[:40], c, v, v / c))
PY
rm -rf $O/pmc_F
