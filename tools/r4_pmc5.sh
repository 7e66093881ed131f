#!/bin/bash
# HBM-side traffic per kernel at configs[4] (FETCH_SIZE: 32-byte units x2 on gfx950 per the guide's correction -> reported here in GB per launch)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc5}; mkdir -p $O
cd /tmp
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p -o p -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc.log 2>&1
python - $O <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
f = glob.glob(O + "/p/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", ""), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
for (k, g), v in rows[:16]:
    print("%-40s grid %-9s launches %4d  fetch/launch %8.3f GB   total %7.2f GB" % (k[:40], g, len(v), 2 * 32 * sum(v) / len(v) / 1e9 / 1.0, 2 * 32 * sum(v) / 1e9))
PY
rm -rf $O/p
