#!/bin/bash
# The round's measurement set on one MI355X box: bench lines of cfg2 .. cfg5, kernel traces, cfg2 timeline, HBM / MFMA / VALU counters, GEMM microbenchmarks.
# usage (from the repo root, through gpurun):  bash tools/run_profiles.sh <tag>     -> gpurun_out/<tag>/...; copy what is judged into profiles/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-prof}; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err
for c in cfg3 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-configs --no-shard-probe 2> $O/b_$c.err | tail -1 > $O/b_$c.json; done
cd /tmp
for c in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-shard-probe > $O/kt_$c.log 2>&1
done
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_$cnt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc_$cnt.log 2>&1
done
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_valu -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc_valu.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
for c in cfg2 cfg3 cfg4 cfg5; do python tools/ktrace.py $O/kt_$c --csv $O/${c}_kernel_stats.csv > /dev/null 2>&1; done
python tools/timeline.py $O/kt_cfg2 > $O/cfg2_timeline.txt 2>&1
python tools/gemm_rate.py $O/kt_cfg2 >> $O/cfg2_timeline.txt 2>&1
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv") 5 $O/pmc_traffic.json > $O/pmc_hbm_traffic.csv 2>&1
python - $O <<'PY' > $O/pmc_counters.txt 2>&1
import csv, glob, collections, sys
O = sys.argv[1]
for d in ("pmc_valu", "pmc_mfma"):
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("# " + d + ": per-launch averages (SQ_* counters count in units of 4 cycles / per wave instruction; BUSY_CYCLES is summed over 32 shader engines)")
        for k, v in sorted(acc.items()):
            print("%-44s launches %5d  " % (k[:44], len(next(iter(v.values())))) + "  ".join("%s=%.4g" % (c, sum(x) / len(x)) for c, x in sorted(v.items())))
PY
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Imogptk_amd/csrc -Iinclude tools/micro/gemm_rank.hip -o /tmp/gemm_rank 2> $O/micro_build.err
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGEMM_TIMING -Imogptk_amd/csrc -Iinclude tools/micro/gemm_timing.hip -o /tmp/gemm_timing 2>> $O/micro_build.err
(timeout 100 /tmp/gemm_rank; timeout 100 /tmp/gemm_timing) > $O/gemm_micro.txt 2>&1
timeout 100 /tmp/gemm_rank sk > $O/gemm_streamk.txt 2>&1
timeout 100 /tmp/gemm_rank series > $O/gemm_clock_ramp.txt 2>&1
(timeout 200 python tools/long_series.py 8192 10; timeout 200 python tools/long_series.py 8192 3; timeout 300 python tools/long_series.py 16384 10) > $O/long_series.txt 2>&1
tail -c 1500 $O/bench_line.json; for c in cfg3 cfg4 cfg5; do python -c "
import json; d=json.loads(open('$O/b_$c.json').read()); print('$c', round(d['ms_per_step'],2),'ms frac',round(d['roofline']['frac'],3))"; done
