#!/bin/bash
# The round's measurement set on one MI355X box: bench lines of cfg2 .. cfg5, kernel traces, the dataflow kernel's own timeline, HBM / MFMA / VALU counters.
# usage (from the repo root, through gpurun):  bash tools/run_profiles.sh <tag>     -> gpurun_out/<tag>/...; copy what is judged into profiles/
# rocprofv3 --pmc serialises kernel dispatches; the dataflow kernel and the chain kernels that feed it must run CONCURRENTLY (they talk through
# counters), so under --pmc a dataflow evaluation times out and falls back, loudly, to the stream schedule: the counter passes therefore run with
# MOGP_FLOW=0 and describe the stream schedule's launches of the SAME tile products (and the Gram / moment kernels, which are the same either way).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-prof}; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err
for c in cfg3 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-configs --no-shard-probe 2> $O/b_$c.err | tail -1 > $O/b_$c.json; done
(timeout 150 python tools/flow_trace.py 8192) > $O/cfg2_timeline.txt 2>&1
(timeout 150 python tools/flow_trace_predict.py) > $O/cfg4_timeline.txt 2>&1
cd /tmp
for c in cfg2 cfg3 cfg4 cfg5; do
  st=5; wu=2; if [ $c = cfg2 ]; then st=20; wu=5; fi      # the headline as the driver runs it: the average then is the steady state's, not the warm-up's
  timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps $st --warmup $wu --sustained 0 --no-cpu-baseline --no-configs --no-shard-probe > $O/kt_$c.log 2>&1
done
# the dataflow kernel itself: ALONE on the replay plan (mogp_model_flow_replay; tools/flow_replay.py checks that it forms the same Kj^-1 bit for bit)
for cnt in FETCH_SIZE WRITE_SIZE; do
  FLOW_REPLAY_SERIAL=1 timeout -k 5 400 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmcflow_$cnt -o p -- python $GRAFT_REPO_ROOT/tools/flow_replay.py 8192 3 > $O/pmcflow_$cnt.log 2>&1
done
(cd $GRAFT_REPO_ROOT; python tools/pmc_flow.py "$(find $O/pmcflow_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmcflow_WRITE_SIZE -name '*counter_collection.csv' | head -1)" 8192 $O/pmc_flow_traffic.json > $O/pmc_flow.txt 2>&1)
(cd $GRAFT_REPO_ROOT; timeout 200 python tools/flow_replay.py 8192 5) > $O/flow_replay.txt 2>&1
# ... and what its waves do with their cycles (round 6: the counter-backed bound of the headline kernel)
FLOW_REPLAY_SERIAL=1 timeout -k 5 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmcflow_mfma -o p -- python $GRAFT_REPO_ROOT/tools/flow_replay.py 8192 3 > $O/pmcflow_mfma.log 2>&1
FLOW_REPLAY_SERIAL=1 timeout -k 5 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmcflow_wave -o p -- python $GRAFT_REPO_ROOT/tools/flow_replay.py 8192 3 > $O/pmcflow_wave.log 2>&1
rm -rf $O/pmcflow_FETCH_SIZE $O/pmcflow_WRITE_SIZE
for cnt in FETCH_SIZE WRITE_SIZE; do
  MOGP_FLOW=0 timeout -k 5 300 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_$cnt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --sustained 0 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc_$cnt.log 2>&1
done
MOGP_FLOW=0 timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_valu -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --sustained 0 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc_valu.log 2>&1
MOGP_FLOW=0 timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --sustained 0 --no-cpu-baseline --no-configs --no-shard-probe > $O/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
for c in cfg2 cfg3 cfg4 cfg5; do python tools/ktrace.py $O/kt_$c --csv $O/${c}_kernel_stats.csv > /dev/null 2>&1; done
python tools/eval_timeline.py $O/kt_cfg5 60 > $O/cfg5_timeline.txt 2>&1
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv") 5 $O/pmc_traffic.json > $O/pmc_hbm_traffic.csv 2>&1
python - $O <<'PY' > $O/pmc_counters.txt 2>&1
import csv, glob, collections, sys
O = sys.argv[1]
print("# stream schedule (MOGP_FLOW=0: rocprofv3 --pmc serialises dispatches, which the co-operating dataflow / chain kernels cannot run under)")
for d in ("pmc_valu", "pmc_mfma", "pmcflow_mfma", "pmcflow_wave"):
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("# " + d + ": per-launch averages (SQ_* counters count in units of 4 cycles / per wave instruction; BUSY_CYCLES is summed over 32 shader engines)")
        for k, v in sorted(acc.items()):
            print("%-44s launches %5d  " % (k[:44], len(next(iter(v.values())))) + "  ".join("%s=%.4g" % (c, sum(x) / len(x)) for c, x in sorted(v.items())))
PY
rm -rf $O/kt_cfg2 $O/kt_cfg3 $O/kt_cfg4 $O/kt_cfg5 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_valu $O/pmc_mfma $O/pmcflow_mfma $O/pmcflow_wave
tail -c 1800 $O/bench_line.json; for c in cfg3 cfg4 cfg5; do python -c "
import json; d=json.loads(open('$O/b_$c.json').read()); print('$c', round(d['ms_per_step'],2),'ms frac',round(d['roofline']['frac'],3))"; done
