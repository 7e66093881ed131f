#!/bin/bash
# round 5, call k: own rows skipped in the unpack, the two-message exchange also for a group of one rank
O=gpurun_out/r5k; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shard" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --shard-probe > $O/bench_probe.json 2> $O/bench_probe.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5k/bench_probe.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
s = d.get("sharded") or {}
for k, v in s.items():
    print(k, json.dumps({a: b for a, b in v.items() if a != "split_note"})[:1200])
PY
