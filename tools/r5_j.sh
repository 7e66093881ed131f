#!/bin/bash
# round 5, call j: the configs[4] truth fixture on the device, the reference-format checkpoint round trip, the 1-GPU shard probe of bench.py
O=gpurun_out/r5j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "extended_precision_truth or reference_checkpoints_on_device or cfg5_titsias_golden" > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --shard-probe > $O/bench_probe.json 2> $O/bench_probe.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5j/bench_probe.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
print("sharded", json.dumps(d.get("sharded") or d.get("sharded_cfg3"))[:1500])
PY
