"""The N > 1 path of bench.py on CPU: two gloo ranks run the timed-region logic (barrier, K timed steps, max over ranks,
whole-job aggregate) with a dummy step; no GPU, no native calls."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, time, json
    sys.path.insert(0, %r)
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    calls = []
    def step(i):
        calls.append(i)
        time.sleep(0.01 * (1 + rank))          # rank 1 is twice as slow: the MAX over ranks must be reported
    dt = bench.timed_region(step, steps=5, warmup=2, dist=dist)
    assert calls == [-1, -1, 0, 1, 2, 3, 4], calls
    if rank == 0:
        print(json.dumps({"dt": dt, "value": bench.aggregate_value(world, 5, dt), "world": world}))
    dist.destroy_process_group()
''')


def test_two_rank_timing_and_aggregate(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2
    assert r["dt"] >= 5 * 0.02 * 0.9                       # the slow rank (20 ms/step) sets the time
    assert abs(r["value"] - 2 * 5 / r["dt"]) < 1e-9        # whole-job aggregate over both ranks
