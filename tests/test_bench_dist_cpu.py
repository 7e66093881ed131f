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


PROBE_WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r)
    import bench
    rank = int(os.environ["RANK"])
    base = int(os.environ["MASTER_PORT"]) + 20
    out = bench.run_probes(["dummy"], base, timeout_s=60.0)
    out2 = bench.run_probes(["hang", "dummy"], base + 10, timeout_s=25.0)
    if rank == 0:
        print(json.dumps({"dummy": out["dummy"], "hang": out2["hang"], "after_hang": out2["dummy"]}))
''')


def test_sharded_probes_are_child_groups_with_their_own_watchdog(tmp_path):
    """bench.py at N > 1 runs every sharded probe as a process group of its own BEFORE the timed region: two ranks each start a child, the
    children rendezvous on a port of their own (gloo here), rank 0's child reports through stdout; a probe whose collective never returns
    is killed by its watchdog and costs that entry only (`dummy` = one all-reduce of (1, rank + 1), `hang` = a rank that never arrives)."""
    import json
    script = tmp_path / "probe_worker.py"
    script.write_text(PROBE_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["dummy"]["rccl_ranks"] == 2 and r["dummy"]["rank_sum_ok"] is True, r
    assert r["dummy"]["probe_wall_s"] > 0
    assert "watchdog" in r["hang"]["error"], r
    assert r["hang"].get("rccl_ranks") == 2 and r["hang"].get("stage") == "before the hang", r      # what the child had finished is kept
    assert "skipped" in r["after_hang"]["error"], r            # a probe behind one that hung is not started


def test_line_objects_have_the_keys_the_driver_reads():
    """`configs` entries (cfg3 / cfg4 / cfg5 on the GPU of the default line) and the cpu_baseline flag, without a GPU"""
    sys.path.insert(0, ROOT)
    import bench
    e = bench.config_entry("cfg3", 550.0, 3, float(32768) ** 3)
    assert set(["ms_per_step", "frac", "steps", "workload", "value", "unit"]) <= set(e)
    assert abs(e["frac"] - (32768.0 ** 3 / 0.55 / 1e12) / bench.FP64_MFMA_PEAK_TFLOPS) < 1e-12
    assert bench.config_entry("cfg4", 49.0, 5, 2.57e12)["unit"] == "calls/s"
    assert set(bench.PROBES) >= {"cfg3", "cfg2", "cfg5"} and bench.PROBES[0] == "cfg3"       # the cfg3 probe runs first
    b = bench.cpu_baseline(256, 2, 2)              # tiny: timed directly at its own N
    assert b["extrapolated"] is False and b["kind"] == "port" and b["value"] > 0 and "N=256" in b["sample"]


def test_plain_process_launches_its_own_ranks():
    """`python bench.py --gpus 2` as a PLAIN process (WORLD_SIZE unset: how the driver starts the 1-GPU line) must not run on one device and
    print n_gpus 1: it starts the two ranks itself.  --selftest keeps the GPU out of it (gloo ranks, a sleeping step, the `dummy` probe)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                        # ONE line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["config"]["self_launched"] is True
    assert r["rccl_ranks"] == 2 and r["sharded"]["dummy"]["rank_sum_ok"] is True      # the probe's child group spanned both ranks
    assert abs(r["value"] - 2 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]


def test_more_gpus_than_the_node_has_is_an_error_with_a_reason():
    """--gpus 8 on a box with fewer GPUs (none here): exit code 2 and one line on stderr saying why -- not a 1-GPU line"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    msg = [l for l in out.stderr.splitlines() if l.startswith("bench.py:")]
    assert len(msg) == 1 and "--gpus 8" in msg[0] and "GPU" in msg[0], out.stderr


def test_sharded_headline_is_top_level():
    sys.path.insert(0, ROOT)
    import bench
    sh = {"cfg3": {"ms_one_gpu": 540.0, "ms_sharded": 120.0, "speedup": 4.5, "rccl_ranks": 8, "rel_loss": 1e-13, "rel_grad": 1e-9, "probe_wall_s": 30.0},
          "ranks": 8}
    top = bench.sharded_headline(sh, 8)
    assert top["speedup"] == 4.5 and top["rccl_ranks"] == 8 and top["ranks"] == 8 and "N=32768" in top["workload"] and top["scaling"] == "strong"
    assert bench.sharded_headline({"cfg3": {"error": "watchdog"}}, 8)["error"] == "watchdog"
    ident = bench.cpu_identity()
    assert "cpu_model" in ident and ident["logical_cpus"] >= 1
