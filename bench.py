#!/usr/bin/env python
"""
bench.py -- BASELINE.json's headline metric on MI355X:
    log-marginal-likelihood + gradient evaluations per second, MOSM C=4 Q=3 N=8192 (configs[1]), exact GP, fp64.

A "step" is one iteration of the reference's training loop (mogptk/model.py:563-566): `gpr.Exact.loss()` -- term table upload, Gram build
(+noise +jitter), Cholesky, triangular inverse, alpha / log-det, K^-1, gradient-moment pass, moments back to the host, host chain rule to
the raw-parameter gradient -- followed by an Adam update of the raw parameters (lr 1e-6: the parameters CHANGE every step, so nothing the
host memoises between evaluations at equal parameters can hit).  X and y are resident in HBM before the timed region (model creation); only
the O(C^2 Q) parameter table goes host->device per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5] [--mode replicas|sharded]

The default line (cfg2) also carries, outside its timed region:
  configs       the other BASELINE.json configurations on this GPU: cfg3 (MOSM C=8 Q=5 N=32768 LML+gradient), cfg4 (CSM C=4 Q=3 N=16384
                predict_f at 4096 points), cfg5 (Titsias N=100000 M=2048 ELBO+gradient): ms_per_step, roofline fraction, steps
  cpu_baseline  the torch-CPU port of the reference's op sequence (oracle/torch_port.py) timed on this box's host cores AT N = 8192
  roofline / gram_hbm / moments_hbm / stages_ms_per_eval from HIP events inside the timed region
`--config cfgN` makes that configuration the timed step instead (same JSON shape).

N > 1: one rank per GPU.  Under a launcher (torch.distributed.run: WORLD_SIZE / RANK / LOCAL_RANK / MASTER_* in the environment) this process is one
rank; as a PLAIN process (`python bench.py --gpus N`, WORLD_SIZE unset) it starts the N ranks itself (self_launch: rank r on GPU r, rendezvous on
127.0.0.1) and exits non-zero with a one-line reason when the node shows fewer than N GPUs -- it never prints a 1-GPU line for an N-GPU request.
The N > 1 line carries `sharded_cfg3` at top level (ONE N = 32768 evaluation split over the ranks: ms_one_gpu, ms_sharded, speedup, rccl_ranks,
rel_loss / rel_grad) next to the replicas `value`.
  replicas (default)  `value` = aggregate evals/s of N independent replicas of the workload (one 13 ms evaluation does not pay for an
                      exchange per pivot block; N GPUs are best used as N evaluations: restarts, models); scaling "weak".  BEFORE the timed
                      region every rank runs the `sharded` probes -- the north_star's split of ONE evaluation over the GPUs
                      (mogp_exact_eval_sharded: owned Gram / moment tiles, one RCCL all-gather per 512-wide pivot block issued by the
                      library on its own stream): cfg3 first, then cfg2 and the data-parallel sparse bound (cfg5).  Each probe is a child
                      process group of its own under its own watchdog, so a stuck collective costs that probe, not the line.
  sharded             `value` = evals/s of ONE evaluation spread over all ranks; scaling "strong" (`--config cfg3 --mode sharded` is
                      configs[2] as BASELINE.json words it).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X vendor figure for FP64 matrix (v_mfma_f64_16x16x4_f64); see DESIGN.md section 3
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    # name: (kind, C, Q, N, extra, description)
    "cfg2": ("exact", 4, 3, 8192, None, "MOSM C=4 Q=3 N=8192 exact GP LML+gradient (BASELINE.json configs[1])"),
    "cfg3": ("exact", 8, 5, 32768, None, "MOSM C=8 Q=5 N=32768 exact GP LML+gradient (BASELINE.json configs[2])"),
    "cfg4": ("predict", 4, 3, 16384, 4096, "CSM C=4 Q=3 N=16384 exact GP predict_f at 4096 test points (BASELINE.json configs[3])"),
    "cfg5": ("titsias", 4, 3, 100000, 2048, "Titsias MOSM C=4 Q=3 N=100000 M=2048 ELBO+gradient (BASELINE.json configs[4])"),
}
METRICS = {
    "exact": ("log-marginal-likelihood+grad evals/sec, MOSM C=%d N=%d; 1/2/4/8 GPU", "evals/s"),
    "predict": ("predict_f (mean+variance, %d test points) calls/sec, CSM C=%d N=%d", "calls/s"),
    "titsias": ("Titsias ELBO+grad evals/sec, MOSM C=%d N=%d M=%d", "evals/s"),
}
STEP_NOTE = {
    "exact": "loss() = LML + gradient of every raw parameter, then one Adam update (lr 1e-6) of the raw parameters: the reference's training iteration (mogptk/model.py:563-566)",
    "titsias": "loss() = ELBO + gradient of every raw parameter and of the inducing inputs, then one Adam update (lr 1e-6): the reference's training iteration",
    "predict": "one predict_f call (factorisation + predictive mean and variance); no parameters change",
}
PROBES = ("cfg3", "cfg2", "cfg5", "cfg5_weak")


def build_mosm(N, C, Q, device):
    from mogptk_amd import gpr, synth
    if device is not None:
        gpr.config.device = device
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    return m


def build_model(cfg, device, n_override=None):
    """-> (model, step callable, algorithmic flops per step)"""
    from mogptk_amd import gpr, synth
    kind, C, Q, N, extra, _ = CONFIGS[cfg]
    if n_override:
        N = n_override
    if device is not None:
        gpr.config.device = device
    if kind == "exact":
        m = build_mosm(N, C, Q, device)
        return m, (lambda: m.loss()), float(N) ** 3                                   # POTRF + TRTRI + LAUUM = N^3
    if kind == "predict":
        S = extra
        X, y = synth.make_data(N, C)
        h = synth.csm_hypers(C, Q)
        k = gpr.MixtureKernel(gpr.CrossSpectralKernel(output_dims=C, input_dims=1, Rq=1), Q)
        for q in range(Q):
            k[q].amplitude.assign(h["amplitude"][q]); k[q].mean.assign(h["mean"][q])
            k[q].variance.assign(h["variance"][q]); k[q].shift.assign(h["shift"][q])
        m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
        m.likelihood.scale.assign(h["scale"])
        Xs = synth.test_inputs(S, C)
        return m, (lambda: m.predict_f(Xs)), float(N) ** 3 / 3.0 + float(N) ** 2 * S        # Cholesky + the N x S triangular solve
    M = extra
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    s = float(np.mean(h["scale"]))
    m = gpr.Titsias(k, X, y, Z=[M // C] * C, variance=s ** 2)
    m.likelihood.scale.assign(s)
    # forward: v = L^-1 Kuf (M^2 N) + Q = v v^T (M^2 N); backward: (L^-T (I - Pq)) v (2 M^2 N); the M^3 terms are 1 %.  (Until round 4 the backward
    # part was (I - Pq) v followed by an M x N solve with L^T, and this count was 5 M^2 N: the solve is gone from the algorithm, so it is gone from here.)
    return m, (lambda: m.loss()), 4.0 * float(M) ** 2 * N + 5.0 * float(M) ** 3


def training_step(m, run_step, kind):
    """the timed step: the evaluation, then -- where there is a gradient -- one Adam update of the raw parameters (reference
    mogptk/model.py:563-566: loss, backward, optimizer.step), so that every step sees new parameters"""
    if kind == "predict":
        return run_step
    from mogptk_amd.model import _Adam
    opt = _Adam(list(m.parameters()), lr=1e-6)

    def step():
        run_step()
        opt.step()
    return step


def cpu_baseline(N, C, Q, budget_s=240.0):
    """torch-CPU port of the reference op sequence (oracle/torch_port.py), ONE evaluation of the same workload at the same N, timed
    directly.  Only when a probe at N = 2048 predicts more than `budget_s` for it (a very slow or very busy host) is the largest
    power-of-two N that fits timed instead and scaled by N^3 -- `extrapolated` says which."""
    import torch
    from oracle import torch_port
    from mogptk_amd import synth, gpr
    cores = torch.get_num_threads()

    def one(n):
        X, y = synth.make_data(n, C)
        h = synth.mosm_hypers(C, Q)
        raws = {}
        for name in ("weight", "mean", "variance", "scale"):
            raws[name] = gpr.Parameter(h[name], lower=1e-8).data
        raws["delay"], raws["phase"] = h["delay"], h["phase"]
        t = time.perf_counter()
        torch_port.mosm_loss_and_grad(X, y, raws, C)
        return time.perf_counter() - t

    n = min(2048, N)
    t = one(n)
    predicted = t * (N / n) ** 3 * 1.3            # the reference grows 6.4-8.8x per doubling here
    if n < N and predicted <= budget_s:
        n = N
        t = one(N)
    else:
        while n * 2 <= N and t * 8.5 * 1.3 <= budget_s:
            n *= 2
            t = one(n)
    if n == N:
        return dict(value=1.0 / t, unit="evals/s", cores=cores, torch_threads=cores, kind="port", extrapolated=False, seconds=t,
                    sample="1 LML+grad eval of the same workload (MOSM C=%d Q=%d N=%d), torch-CPU fp64 port of the "
                           "reference op sequence, timed directly: %.1f s" % (C, Q, N, t))
    scale = (N / n) ** 3
    return dict(value=1.0 / (t * scale), unit="evals/s", cores=cores, torch_threads=cores, kind="port", extrapolated=True, seconds=t * scale,
                sample="1 eval at N=%d took %.1f s; a direct evaluation at N=%d was predicted to exceed %.0f s on this host; "
                       "extrapolated by N^3 (x%.0f)" % (n, t, N, budget_s, scale))


def timed_region(step, steps, warmup, dist=None, sync=lambda: None, device="cpu", per_step=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both sides; returns the MAX
    over ranks of the elapsed wall time (seconds).  `dist` is torch.distributed (initialised) or None.  per_step (a list): this rank's
    wall time of every step is appended -- a step ends with its result on the host, so the stamps cost nothing and split the region exactly."""
    import torch
    for _ in range(warmup):
        step(-1)
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    tp = t0
    for i in range(steps):
        step(i)
        if per_step is not None:
            tn = time.perf_counter()
            per_step.append(tn - tp)
            tp = tn
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def aggregate_value(world, steps, dt, sharded=False):
    """whole-job units per second: `steps` evaluations per rank in `dt` (max over ranks) -- replicas: every rank ran its own; sharded: all
    ranks ran the same ones together"""
    return (1 if sharded else world) * steps / dt


def config_entry(cfg, ms_per_step, steps, algo_flops):
    """one entry of the `configs` object: a BASELINE.json configuration timed on this GPU outside the headline's timed region"""
    kind, C, Q, N, extra, desc = CONFIGS[cfg]
    tflops = algo_flops / (ms_per_step * 1e-3) / 1e12
    return {"workload": desc, "ms_per_step": ms_per_step, "steps": steps, "value": 1e3 / ms_per_step, "unit": METRICS[kind][1],
            "achieved_tflops": tflops, "frac": tflops / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops": algo_flops, "step": STEP_NOTE[kind]}


def extra_configs(device, names=("cfg3", "cfg4", "cfg5"), steps=None):
    """the other BASELINE.json configurations on this GPU (SURVEY.md 8d asks for predict_f time at cfg4 and ELBO+gradient at cfg5 next to the
    headline): one warm-up step, then `steps` timed steps each; the models are freed again before the next one is built"""
    import gc
    out = {}
    for cfg in names:
        kind = CONFIGS[cfg][0]
        k = (steps or {}).get(cfg, 3 if cfg == "cfg3" else 5)
        try:
            m, run_step, flops = build_model(cfg, device)
            step = training_step(m, run_step, kind)
            step()                                     # device handle, workspaces, first-call allocations
            step()
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            ms = 1e3 * (time.perf_counter() - t0) / k   # every call returns after its one stream sync: wall clock is device-complete
            out[cfg] = config_entry(cfg, ms, k, flops)
            h = getattr(m, "_handle", None)
            if h is not None:
                if kind in ("exact", "predict") and hasattr(h, "schedule"):        # which schedule the last call ran on (mogp_model_schedule): a fallback shows
                    sc = h.schedule()
                    out[cfg]["dataflow_kernel"] = sc["dataflow"]
                    out[cfg]["fell_back"] = sc["dataflow_fell_back"] or sc["chain_fell_back"]
                h.close()
            del m, run_step, step
        except Exception as e:          # a report, never a reason to lose the headline
            out[cfg] = {"error": repr(e)}
        gc.collect()
    return out


# ---- sharded probes: each one a process group of its own (one child per rank), so that a collective that never returns is killed with
# ---- its processes and costs one entry of the line -------------------------------------------------------------------------------------
def probe_child(name, reps):
    """runs inside the child: rank / world from the environment, a fresh NCCL group on MASTER_PORT; rank 0 prints one JSON line"""
    import datetime
    import torch
    import torch.distributed as dist
    import mogptk_amd
    from mogptk_amd import gpr, _lib
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if name in ("dummy", "hang"):                 # CPU self-test of the orchestration (tests/test_bench_dist_cpu.py): no GPU, gloo
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
        t = torch.tensor([1.0, rank + 1.0], dtype=torch.float64)
        dist.all_reduce(t)
        if name == "hang":
            if rank == 0:                         # a finished stage's line must survive the watchdog
                print(json.dumps({"rccl_ranks": int(t[0].item()), "stage": "before the hang"}), flush=True)
            if rank == world - 1:
                time.sleep(3600)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"rccl_ranks": int(t[0].item()), "rank_sum_ok": int(t[1].item()) == world * (world + 1) // 2}), flush=True)
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=180))
    gpr.config.device = local_rank
    if name == "cfg3":
        m = build_mosm(32768, 8, 5, local_rank)
    elif name == "cfg2":
        m = build_mosm(8192, 4, 3, local_rank)
    elif name == "cfg5":
        m, _, _ = build_model("cfg5", local_rank)
    elif name == "cfg5_weak":
        m, _, _ = build_model("cfg5", local_rank, n_override=100000 * world)
    else:
        raise ValueError(name)

    def sync():
        torch.cuda.synchronize()

    def emit(d):
        """rank 0: one JSON line per finished stage -- a later stage that hangs is killed by the parent's watchdog, which keeps the last line"""
        if rank == 0:
            sys.stdout.flush()
            import ctypes
            ctypes.CDLL(None).fflush(None)
            print(json.dumps(d), flush=True)

    mogptk_amd.use_single_device()
    l0 = float(m.loss()); g0 = [p.grad.copy() for p in m.parameters()]
    sync(); t = time.perf_counter()
    for _ in range(reps):
        m.loss()
    sync(); t_single = (time.perf_counter() - t) / reps
    if name in ("cfg3", "cfg2"):
        # the sharded evaluations run on a handle of their own whose FIRST evaluation is sharded: its work matrix comes in the owned-rows form (physical
        # memory under this rank's tile rows only, no second matrix: include/mogp_hip.h, mogp_model_work_bytes); the one-GPU model's device memory goes first
        m._handle = None
        import gc
        gc.collect()
        m = build_mosm(32768, 8, 5, local_rank) if name == "cfg3" else build_mosm(8192, 4, 3, local_rank)
    comm = mogptk_amd.use_distributed()
    comm.force = True
    seen, rsum = _lib.comm_selftest(local_rank)            # one all-reduce issued by the library over ITS communicator
    res = dict(ms_one_gpu=1e3 * t_single, rccl_ranks=seen, rank_sum_ok=(rsum == world * (world + 1) // 2), transport=comm.transport, stage="communicator up")
    emit(res)

    def timed_sharded(env):
        """one sharded evaluation for the parity numbers, then `reps` timed ones, under the given switches (read per evaluation by the library)"""
        old_env = {k_: os.environ.get(k_) for k_ in env}
        os.environ.update(env)
        try:
            lv = float(m.loss()); gv = [p.grad.copy() for p in m.parameters()]
            dist.barrier(); sync(); t0 = time.perf_counter()
            for _ in range(reps):
                m.loss()
            sync(); dist.barrier()
            ts = (time.perf_counter() - t0) / reps
            free_b, total_b = torch.cuda.mem_get_info()
            out = {"ms_sharded": 1e3 * ts, "rel_loss": abs(lv - l0) / abs(l0),
                   "rel_grad": max(float(np.max(np.abs(b - c)) / np.max(np.abs(c))) for b, c in zip(gv, g0)),
                   # what this rank's device holds with the model up (the training set, this rank's rows of the work matrix, panel and exchange buffers)
                   "device_bytes_in_use_per_rank": int(total_b - free_b)}
            hh = getattr(m, "_handle", None)
            if hh is not None and hasattr(hh, "work_bytes"):
                out["work_matrix_bytes_backed_per_rank"], out["work_matrix_bytes_whole"] = hh.work_bytes()
            return out
        finally:
            for k_, v_ in old_env.items():
                if v_ is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v_

    if name == "cfg3":
        # Three forms of the exchange (mogp_api.hip:sharded_inverse), the most conservative FIRST, a line after each: the whole panel of a pivot
        # block in ONE message on the critical stream (rounds 1-4); the default -- two messages, the large one on a communication stream underneath the
        # block's inversion; and the pivot block inverted ONCE, by its owner, with an all-reduce of the factor.  The headline numbers are the default's
        # (the one-message form's if the default did not finish).
        variants = {}
        for vname, env in (("one_message", {"MOGP_SHARD_SPLIT": "0"}), ("split", {}), ("factor_once", {"MOGP_SHARD_FACTOR_ONCE": "1"})):
            if vname == "factor_once" and world == 1:
                continue                                 # (one rank owns every pivot block: nothing to compare)
            try:
                variants[vname] = timed_sharded(env)
            except Exception as e:
                variants[vname] = {"error": repr(e)}
            best = variants.get("split") if "ms_sharded" in variants.get("split", {}) else variants.get("one_message", {})
            if "ms_sharded" in best:
                res.update(ms_sharded=best["ms_sharded"], speedup=1e3 * t_single / best["ms_sharded"], evals_per_s_sharded=1e3 / best["ms_sharded"],
                           rel_loss=best["rel_loss"], rel_grad=best["rel_grad"], headline_variant="split" if best is variants.get("split") else "one_message")
            res["variants"] = variants
            res["stage"] = "variant %s done" % vname
            emit(res)
    else:
        v = timed_sharded({})
        res.update(ms_sharded=v["ms_sharded"], speedup=1e3 * t_single / v["ms_sharded"], evals_per_s_sharded=1e3 / v["ms_sharded"], rel_loss=v["rel_loss"],
                   rel_grad=v["rel_grad"], stage="timed")
        emit(res)
    h = getattr(m, "_handle", None)
    if name in ("cfg3", "cfg2") and h is not None:         # where one sharded evaluation spends its time (HIP events, summed over the pivot blocks)
        h.set_profiling(True)
        m.loss()
        ex, ser, nxt, blk, exc, stall = [float(v) for v in h.shard_stage_ms()]
        h.set_profiling(False)
        res.update(exchange_ms=ex, serial_ms=ser, next_cols_ms=nxt, bulk_ms=blk, exchange_comm_stream_ms=exc, wait_for_comm_stream_ms=stall,
                   split_note="critical stream: exchange (the pivot block's own rows) + serial (Schur block inversion and panels, repeated on every "
                              "rank) + next_cols; underneath, the rank's share of the bulk update on the bulk stream and the rest of the panel "
                              "(exchange_comm_stream) on the communication stream -- wait_for_comm_stream is what the critical stream still waited for it")
    if name.startswith("cfg5"):
        res["N"] = 100000 * (world if name == "cfg5_weak" else 1)
    res["stage"] = "complete"
    emit(res)
    mogptk_amd.use_single_device()
    mogptk_amd.shutdown_distributed()
    dist.destroy_process_group()


def run_probe(name, port, timeout_s, reps):
    """parent side, on every rank: start this rank's child of probe `name`, wait at most timeout_s, kill its process group otherwise;
    -> the child's JSON (rank 0; other ranks get {}) or {"error": ...}"""
    import signal
    env = dict(os.environ, MASTER_PORT=str(port))
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("RANK", "0"); env.setdefault("LOCAL_RANK", "0"); env.setdefault("WORLD_SIZE", "1")     # a plain `python bench.py --shard-probe`
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # the child group's rank 0 hosts its own store on `port`
    env.pop("TORCHELASTIC_RUN_ID", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--probe-child", name, "--probe-reps", str(reps)]
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        so, se = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except Exception:
            p.kill()
        so = ""
        try:
            so, _ = p.communicate(timeout=10)
        except Exception:
            pass
        lines = [l for l in (so or "").splitlines() if l.startswith("{")]
        r = {}
        if lines:                                   # what the child had finished before it hung: every stage prints a line
            try:
                r = json.loads(lines[-1])
            except Exception:
                r = {}
        r["error"] = "the probe did not finish within %.0f s (watchdog); its processes were killed%s" % (
            timeout_s, (" -- the numbers are those of the stages it had finished (last: %s)" % r.get("stage")) if r else "")
        return r
    took = time.perf_counter() - t0
    if p.returncode != 0:
        return {"error": "probe exited with code %d: %s" % (p.returncode, (se or "")[-400:])}
    lines = [l for l in (so or "").splitlines() if l.startswith("{")]
    if not lines:
        return {}
    r = json.loads(lines[-1])
    r["probe_wall_s"] = took
    return r


def run_probes(names, base_port, timeout_s, reps=None):
    """every probe in turn; once one has run into its watchdog the rest are skipped (a communicator that hangs once hangs again, and the
    driver's clock is running)"""
    out, hung = {}, False
    for i, name in enumerate(names):
        if hung:
            out[name] = {"error": "skipped: an earlier probe ran into its watchdog"}
            continue
        out[name] = run_probe(name, base_port + 1 + i, timeout_s, (reps or {}).get(name, 2 if name in ("cfg3", "cfg5_weak") else 3))
        hung = "watchdog" in str(out[name].get("error", ""))
    return out


def device_count():
    """GPUs this process can see, from the library the benchmark measures (mogp_device_count = hipGetDeviceCount)"""
    from mogptk_amd import _lib
    return int(_lib.lib().mogp_device_count())


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv, selftest=False):
    """`python bench.py --gpus N` as a PLAIN process (no torch.distributed.run around it, WORLD_SIZE unset): start the N ranks here -- one
    process per GPU, rank r pinned to GPU r, rendezvous on 127.0.0.1 and a free port of its own -- and wait for them.  Rank 0 prints the line on
    the inherited stdout.  Fewer than N visible GPUs is an error with a one-line reason (exit code 2), never a silent 1-GPU run that reports
    n_gpus 1; a rank that dies takes the others with it and its exit code becomes this process's."""
    import signal
    if not selftest:
        try:
            have = device_count()
        except Exception as e:
            print("bench.py: --gpus %d: cannot count the GPUs (%r)" % (n, e), file=sys.stderr)
            return 2
        if have < n:
            print("bench.py: --gpus %d asked for, but this node shows %d GPU%s (hipGetDeviceCount); not starting" % (n, have, "" if have == 1 else "s"),
                  file=sys.stderr)
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MOGP_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # RCCL across processes needs dmabuf IPC on these hosts
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, start_new_session=True))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.05)
            for p in list(live):
                c = p.poll()
                if c is None:
                    continue
                live.remove(p)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    print("bench.py: rank %d of the self-launched job exited with code %d; stopping the other ranks" % (procs.index(p), c), file=sys.stderr)
                    for q in live:
                        try:
                            os.killpg(q.pid, signal.SIGTERM)
                        except Exception:
                            pass
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except Exception:
                    pass
    return rc


def cpu_identity():
    """CPU model string and the thread counts behind `cpu_baseline.cores` (core count and CPU model printed)"""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    out = {"cpu_model": model, "logical_cpus": os.cpu_count()}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return out


def sharded_headline(sharded, world):
    """the north_star's multi-GPU number, first-class on the N > 1 line: ONE configs[2] evaluation (MOSM C=8 Q=5 N=32768) split over the ranks by
    mogp_exact_eval_sharded against the same evaluation on one GPU of the same job (`value` beside it is N independent replicas of configs[1])"""
    r = dict((sharded or {}).get("cfg3") or {})
    keys = ("ms_one_gpu", "ms_sharded", "speedup", "evals_per_s_sharded", "rccl_ranks", "rank_sum_ok", "transport", "rel_loss", "rel_grad",
            "exchange_ms", "serial_ms", "next_cols_ms", "bulk_ms", "exchange_comm_stream_ms", "wait_for_comm_stream_ms", "variants", "headline_variant", "stage", "error")
    out = {"workload": CONFIGS["cfg3"][5], "ranks": world, "scaling": "strong"}
    out.update({k: r[k] for k in keys if k in r})
    if "speedup" in r:
        out["target"] = "north_star: >= 3.5x at 8 GPUs"
    return out


def selftest_main(a, rank, world):
    """the orchestration of an N > 1 run without a GPU or a native call (tests/test_bench_dist_cpu.py): gloo ranks, the probe mechanism with its
    `dummy` probe (one all-reduce of (1, rank + 1) in a child process group), the timed region around a sleeping step, the line's top level"""
    dist = None
    sharded = None
    if world > 1:
        base_port = int(os.environ.get("MASTER_PORT", "29655")) + 20
        sharded = run_probes(["dummy"], base_port, a.probe_timeout)
        sharded["ranks"] = world
        import torch.distributed as dist
        dist.init_process_group("gloo")
    steps = a.steps if a.steps is not None else 5
    warmup = a.warmup if a.warmup is not None else 1
    dt = timed_region(lambda i: time.sleep(0.002), steps, warmup, dist)
    if rank == 0:
        out = {"metric": "selftest (no GPU work)", "value": aggregate_value(world, steps, dt), "unit": "steps/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic", "config": {"workload": "selftest", "parallelism": "replicas x%d" % world,
                                                               "self_launched": os.environ.get("MOGP_BENCH_SELF_LAUNCHED") == "1"}}
        if sharded is not None:
            out["sharded"] = sharded
            out["rccl_ranks"] = (sharded.get("dummy") or {}).get("rccl_ranks")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="replicas", choices=["replicas", "sharded"])
    ap.add_argument("--n", type=int, default=None, help="override the configuration's N (development)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained", type=int, default=200, help="steps of the longer sample taken behind the timed region of the headline configuration (0: none)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (cfg3 / cfg4 / cfg5 on this GPU)")
    ap.add_argument("--shard-probe", action="store_true", help="also run the sharded probes at --gpus 1 (1-rank RCCL group)")
    ap.add_argument("--no-shard-probe", action="store_true")
    ap.add_argument("--probes", default=",".join(PROBES), help="which sharded probes, in order")
    ap.add_argument("--probe-timeout", type=float, default=180.0, help="watchdog of EACH sharded probe in seconds")
    ap.add_argument("--probe-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--probe-reps", type=int, default=3, help=argparse.SUPPRESS)
    ap.add_argument("--selftest", action="store_true", help=argparse.SUPPRESS)     # the launch / probe / timing orchestration on CPU ranks (gloo, a dummy step): tests/test_bench_dist_cpu.py
    a = ap.parse_args()
    if a.probe_child:
        probe_child(a.probe_child, a.probe_reps)
        return
    if a.gpus < 1:
        ap.error("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # a plain `python bench.py --gpus N`: nobody started the ranks, so this process does (and fails loudly if the node has fewer GPUs)
        sys.exit(self_launch(a.gpus, sys.argv[1:], selftest=a.selftest))
    kind, C, Q, N, extra, desc = CONFIGS[a.config]
    if a.n:
        N = a.n
    big = a.config != "cfg2"
    steps = a.steps if a.steps is not None else (5 if big else 20)
    warmup = a.warmup if a.warmup is not None else (1 if big else 3)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d rank%s (WORLD_SIZE); the line reports n_gpus = %d" % (a.gpus, world, "" if world == 1 else "s", world),
              file=sys.stderr)
    if a.selftest:
        selftest_main(a, rank, world)
        return
    if world > 1:
        have = device_count()
        if local_rank >= have:
            print("bench.py: rank %d is pinned to GPU %d but this node shows %d GPU%s" % (rank, local_rank, have, "" if have == 1 else "s"), file=sys.stderr)
            sys.exit(2)
    sharded_mode = a.mode == "sharded"       # exact / predict: one evaluation's tiles over the ranks; titsias: its data points over the ranks

    # ---- the sharded probes FIRST (N > 1, replicas mode): nothing of this process is on the GPU yet, every probe is a group of child
    # ---- processes with a watchdog of its own; a probe that hangs is killed and the line goes on
    sharded = None
    want_probe = (world > 1 or a.shard_probe) and not a.no_shard_probe and kind == "exact" and not sharded_mode
    if want_probe:
        base_port = int(os.environ.get("MASTER_PORT", "29655")) + 20
        names = [n for n in a.probes.split(",") if n and not (n == "cfg5_weak" and world == 1)]
        sharded = run_probes(names, base_port, a.probe_timeout)
        sharded["ranks"] = world

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import torch
    import mogptk_amd
    from mogptk_amd import _lib

    m, run_step, algo_flops = build_model(a.config, local_rank, a.n)
    train_step = training_step(m, run_step, kind)

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    pd = dist
    if sharded_mode and dist is None:           # one GPU: a 1-rank RCCL group exercises the same code path
        import torch.distributed as dist1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
        dist1.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        pd = dist1

    run_step()                    # creates the device handle (X, y resident in HBM) before anything is timed
    h = m._handle
    rccl = None
    if sharded_mode:
        comm = mogptk_amd.use_distributed()
        comm.force = True
        rccl = _lib.comm_selftest(local_rank)
        run_step()
    stage = np.zeros(_lib.ST_COUNT)
    acc = dict(flops=0.0, launches=0, nprof=0)
    PROFILE_EVERY = max(20, steps)          # HIP events around every GEMM launch cost ~5 % of a step: ONE timed step (the first) carries them

    def step(i):
        prof = i >= 0 and (i % PROFILE_EVERY) == 0 and kind == "exact" and not sharded_mode
        h.set_profiling(prof)
        train_step()
        if prof:
            ms, nl, fl = h.stage_ms()
            stage[:] += ms
            acc["flops"] += fl
            acc["launches"] += nl
            acc["nprof"] += 1

    step_times = []
    dt = timed_region(step, steps, warmup, dist, sync, "cuda" if dist is not None else "cpu", per_step=step_times)
    gemm_flops, gemm_launches, nprof = acc["flops"], acc["launches"], max(acc["nprof"], 1)
    h.set_profiling(False)
    # a longer sample of the same step right behind the timed region (the chip's clocks take tens of ms of load to settle, and K = 20 steps of
    # configs[1] are 0.2 s): reported beside `value`, never instead of it
    sustained = None
    if a.sustained > 0 and not big and not sharded_mode:
        sus_times = []
        dts = timed_region(lambda i: train_step(), a.sustained, 0, dist, sync, "cuda" if dist is not None else "cpu", per_step=sus_times)
        sustained = {"steps": a.sustained, "ms_per_step": 1e3 * dts / a.sustained, "value": aggregate_value(world, a.sustained, dts),
                     "median_ms_per_step": 1e3 * float(np.median(sus_times)), "max_ms_per_step": 1e3 * float(np.max(sus_times)),
                     "note": "the same step, %d more times behind the timed region" % a.sustained}
    # the Gram tile kernel BY ITSELF: inside the gradient evaluation's dataflow schedule it is two launches, the larger one on the bulk stream underneath the first
    # chain kernel and next to the resident dataflow kernel's polling workgroups (which is where it is hidden, and why its duration there -- gram_kernel in
    # stages_ms_per_eval -- says little about the kernel); an LML-only evaluation builds the same matrix in ONE launch with nothing beside it
    gram_alone_ms = None
    # (how the LAST evaluation was scheduled is read here, behind the timed steps: the LML-only evaluations below are not gradient evaluations)
    sched_timed = h.schedule() if (kind in ("exact", "predict") and not sharded_mode and hasattr(h, "schedule")) else None
    if rank == 0 and kind == "exact" and not sharded_mode and hasattr(m, "log_marginal_likelihood"):
        try:
            h.set_profiling(True)
            ts = []
            for _ in range(12):
                m.log_marginal_likelihood()
                ts.append(float(h.stage_ms()[0][_lib.ST_GRAM_KERNEL]))
            gram_alone_ms = float(np.median(ts[2:]))
        except Exception:
            gram_alone_ms = None
        finally:
            h.set_profiling(False)
        train_step()                  # the handle's "last evaluation" is a gradient evaluation again (inverse_fraction, schedule)
    if sharded_mode:
        mogptk_amd.use_single_device()

    out = None
    if rank == 0:
        # SURVEY 8d's statistic: the MEDIAN of the K timed steps (K >= 10) is the headline at N = 1 -- the mean of the same K steps (what the
        # bracketed region divides out to) stays beside it; at N > 1 the region's max-over-ranks time is the only number all ranks share
        mean_ms_per_step = 1e3 * dt / steps
        value_mean = aggregate_value(world, steps, dt, sharded_mode)
        use_median = world == 1 and not sharded_mode and len(step_times) >= 10
        ms_per_step = 1e3 * float(np.median(step_times)) if use_median else mean_ms_per_step
        value = 1e3 / ms_per_step if use_median else value_mean
        # The GEMM launches of one evaluation run on up to five streams at once (potri.hip), so the sum of their durations exceeds the
        # wall-clock time they occupy: `span` prices them over the factorisation + inversion stage, `per_launch` over the sum of their
        # own durations (what a kernel trace averages to); `frac` -- the headline -- prices the ALGORITHMIC flops over the whole step.
        gemm_s = stage[_lib.ST_GEMM_KERNEL] * 1e-3
        span_s = (stage[_lib.ST_POTRF] + stage[_lib.ST_TRTRI] + stage[_lib.ST_SOLVE] + stage[_lib.ST_LAUUM]) * 1e-3
        # an evaluation whose gradient reads only a band of Kj^-1 forms only those tiles (mogp_model_inverse_fraction; 1.0 at every BASELINE
        # config): the MFMA rate is then priced on the flops actually issued, never on the full count
        inv_frac = h.inverse_fraction() if (kind == "exact" and not sharded_mode and hasattr(h, "inverse_fraction")) else 1.0
        if inv_frac < 1.0 and acc["nprof"] > 0:
            algo_flops = min(algo_flops, acc["flops"] / max(acc["nprof"], 1))
        achieved = algo_flops / (ms_per_step * 1e-3) / 1e12
        if sharded_mode:
            achieved /= world            # per GPU
        gram_bytes = 4.0 * N * (N + 1)            # lower triangle written / read once
        # the two HBM-bound passes: algorithmic bytes over the duration of the tile kernel alone (HIP events around that one launch)
        gram_sched_gbs = gram_bytes * nprof / (stage[_lib.ST_GRAM_KERNEL] * 1e-3) / 1e9 if stage[_lib.ST_GRAM_KERNEL] > 0 else None
        gram_gbs = gram_bytes / (gram_alone_ms * 1e-3) / 1e9 if gram_alone_ms else gram_sched_gbs
        mom_gbs = gram_bytes * nprof / (stage[_lib.ST_MOMENT_KERNEL] * 1e-3) / 1e9 if stage[_lib.ST_MOMENT_KERNEL] > 0 else None
        traffic, traffic_src, traffic_eval = None, None, None
        sched = sched_timed
        for tf in ("r6_pmc_traffic_stream_schedule.json", "r5_pmc_traffic_stream_schedule.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json"):     # L2<->fabric bytes per k_gemm launch from the committed rocprofv3 --pmc passes of this command
            try:
                with open(os.path.join(ROOT, "profiles", tf)) as f:
                    t = json.load(f)
                if a.config == "cfg2" and not sharded_mode:
                    traffic, traffic_src = t["bytes_per_launch"], "profiles/%s: %s" % (tf, t["source"])
                    traffic_eval = t.get("fetch_bytes_per_eval", 0.0) + t.get("write_bytes_per_eval", 0.0)
                break
            except Exception:
                continue
        if kind == "exact":
            metric = METRICS[kind][0] % (C, N)
        elif kind == "predict":
            metric = METRICS[kind][0] % (extra, C, N)
        else:
            metric = METRICS[kind][0] % (C, N, extra)
        if a.config == "cfg2":
            metric = "log-marginal-likelihood+grad evals/sec, MOSM C=4 N=8192; 1/2/4/8 GPU"      # BASELINE.json's wording
        if world == 1 and not sharded_mode:
            par = "1 gpu"
        elif sharded_mode:
            par = ("data-parallel x%d (points cyclic over the ranks, RCCL all-reduce of the M x M sums)" % world if kind == "titsias" else
                   "sharded x%d (tile rows cyclic, RCCL all-gather per pivot block)" % world)
        else:
            par = ("replicas x%d: `value` is the aggregate of %d INDEPENDENT evaluations streams, one per GPU, no collective on the data path; "
                   "ONE evaluation split over the GPUs (the north_star's distributed Cholesky) is measured in `sharded`" % (world, world))
        out = {
            "metric": metric, "value": value, "unit": METRICS[kind][1], "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "statistic": ("median of the %d timed steps (SURVEY.md 8d)" % steps) if use_median else "steps / bracketed region (max over ranks)",
            "mean_ms_per_step": mean_ms_per_step, "value_mean": value_mean, "timed_region_s": dt,
            "min_ms_per_step": 1e3 * float(np.min(step_times)) if step_times else None, "max_ms_per_step": 1e3 * float(np.max(step_times)) if step_times else None,
            "higher_is_better": True, "scaling": "strong" if sharded_mode else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc if not a.n else desc + " [N overridden: %d]" % N, "channels": C, "Q": Q, "N": N, "parallelism": par,
                       "inverse_tiles_formed": inv_frac,
                       # how the factorisation + inversion was scheduled (mogp_model_schedule): a fallback to the stream schedule or to the
                       # launch-per-step chain (a hand-off timed out: GPU shared with another process) would show here, not only as a slowdown
                       "dataflow_kernel": None if sched is None else sched["dataflow"],
                       "chain_kernel": None if sched is None else sched["chain_kernel"],
                       "fell_back": None if sched is None else (sched["dataflow_fell_back"] or sched["chain_fell_back"]),
                       "dataflow_timeouts": None if sched is None else sched.get("dataflow_timeouts"),
                       "step": STEP_NOTE[kind], "device": _lib.device_name(local_rank)},
            "roofline": {"bound": "mfma", "kernel": ("k_flow: the resident tile-dataflow kernel, k_gemm's k loop (fp64 v_mfma_f64_16x16x4_f64)"
                                                     if sched and sched["dataflow"] else "k_gemm (fp64 v_mfma_f64_16x16x4_f64)"), "achieved": achieved,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "basis": "algorithmic flops of one step (SURVEY.md 8d: %.3e) / ms_per_step%s" % (algo_flops, " / ranks" if sharded_mode else ""),
                         "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src},
        }
        if sched and sched["dataflow"] and a.config == "cfg2" and not sharded_mode:
            # rocprofv3 --pmc serialises dispatches, and the dataflow kernel normally co-operates with the chain kernels (under --pmc an evaluation
            # falls back to the stream schedule).  Its traffic is therefore counted with the kernel running ALONE on the replay plan
            # (mogp_model_flow_replay: chain counters preset, same tile products, same Kj^-1 bit for bit -- tools/flow_replay.py, tools/pmc_flow.py);
            # the stream schedule's launches of the same products stay beside it.
            stream = {"bytes_per_eval": traffic_eval, "bytes_per_launch": traffic, "source": traffic_src} if traffic is not None else None
            flow_t = None
            for tf in ("r6_pmc_traffic.json", "r5_pmc_traffic.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", tf)) as f:
                        flow_t = json.load(f)
                    flow_t["file"] = tf
                    break
                except Exception:
                    continue
            if flow_t is not None and flow_t.get("kernel", "").startswith("k_flow"):
                out["roofline"]["traffic"] = flow_t["bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/%s: %s" % (flow_t["file"], flow_t["source"])
                out["roofline"]["traffic_fetch"] = flow_t["fetch_bytes_per_launch"]
                out["roofline"]["traffic_write"] = flow_t["write_bytes_per_launch"]
                out["roofline"]["traffic_over_algorithmic"] = flow_t["bytes_per_launch"] / (8.0 * N * N * (N / 512.0))
            else:
                out["roofline"]["traffic"] = None
                out["roofline"]["traffic_note"] = "null: no counter pass of the dataflow kernel (profiles/r5_pmc_traffic.json) in this tree"
            if stream is not None:
                out["roofline"]["traffic_stream_schedule"] = stream
        if rccl is not None:
            out["config"]["rccl_ranks"] = rccl[0]
        if kind == "exact" and not sharded_mode and acc["nprof"] > 0:
            out["roofline"].update({
                "algorithmic_bytes_per_launch": 8.0 * N * N * (N / 512.0) / max(gemm_launches / nprof, 1.0),
                # SURVEY 8d's own figure for one evaluation (24 N^2: Gram written, G read, L read twice) beside the blocked algorithm's minimum C traffic above
                "algorithmic_bytes_per_eval": 24.0 * N * N,
                "traffic_over_algorithmic_bytes_per_eval": (out["roofline"]["traffic"] / (24.0 * N * N)) if out["roofline"].get("traffic") else None,
                # the same algorithmic flops over the DOMINANT KERNEL's own duration (HIP events around its launch in the profiled step): what a
                # kernel trace's average duration gives -- `frac` above prices them over the whole step (Gram, moments, host share included)
                "kernel_frac": (algo_flops / (gemm_s / nprof) / 1e12 / FP64_MFMA_PEAK_TFLOPS) if (gemm_s > 0 and gemm_launches / nprof <= 2.0) else None,
                "span": gemm_flops / span_s / 1e12 if span_s > 0 else None,
                "per_launch": gemm_flops / gemm_s / 1e12 if gemm_s > 0 else None,
                "overlap": gemm_s / span_s if span_s > 0 else None,
                "launches_per_eval": gemm_launches / nprof, "profiled_steps": nprof,
                "avg_launch_us": 1e6 * gemm_s / max(gemm_launches, 1), "flops_per_eval_issued": gemm_flops / nprof})
            out["stages_ms_per_eval"] = {k: float(stage[i] / nprof) for k, i in
                                         (("gram", _lib.ST_GRAM), ("potrf", _lib.ST_POTRF), ("trtri", _lib.ST_TRTRI),
                                          ("solve", _lib.ST_SOLVE), ("lauum", _lib.ST_LAUUM), ("moments", _lib.ST_MOMENTS),
                                          ("device_total", _lib.ST_TOTAL), ("gemm_kernel", _lib.ST_GEMM_KERNEL),
                                          ("gram_kernel", _lib.ST_GRAM_KERNEL), ("moment_kernel", _lib.ST_MOMENT_KERNEL))}
            # secondary rooflines: the two HBM-bound passes, priced on the tile kernel alone (the stage also holds the phase-table pre-pass)
            out["gram_hbm"] = {"bound": "hbm", "achieved": gram_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gram_gbs / HBM_PEAK_GBS if gram_gbs else None,
                               "bytes_per_launch": gram_bytes, "launch_us": 1e3 * gram_alone_ms if gram_alone_ms else None,
                               "basis": "the tile kernel alone, one launch (median of 10 LML-only evaluations, HIP events around the launch)" if gram_alone_ms else "inside the gradient evaluation's schedule",
                               "in_schedule": {"achieved": gram_sched_gbs, "us_per_eval": 1e3 * stage[_lib.ST_GRAM_KERNEL] / nprof,
                                               "note": "two launches per gradient evaluation, the larger one on the bulk stream underneath the first chain kernel, next to the dataflow kernel's waiting workgroups"}}
            out["moments_hbm"] = {"bound": "hbm", "achieved": mom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": mom_gbs / HBM_PEAK_GBS if mom_gbs else None,
                                  "bytes_per_launch": gram_bytes}
        if sustained is not None:
            out["sustained"] = sustained
        if sharded is not None:
            out["sharded"] = sharded
            out["sharded_cfg3"] = sharded_headline(sharded, world)

    # ---- the other configurations on this GPU, then the CPU baseline: outside the timed region, rank 0 at N = 1 only -----------------------
    if rank == 0 and world == 1 and not sharded_mode and a.config == "cfg2" and not a.n:
        if not a.no_configs:
            h.close()                   # the headline model's 1.5 GB of workspaces are not needed any more
            out["configs"] = extra_configs(local_rank)
        if not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, C, Q)
                out["cpu_baseline"].update(cpu_identity())
            except Exception as e:      # the baseline is a report, never a reason to lose the GPU measurement
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
    elif rank == 0 and world == 1 and not a.no_cpu_baseline and not sharded_mode:
        out["cpu_baseline"] = {"value": None, "note": "cpu_baseline is timed on the headline configuration (cfg2) only; reference CPU timings of the "
                                                      "other configurations are in BASELINE.md / DESIGN.md section 5"}

    try:
        mogptk_amd.shutdown_distributed()
    except Exception:
        pass
    if pd is not None:
        try:
            pd.destroy_process_group()
        except Exception:
            pass
    if rank == 0:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)      # RCCL's start-up banner sits in the C stdio buffer: get it out BEFORE the result line
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
