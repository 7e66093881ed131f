#!/usr/bin/env python
"""
bench.py -- BASELINE.json's headline metric on MI355X:
    log-marginal-likelihood + gradient evaluations per second, MOSM C=4 Q=3 N=8192 (configs[1]), exact GP, fp64.

A "step" is one `gpr.Exact.loss()`-equivalent: term table upload, Gram build (+noise +jitter), Cholesky, triangular
inverse, alpha / log-det, K^-1, gradient-moment pass, moments back to the host, host chain rule to the raw-parameter
gradient.  X and y are resident in HBM before the timed region (model creation); only the O(C^2 Q) parameter table
goes host->device per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5] [--mode replicas|sharded]

--config picks the BASELINE.json configuration a step runs: cfg2 (default, the metric's own), cfg3 = the same evaluation
at MOSM C=8 Q=5 N=32768, cfg4 = one `predict_f` of 4096 test points on CSM C=4 Q=3 N=16384, cfg5 = one Titsias ELBO+gradient
evaluation at N=100000, M=2048.  Every config reports the same JSON line; `roofline` prices the algorithmic flops of one step
 over `ms_per_step`.

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU).  Two modes:
  replicas (default)  `value` = aggregate evals/s of N independent replicas of the workload: one 15 ms evaluation does not pay for an
                      exchange per pivot block, N GPUs are best used as N evaluations (restarts, models); scaling "weak".
  sharded             `value` = evals/s of ONE evaluation spread over all ranks (mogp_exact_eval_sharded: owned Gram / moment tiles,
                      one RCCL all-gather per 512-wide pivot block issued by the library on its own streams); scaling "strong".
                      This is the mode configs[2] (`--config cfg3 --mode sharded`) is meant for.
In replicas mode at N > 1 the sharded numbers still ride along in the extra `sharded` object (this workload and cfg3, each next to its
one-GPU time), measured outside the timed region under a watchdog so that a stuck collective cannot cost the line.

Prints ONE JSON line (rank 0).  `roofline.frac` = algorithmic flops of a step / ms_per_step / fp64-MFMA peak; `span` / `per_launch`
price the dominant kernel (k_gemm) from HIP events around each of its launches inside the timed region; `cpu_baseline` times the
torch-CPU port of the reference's op sequence (oracle/torch_port.py) on this box's host cores.
"""
import argparse
import json
import os
import sys
import threading
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
    # forward: v = L^-1 Kuf (M^2 N) + Q = v v^T (M^2 N); backward: (I - Pq) v (2 M^2 N) + L^-T (.) (M^2 N); the M^3 terms are 1 %
    return m, (lambda: m.loss()), 5.0 * float(M) ** 2 * N + 4.0 * float(M) ** 3


def cpu_baseline(N, C, Q, budget_s=45.0):
    """torch-CPU port of the reference op sequence, one evaluation of the same workload when it fits the time
    budget, otherwise the largest power-of-two N that does (stated in `sample`)."""
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

    n = 2048
    t = one(n)
    while n < N and t * 8.5 < budget_s:       # ~cubic growth per doubling
        n *= 2
        t = one(n)
    if n == N:
        return dict(value=1.0 / t, unit="evals/s", cores=cores, kind="port",
                    sample="1 LML+grad eval of the same workload (MOSM C=%d Q=%d N=%d), torch-CPU fp64 port of the "
                           "reference op sequence, %.1f s" % (C, Q, N, t))
    scale = (N / n) ** 3
    return dict(value=1.0 / (t * scale), unit="evals/s", cores=cores, kind="port",
                sample="1 eval at N=%d took %.1f s; extrapolated to N=%d by N^3 (x%.0f)" % (n, t, N, scale))


def timed_region(step, steps, warmup, dist=None, sync=lambda: None, device="cpu"):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both sides; returns the MAX
    over ranks of the elapsed wall time (seconds).  `dist` is torch.distributed (initialised) or None."""
    import torch
    for _ in range(warmup):
        step(-1)
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
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


def sharded_probe(m, dist, sync, world, cfg3=True, reps=5, out=None):
    """Extra, outside the timed region: the SAME evaluation sharded over all ranks (mogp_exact_eval_sharded) next
    to the one-GPU evaluation, at the bench workload and at configs[2] (MOSM C=8 Q=5 N=32768); and configs[4] data-parallel.  Reported as
    `sharded`; `out` is filled entry by entry, so that a watchdog firing in a later entry still has the earlier ones."""
    import mogptk_amd

    def run(model, reps):
        mogptk_amd.use_single_device()
        l0 = float(model.loss()); g0 = [p.grad.copy() for p in model.parameters()]
        sync(); t = time.perf_counter()
        for _ in range(reps):
            model.loss()
        sync(); t_single = (time.perf_counter() - t) / reps
        comm = mogptk_amd.use_distributed()
        comm.force = True
        l1 = float(model.loss()); g1 = [p.grad.copy() for p in model.parameters()]
        dist.barrier(); sync(); t = time.perf_counter()
        for _ in range(reps):
            model.loss()
        sync(); dist.barrier(); t_shard = (time.perf_counter() - t) / reps
        mogptk_amd.use_single_device()
        err = max(float(np.max(np.abs(b - c)) / np.max(np.abs(c))) for b, c in zip(g1, g0))
        return dict(ms_one_gpu=1e3 * t_single, ms_sharded=1e3 * t_shard, speedup=t_single / t_shard, evals_per_s_sharded=1.0 / t_shard,
                    rel_loss=abs(l1 - l0) / abs(l0), rel_grad=err, transport=comm.transport)

    if out is None:
        out = {}
    out["ranks"] = world
    out["bench_workload"] = run(m, reps)
    if cfg3:
        # the sparse bound (configs[4]) DATA-PARALLEL: every rank holds every world-th training point, the sums over points are all-reduced
        # inside the library (mogp_titsias_eval_sharded: M^2 + M + 3 doubles, then the (Z, X) moments); strong scaling at N = 100 000 and
        # weak scaling at 100 000 points per rank, each next to the one-GPU evaluation of the same model
        for tag, n5 in (("cfg5_titsias_n100000_m2048", 100000), ("cfg5_weak_%d_points_per_rank" % 100000, 100000 * world)):
            if tag.startswith("cfg5_weak") and world == 1:
                continue
            try:
                m5, _, _ = build_model("cfg5", None, n_override=n5)
                out[tag] = run(m5, 3)
                out[tag]["N"] = n5
                del m5
            except Exception as e:
                out[tag] = {"error": repr(e)}
        try:                                           # last: the largest exchange (one 134 MB all-gather per pivot block)
            m3 = build_mosm(32768, 8, 5, None)         # gpr.config.device is already this rank's GPU
            out["cfg3_mosm_c8_q5_n32768"] = run(m3, 2)
        except Exception as e:
            out["cfg3_mosm_c8_q5_n32768"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="replicas", choices=["replicas", "sharded"])
    ap.add_argument("--n", type=int, default=None, help="override the configuration's N (development)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard-probe", action="store_true", help="also run the sharded-evaluation probe at --gpus 1 (1-rank RCCL group)")
    ap.add_argument("--no-shard-probe", action="store_true")
    ap.add_argument("--no-cfg3-probe", action="store_true")
    ap.add_argument("--probe-timeout", type=float, default=420.0, help="watchdog of the sharded probe in seconds")
    a = ap.parse_args()
    kind, C, Q, N, extra, desc = CONFIGS[a.config]
    if a.n:
        N = a.n
    big = a.config != "cfg2"
    steps = a.steps if a.steps is not None else (5 if big else 20)
    warmup = a.warmup if a.warmup is not None else (1 if big else 3)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    sharded_mode = a.mode == "sharded"       # exact / predict: one evaluation's tiles over the ranks; titsias: its data points over the ranks

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    pd = dist
    if sharded_mode or ((world > 1 or a.shard_probe) and not a.no_shard_probe and kind == "exact"):
        if dist is None:                    # one GPU: a 1-rank RCCL group exercises the same code path
            import torch.distributed as dist1
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
            dist1.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            pd = dist1

    run_step()                    # creates the device handle (X, y resident in HBM) before anything is timed
    h = m._handle
    if sharded_mode:
        comm = mogptk_amd.use_distributed()
        comm.force = True
        run_step()
    stage = np.zeros(_lib.ST_COUNT)
    acc = dict(flops=0.0, launches=0, nprof=0)
    PROFILE_EVERY = max(20, steps)          # HIP events around every GEMM launch cost ~5 % of a step: ONE timed step (the first) carries them

    def step(i):
        prof = i >= 0 and (i % PROFILE_EVERY) == 0 and kind == "exact" and not sharded_mode
        h.set_profiling(prof)
        run_step()
        if prof:
            ms, nl, fl = h.stage_ms()
            stage[:] += ms
            acc["flops"] += fl
            acc["launches"] += nl
            acc["nprof"] += 1

    dt = timed_region(step, steps, warmup, dist, sync, "cuda" if dist is not None else "cpu")
    gemm_flops, gemm_launches, nprof = acc["flops"], acc["launches"], max(acc["nprof"], 1)
    h.set_profiling(False)
    if sharded_mode:
        mogptk_amd.use_single_device()

    # ---- the line without the probe: what the watchdog prints if the probe below does not come back -------------------------------
    out = None
    if rank == 0:
        ms_per_step = 1e3 * dt / steps
        value = aggregate_value(world, steps, dt, sharded_mode)
        # The GEMM launches of one evaluation run on up to four streams at once (potri.hip), so the sum of their durations exceeds the
        # wall-clock time they occupy: `span` prices them over the factorisation + inversion stage, `per_launch` over the sum of their
        # own durations (what a kernel trace averages to); `frac` -- the headline -- prices the ALGORITHMIC flops over the whole step.
        gemm_s = stage[_lib.ST_GEMM_KERNEL] * 1e-3
        span_s = (stage[_lib.ST_POTRF] + stage[_lib.ST_TRTRI] + stage[_lib.ST_SOLVE] + stage[_lib.ST_LAUUM]) * 1e-3
        achieved = algo_flops / (ms_per_step * 1e-3) / 1e12
        if sharded_mode:
            achieved /= world            # per GPU
        gram_bytes = 4.0 * N * (N + 1)            # lower triangle written / read once
        # the two HBM-bound passes: algorithmic bytes over the duration of the tile kernel alone (HIP events around that one launch)
        gram_gbs = gram_bytes * nprof / (stage[_lib.ST_GRAM_KERNEL] * 1e-3) / 1e9 if stage[_lib.ST_GRAM_KERNEL] > 0 else None
        mom_gbs = gram_bytes * nprof / (stage[_lib.ST_MOMENT_KERNEL] * 1e-3) / 1e9 if stage[_lib.ST_MOMENT_KERNEL] > 0 else None
        traffic, traffic_src = None, None
        try:        # L2<->fabric bytes per k_gemm launch from the committed rocprofv3 --pmc passes of this same command (profiles/)
            with open(os.path.join(ROOT, "profiles", "r2_pmc_traffic.json")) as f:
                t = json.load(f)
            if a.config == "cfg2" and not sharded_mode:
                traffic, traffic_src = t["bytes_per_launch"], "profiles/r2_pmc_traffic.json: " + t["source"]
        except Exception:
            pass
        if kind == "exact":
            metric = METRICS[kind][0] % (C, N)
        elif kind == "predict":
            metric = METRICS[kind][0] % (extra, C, N)
        else:
            metric = METRICS[kind][0] % (C, N, extra)
        if a.config == "cfg2":
            metric = "log-marginal-likelihood+grad evals/sec, MOSM C=4 N=8192; 1/2/4/8 GPU"      # BASELINE.json's wording
        par = "1 gpu" if world == 1 and not sharded_mode else (("data-parallel x%d (points cyclic over the ranks, RCCL all-reduce of the M x M sums)" % world if kind == "titsias" else
                                                                       "sharded x%d (tile rows cyclic, RCCL all-gather per pivot block)" % world) if sharded_mode else "replicas x%d" % world)
        out = {
            "metric": metric, "value": value, "unit": METRICS[kind][1], "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if sharded_mode else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc if not a.n else desc + " [N overridden: %d]" % N, "channels": C, "Q": Q, "N": N, "parallelism": par,
                       "device": _lib.device_name(local_rank)},
            "roofline": {"bound": "mfma", "kernel": "k_gemm (fp64 v_mfma_f64_16x16x4_f64)", "achieved": achieved,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "basis": "algorithmic flops of one step (SURVEY.md 8d: %.3e) / ms_per_step%s" % (algo_flops, " / ranks" if sharded_mode else ""),
                         "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src},
        }
        if kind == "exact" and not sharded_mode and acc["nprof"] > 0:
            out["roofline"].update({
                "algorithmic_bytes_per_launch": 8.0 * N * N * (N / 512.0) / max(gemm_launches / nprof, 1.0),
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
                               "bytes_per_launch": gram_bytes}
            out["moments_hbm"] = {"bound": "hbm", "achieved": mom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": mom_gbs / HBM_PEAK_GBS if mom_gbs else None,
                                  "bytes_per_launch": gram_bytes}

    def emit():
        if rank == 0:
            import ctypes
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)      # RCCL's start-up banner sits in the C stdio buffer: get it out BEFORE the result line
            print(json.dumps(out), flush=True)

    # the CPU baseline runs BEFORE any communicator exists (RCCL proxy threads would compete with it for the host cores)
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not sharded_mode:
        try:
            if a.config == "cfg2":
                out["cpu_baseline"] = cpu_baseline(N, C, Q)
            else:
                out["cpu_baseline"] = {"value": None, "note": "cpu_baseline is timed on the headline configuration (cfg2) only; reference CPU timings of the "
                                                              "other configurations are in BASELINE.md / DESIGN.md section 5"}
        except Exception as e:      # the baseline is a report, never a reason to lose the GPU measurement
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    # ---- extras that may not come back: the sharded probe (collectives on hardware this code has never seen) under a watchdog --------
    want_probe = (world > 1 or a.shard_probe) and not a.no_shard_probe and kind == "exact" and not sharded_mode
    if want_probe:
        done = threading.Event()
        sharded = {}

        def watchdog():
            if not done.wait(a.probe_timeout):
                if out is not None:                 # whatever entries were finished, plus the reason the rest is missing
                    out["sharded"] = dict(sharded, error="the sharded probe did not finish within %.0f s (watchdog)" % a.probe_timeout)
                emit()
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            sharded_probe(m, pd, sync, world, cfg3=not a.no_cfg3_probe, out=sharded)
        except Exception as e:              # symmetric across ranks (same code, same inputs); the measurement above stands
            sharded["error"] = repr(e)
        done.set()
        if out is not None:
            out["sharded"] = sharded

    try:
        mogptk_amd.shutdown_distributed()
    except Exception:
        pass
    if pd is not None:
        try:
            pd.destroy_process_group()
        except Exception:
            pass
    emit()


if __name__ == "__main__":
    main()
