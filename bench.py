#!/usr/bin/env python
"""
bench.py -- BASELINE.json's headline metric on MI355X:
    log-marginal-likelihood + gradient evaluations per second, MOSM C=4 Q=3 N=8192 (configs[1]), exact GP, fp64.

A "step" is one `gpr.Exact.loss()`-equivalent: term table upload, Gram build (+noise +jitter), Cholesky, triangular
inverse, alpha / log-det, K^-1, gradient-moment pass, moments back to the host, host chain rule to the raw-parameter
gradient.  X and y are resident in HBM before the timed region (model creation); only the O(C^2 Q) parameter table
goes host->device per step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL).  `value` at N > 1 is the aggregate
evals/s of N independent replicas of the workload (one 16 ms evaluation does not pay for an exchange per pivot block -- DESIGN.md
section 6), so scaling is "weak".  The sharded evaluation (one evaluation spread over all ranks, RCCL all-gather per pivot
block) is measured next to it, outside the timed region, and reported in the extra `sharded` object for the bench workload and
for configs[2] (N=32768), where it is the point.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (the fp64 MFMA GEMM, k_gemm) from HIP events
recorded around every one of its launches inside the timed region; `cpu_baseline` times the torch-CPU port of the
reference's op sequence (oracle/torch_port.py) on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X vendor figure for FP64 matrix (v_mfma_f64_16x16x4_f64); see DESIGN.md section 5
HBM_PEAK_GBS = 8000.0


def build_model(N, C, Q, device):
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
    return m, X, y


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


def sharded_probe(m, dist, sync, world, cfg3=True):
    """Extra, outside the timed region: the SAME evaluation sharded over all ranks (mogp_shard_* + one RCCL all-gather per 512-wide
    pivot block) next to the one-GPU evaluation, at the bench workload and at configs[2] (MOSM C=8 Q=5 N=32768).
    Reported as `sharded`; never part of `value`."""
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
        return dict(ms_one_gpu=1e3 * t_single, ms_sharded=1e3 * t_shard, speedup=t_single / t_shard,
                    rel_loss=abs(l1 - l0) / abs(l0), rel_grad=err)

    out = {"ranks": world, "bench_workload": run(m, 5)}
    if cfg3:
        try:                                           # kept apart: a failure here must not cost the numbers above
            m3, _, _ = build_model(32768, 8, 5, None)  # gpr.config.device is already this rank's GPU
            out["cfg3_mosm_c8_q5_n32768"] = run(m3, 2)
        except Exception as e:
            out["cfg3_mosm_c8_q5_n32768"] = {"error": repr(e)}
    return out


def aggregate_value(world, steps, dt):
    """whole-job evals/s: every rank ran `steps` evaluations of its own replica in `dt` (max over ranks)"""
    return world * steps / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--q", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard-probe", action="store_true", help="also run the sharded-evaluation probe at --gpus 1 (1-rank RCCL group)")
    ap.add_argument("--no-shard-probe", action="store_true")
    ap.add_argument("--no-cfg3-probe", action="store_true")
    a = ap.parse_args()

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

    from mogptk_amd import _lib
    m, X, y = build_model(a.n, a.channels, a.q, local_rank)

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    m.loss()                      # creates the device handle (X, y resident in HBM) before anything is timed
    h = m._handle
    stage = np.zeros(_lib.ST_COUNT)
    acc = dict(flops=0.0, launches=0, nprof=0)
    PROFILE_EVERY = 10          # HIP events around every GEMM launch cost ~5 % of a step: sample one step in ten, inside the timed region

    def step(i):
        prof = i >= 0 and (i % PROFILE_EVERY) == 0
        h.set_profiling(prof)
        m.loss()
        if prof:
            ms, nl, fl = h.stage_ms()
            stage[:] += ms
            acc["flops"] += fl
            acc["launches"] += nl
            acc["nprof"] += 1

    dt = timed_region(step, a.steps, a.warmup, dist, sync, "cuda" if dist is not None else "cpu")
    gemm_flops, gemm_launches, nprof = acc["flops"], acc["launches"], max(acc["nprof"], 1)
    h.set_profiling(False)

    sharded = None
    if (world > 1 or a.shard_probe) and not a.no_shard_probe:
        if dist is None:                    # --shard-probe on one GPU: a 1-rank RCCL group exercises the same code path
            import torch.distributed as dist1
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
            dist1.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            pd = dist1
        else:
            pd = dist
        try:
            sharded = sharded_probe(m, pd, sync, world, cfg3=not a.no_cfg3_probe)
        except Exception as e:              # symmetric across ranks (same code, same inputs); the replica measurement above stands
            sharded = {"error": repr(e)}
        if dist is None:
            pd.destroy_process_group()

    if rank == 0:
        ms_per_step = 1e3 * dt / a.steps
        value = aggregate_value(world, a.steps, dt)
        # The GEMM launches of one evaluation run on up to four streams at once (potri.hip), so the sum of their durations
        # exceeds the wall-clock time they occupy.  `achieved` prices the kernel over the SPAN of the factorisation + inversion
        # stage (HIP events on the critical stream); `per_launch` is flops / sum of launch durations (what a kernel trace
        # averages to), `overlap` = sum of durations / span.
        gemm_s = stage[_lib.ST_GEMM_KERNEL] * 1e-3
        # (the two mat-vecs for alpha overlap the last GEMM of the fused schedule, so their stage is part of the span)
        span_s = (stage[_lib.ST_POTRF] + stage[_lib.ST_TRTRI] + stage[_lib.ST_SOLVE] + stage[_lib.ST_LAUUM]) * 1e-3
        achieved = gemm_flops / span_s / 1e12 if span_s > 0 else 0.0
        per_launch = gemm_flops / gemm_s / 1e12 if gemm_s > 0 else 0.0
        N = a.n
        gram_bytes = 4.0 * N * (N + 1)            # lower triangle written once
        gram_gbs = gram_bytes * nprof / (stage[_lib.ST_GRAM] * 1e-3) / 1e9 if stage[_lib.ST_GRAM] > 0 else 0.0
        mom_gbs = gram_bytes * nprof / (stage[_lib.ST_MOMENTS] * 1e-3) / 1e9 if stage[_lib.ST_MOMENTS] > 0 else 0.0
        traffic, traffic_src = None, None
        try:        # L2<->fabric bytes per k_gemm launch from the committed rocprofv3 --pmc passes of this same command (profiles/)
            with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
                t = json.load(f)
            traffic, traffic_src = t["bytes_per_launch"], "profiles/r1_pmc_traffic.json: " + t["source"]
        except Exception:
            pass
        out = {
            "metric": "log-marginal-likelihood+grad evals/sec, MOSM C=4 N=8192; 1/2/4/8 GPU",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "MOSM C=%d Q=%d N=%d exact GP LML+gradient (BASELINE.json configs[1])" % (a.channels, a.q, N),
                       "channels": a.channels, "Q": a.q, "N": N, "parallelism": "replicas x%d" % world if world > 1 else "1 gpu",
                       "device": _lib.device_name(local_rank)},
            "roofline": {"bound": "mfma", "kernel": "k_gemm (fp64 v_mfma_f64_16x16x4_f64)", "achieved": achieved,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": 8.0 * N * N * (N / 512.0) / max(gemm_launches / nprof, 1.0), "basis": "GEMM flops of the profiled evaluations / wall-clock span of their factorisation+inversion stage",
                         "per_launch": per_launch, "overlap": gemm_s / span_s if span_s > 0 else None,
                         "launches_per_eval": gemm_launches / nprof, "profiled_steps": nprof, "avg_launch_us": 1e6 * gemm_s / max(gemm_launches, 1),
                         "flops_per_eval": gemm_flops / nprof},
            "stages_ms_per_eval": {k: float(stage[i] / nprof) for k, i in
                                   (("gram", _lib.ST_GRAM), ("potrf", _lib.ST_POTRF), ("trtri", _lib.ST_TRTRI),
                                    ("solve", _lib.ST_SOLVE), ("lauum", _lib.ST_LAUUM), ("moments", _lib.ST_MOMENTS),
                                    ("device_total", _lib.ST_TOTAL), ("gemm_kernel", _lib.ST_GEMM_KERNEL))},
            "gram_hbm": {"achieved": gram_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gram_gbs / HBM_PEAK_GBS,
                         "bytes_per_launch": gram_bytes},
            "moments_hbm": {"achieved": mom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": mom_gbs / HBM_PEAK_GBS},
        }
        if sharded is not None:
            out["sharded"] = sharded
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, a.channels, a.q)
            except Exception as e:      # the baseline is a report, never a reason to lose the GPU measurement
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)      # RCCL's start-up banner sits in the C stdio buffer: get it out BEFORE the result line
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
