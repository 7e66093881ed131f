"""
Run under torch.distributed.run: every rank evaluates the same exact-GP LML+gradient and prediction once alone and once sharded over
the ranks (mogp_exact_eval_sharded / mogp_exact_predict_sharded: owned Gram / moment tiles, collectives issued inside the library), and
rank 0 prints the differences as one JSON line.
  backend gloo  -> all ranks may share one GPU; the library's collectives go through host-staging callbacks (external communicator):
                   the validation mode of tests/test_gpu_parity.py
  backend nccl  -> one GPU per rank, the library's own RCCL communicator on its own streams: the production mode
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/shard_check.py [--points 3000] [--backend gloo] [--reps 3]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=3000)
ap.add_argument("--channels", type=int, default=4)
ap.add_argument("--q", type=int, default=3)
ap.add_argument("--backend", default="gloo")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--titsias-points", type=int, default=6000)
ap.add_argument("--inducing", type=int, default=96, help="inducing points per channel of the Titsias check")
ap.add_argument("--exact-only", action="store_true", help="the exact model's evaluation and prediction only (the exchange-variant tests)")
ap.add_argument("--protocol", action="store_true", help="the sharded evaluation through the stage-by-stage entry points (mogp_shard_begin / pack / unpack / block / alpha / finish), "
                                                        "the collectives issued by the caller (mogptk_amd.dist.sharded_eval); implies --exact-only")
a = ap.parse_args()
if a.protocol:
    a.exact_only = True

import torch
import torch.distributed as dist
import mogptk_amd
from mogptk_amd import gpr, synth

rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
dev = local if a.backend == "nccl" else 0
if a.backend == "nccl":
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
else:
    dist.init_process_group("gloo")
gpr.config.device = dev

X, y = synth.make_data(a.points, a.channels)
h = synth.mosm_hypers(a.channels, a.q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=a.q, output_dims=a.channels)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
m.likelihood.scale.assign(h["scale"])

l0 = float(m.loss())
g0 = [p.grad.copy() for p in m.parameters()]
t = time.perf_counter()
for _ in range(a.reps):
    m.loss()
t_single = (time.perf_counter() - t) / a.reps

_use = mogptk_amd.use_protocol if a.protocol else mogptk_amd.use_distributed
comm = _use()
comm.force = True
rccl_ranks = None
if a.backend == "nccl":                  # how many ranks a collective issued by the LIBRARY on its own communicator sums over (mogp_comm_selftest)
    from mogptk_amd import _lib
    rccl_ranks = int(_lib.comm_selftest(dev)[0])
l1 = float(m.loss())
g1 = [p.grad.copy() for p in m.parameters()]
dist.barrier()
t = time.perf_counter()
for _ in range(a.reps):
    m.loss()
dist.barrier()
t_shard = (time.perf_counter() - t) / a.reps

# prediction: the inversion sharded the same way, every rank its tile rows' share of the quadratic form K_s. Kj^-1 K_.s, one all-reduce of S doubles
Xs = synth.test_inputs(max(40, a.points // 10), a.channels)
mu1, var1 = m.predict_f(Xs)
mogptk_amd.use_single_device()
mu0, var0 = m.predict_f(Xs)
perr = max(float(np.max(np.abs(mu1 - mu0)) / np.max(np.abs(mu0))), float(np.max(np.abs(var1 - var0)) / np.max(np.abs(var0))))

err = max(float(np.max(np.abs(b - c)) / np.max(np.abs(c))) for b, c in zip(g1, g0))

# owned-rows allocation: a model whose FIRST evaluation is the sharded one holds physical memory only under its own tile rows of the work matrix
# (mogp_model_work_bytes); the sharded prediction on it adds the second matrix; the first one-GPU call afterwards makes it whole -- and must give the
# one-GPU gradient again (the row ownership of the sharded call no longer applies)
def _fresh():
    kk = gpr.MultiOutputSpectralMixtureKernel(Q=a.q, output_dims=a.channels)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(kk, name).assign(h[name])
    mm = gpr.Exact(kk, X, y, variance=h["scale"] ** 2)
    mm.likelihood.scale.assign(h["scale"])
    return mm
_use().force = True
m2 = _fresh()
l2 = float(m2.loss())
g2 = [p.grad.copy() for p in m2.parameters()]
backed, whole = m2._handle.work_bytes()
mu2, var2 = m2.predict_f(Xs)
backed_after_predict = m2._handle.work_bytes()[0] if not a.protocol else backed      # (the sharded prediction works from the owned rows too)
mogptk_amd.use_single_device()
l3 = float(m2.loss())
g3 = [p.grad.copy() for p in m2.parameters()]
backed_after, _ = m2._handle.work_bytes()
owned = dict(rel_loss=abs(l2 - l0) / abs(l0), rel_grad=max(float(np.max(np.abs(b - c)) / np.max(np.abs(c))) for b, c in zip(g2, g0)),
             rel_predict=max(float(np.max(np.abs(mu2 - mu0)) / np.max(np.abs(mu0))), float(np.max(np.abs(var2 - var0)) / np.max(np.abs(var0)))),
             backed_bytes=backed, whole_bytes=whole, backed_after_sharded_predict=backed_after_predict, backed_after_one_gpu_call=backed_after,
             rel_loss_one_gpu_after=abs(l3 - l0) / abs(l0), rel_grad_one_gpu_after=max(float(np.max(np.abs(b - c)) / np.max(np.abs(c))) for b, c in zip(g3, g0)))
del m2

if a.exact_only:
    errs = torch.tensor([abs(l1 - l0) / abs(l0), err, perr], dtype=torch.float64)
    if a.backend == "nccl":
        errs = errs.cuda()
    dist.all_reduce(errs, op=dist.ReduceOp.MAX)
    owned_all = [None] * world
    dist.all_gather_object(owned_all, owned)
    owned = {k: max(o[k] for o in owned_all) for k in owned}          # the worst rank of each
    if rank == 0:
        print(json.dumps(dict(owned_rows=owned, world=world, backend=a.backend, N=a.points, loss=l0, loss_sharded=l1, rel_loss=float(errs[0]), rel_grad=float(errs[1]),
                              rel_predict=float(errs[2]), transport=getattr(comm, "transport", "caller (stage protocol)"), rccl_ranks=rccl_ranks, ms_single=1e3 * t_single, ms_sharded=1e3 * t_shard,
                              split=os.environ.get("MOGP_SHARD_SPLIT", "1"), factor_once=os.environ.get("MOGP_SHARD_FACTOR_ONCE", "0"))))
    mogptk_amd.shutdown_distributed()
    dist.destroy_process_group()
    sys.exit(0)

# the sparse (Titsias) bound DATA-PARALLEL: every rank holds every world-th training point, the sums over points are all-reduced inside the
# library (mogp_titsias_eval_sharded / _predict_sharded); bound, every gradient and the prediction against the one-GPU evaluation
Xt, yt = synth.make_data(a.titsias_points, a.channels)
kt = gpr.MultiOutputSpectralMixtureKernel(Q=a.q, output_dims=a.channels)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(kt, name).assign(h[name])
mt = gpr.Titsias(kt, Xt, yt, Z=a.inducing, variance=float(np.mean(h["scale"])) ** 2, jitter=1e-6)
tl0 = float(mt.loss())
tg0 = [p.grad.copy() for p in mt.parameters()]
tmu0, tvar0 = mt.predict_f(Xs)
t = time.perf_counter()
for _ in range(a.reps):
    mt.loss()
tt_single = (time.perf_counter() - t) / a.reps
comm = mogptk_amd.use_distributed()
comm.force = True
tl1 = float(mt.loss())
tg1 = [p.grad.copy() for p in mt.parameters()]
dist.barrier()
t = time.perf_counter()
for _ in range(a.reps):
    mt.loss()
dist.barrier()
tt_shard = (time.perf_counter() - t) / a.reps
tmu1, tvar1 = mt.predict_f(Xs)
mogptk_amd.use_single_device()
terr = max(float(np.max(np.abs(b - c)) / max(1e-300, np.max(np.abs(c)))) for b, c in zip(tg1, tg0))
tperr = max(float(np.max(np.abs(tmu1 - tmu0)) / np.max(np.abs(tmu0))), float(np.max(np.abs(tvar1 - tvar0)) / np.max(np.abs(tvar0))))

# the variational sparse model (SparseHensman, Student-t likelihood) data-parallel the same way: mogp_svgp_backward_sharded + the
# likelihood's expectation and parameter gradients all-reduced on the host
kh = gpr.MultiOutputSpectralMixtureKernel(Q=a.q, output_dims=a.channels)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(kh, name).assign(h[name])
mh = gpr.SparseHensman(kh, Xt, yt, Z=a.inducing, likelihood=gpr.StudentTLikelihood(dof=4, scale=0.3), jitter=1e-6)
rng = np.random.default_rng(7)
Mh = mh.q_mu().shape[0]
mh.q_mu.assign(rng.normal(0, 0.3, (Mh, 1)))
mh.q_sqrt.assign(np.tril(rng.normal(0, 0.02, (Mh, Mh))) + np.diag(rng.uniform(0.5, 1.0, Mh)))
hl0 = float(mh.loss())
hg0 = [p.grad.copy() for p in mh.parameters()]
t = time.perf_counter()
for _ in range(a.reps):
    mh.loss()
ht_single = (time.perf_counter() - t) / a.reps
comm = mogptk_amd.use_distributed()
comm.force = True
hl1 = float(mh.loss())
hg1 = [p.grad.copy() for p in mh.parameters()]
dist.barrier()
t = time.perf_counter()
for _ in range(a.reps):
    mh.loss()
dist.barrier()
ht_shard = (time.perf_counter() - t) / a.reps
mogptk_amd.use_single_device()
herr = max(float(np.max(np.abs(b - c)) / max(1e-300, np.max(np.abs(c)))) for b, c in zip(hg1, hg0))

# the FITC (Snelson) model, data-parallel the same way (mogp_snelson_eval_sharded / _predict_sharded)
kn = gpr.MultiOutputSpectralMixtureKernel(Q=a.q, output_dims=a.channels)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(kn, name).assign(h[name])
mn = gpr.Snelson(kn, Xt, yt, Z=a.inducing, variance=list(np.asarray(h["scale"]) ** 2), jitter=1e-6)
nl0 = float(mn.loss())
ng0 = [p.grad.copy() for p in mn.parameters()]
nmu0, nvar0 = mn.predict_f(Xs)
comm = mogptk_amd.use_distributed()
comm.force = True
nl1 = float(mn.loss())
ng1 = [p.grad.copy() for p in mn.parameters()]
nmu1, nvar1 = mn.predict_f(Xs)
mogptk_amd.use_single_device()
nerr = max(float(np.max(np.abs(b - c)) / max(1e-300, np.max(np.abs(c)))) for b, c in zip(ng1, ng0))
nperr = max(float(np.max(np.abs(nmu1 - nmu0)) / np.max(np.abs(nmu0))), float(np.max(np.abs(nvar1 - nvar0)) / np.max(np.abs(nvar0))))

errs = torch.tensor([abs(l1 - l0) / abs(l0), err, perr, abs(tl1 - tl0) / abs(tl0), terr, tperr, abs(hl1 - hl0) / abs(hl0), herr,
                     abs(nl1 - nl0) / abs(nl0), nerr, nperr], dtype=torch.float64)
if a.backend == "nccl":
    errs = errs.cuda()
dist.all_reduce(errs, op=dist.ReduceOp.MAX)
owned_all = [None] * world
dist.all_gather_object(owned_all, owned)
owned = {k: max(o[k] for o in owned_all) for k in owned}          # the worst rank of each
per_rank = [None] * world          # every rank's own values: a rank whose ONE-GPU evaluation went wrong shows here, not in the sharded ones
dist.all_gather_object(per_rank, [l0, l1, tl0, tl1, hl0, hl1, nl0, nl1])
if rank == 0:
    print(json.dumps(dict(owned_rows=owned, world=world, backend=a.backend, N=a.points, loss=l0, loss_sharded=l1, rel_loss=float(errs[0]), rel_grad=float(errs[1]), rel_predict=float(errs[2]),
                          transport=comm.transport, rccl_ranks=rccl_ranks, ms_single=1e3 * t_single, ms_sharded=1e3 * t_shard,
                          titsias=dict(N=a.titsias_points, M=int(mt.Z().shape[0]), loss=tl0, rel_loss=float(errs[3]), rel_grad=float(errs[4]),
                                       rel_predict=float(errs[5]), ms_single=1e3 * tt_single, ms_sharded=1e3 * tt_shard),
                          hensman=dict(N=a.titsias_points, M=int(Mh), likelihood="StudentT", loss=hl0, rel_loss=float(errs[6]), rel_grad=float(errs[7]),
                                       ms_single=1e3 * ht_single, ms_sharded=1e3 * ht_shard),
                          snelson=dict(loss=nl0, rel_loss=float(errs[8]), rel_grad=float(errs[9]), rel_predict=float(errs[10])),
                          per_rank=per_rank if any(v != per_rank[0] for v in per_rank) else "identical on every rank")))
mogptk_amd.shutdown_distributed()
dist.destroy_process_group()
