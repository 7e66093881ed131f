"""
The N > 1 path on CPU: two gloo ranks run mogptk_amd.dist.sharded_eval (the production orchestration: all-gather of every
pivot block's panel, broadcasts of the pivot rows, all-reduce of alpha / moments / diag sums) over the numpy twin of the
sharded device stages (oracle/table_model.py), through the product's own gpr.Exact.loss(); the raw-parameter gradients must
equal the single-process ones and the reference's autograd golden.
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import sys, json
    sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
    import numpy as np
    import torch.distributed as dist
    import mogptk_amd, mogptk_amd._lib as L
    from mogptk_amd import gpr, synth
    from oracle.table_model import TableDevice
    from helpers import load, product_exact, fixture_params
    L.ExactHandle = TableDevice
    L.gram = None
    dist.init_process_group("gloo")
    # (1) golden fixture through the sharded path
    fx = load("lml_mosm_c3q2.npz")
    m, fp = product_exact(fx)
    ref_loss = float(m.loss()); ref_grads = [None if p.grad is None else p.grad.copy() for p in m.parameters()]
    # route grad evaluations through dist.sharded_eval: TableDevice has no comm hook of its own, so patch its eval
    comm = mogptk_amd.use_protocol()
    from mogptk_amd import dist as D
    single = TableDevice.eval
    TableDevice.eval = lambda self, noise, jitter, grad=True, data_var=None: (
        D.sharded_eval(self, comm, noise, jitter, data_var) if grad else single(self, noise, jitter, grad, data_var))
    m._handle = None
    loss = float(m.loss())
    err = max(float(np.max(np.abs(p.grad - g)) / max(1.0, np.max(np.abs(g)))) for p, g in zip(m.parameters(), ref_grads) if g is not None)
    gold = max(float(np.max(np.abs(p.grad - f["grad"])) / max(1.0, np.max(np.abs(f["grad"])))) for p, f in zip(m.parameters(), fp) if f["grad"] is not None)
    # (2) a problem with several pivot blocks (N = 1400 -> 11 tile rows, 3 blocks) and ragged channels, shuffled rows
    rng = np.random.default_rng(5)
    X, y = synth.make_data(1400, 4)
    perm = rng.permutation(1400); X, y = X[perm], y[perm]
    h = synth.mosm_hypers(4, 2)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=2, output_dims=4)
    for n in ("weight", "mean", "variance", "delay", "phase"): getattr(k, n).assign(h[n])
    m2 = gpr.Exact(k, X, y, variance=h["scale"] ** 2); m2.likelihood.scale.assign(h["scale"])
    TableDevice.eval = single
    l_single = float(m2.loss()); g_single = [p.grad.copy() for p in m2.parameters()]
    TableDevice.eval = lambda self, noise, jitter, grad=True, data_var=None: (
        D.sharded_eval(self, comm, noise, jitter, data_var) if grad else single(self, noise, jitter, grad, data_var))
    m2._handle = None
    l_shard = float(m2.loss())
    err2 = max(float(np.max(np.abs(p.grad - g)) / np.max(np.abs(g))) for p, g in zip(m2.parameters(), g_single))
    if dist.get_rank() == 0:
        print(json.dumps(dict(loss=loss, ref_loss=ref_loss, err=err, gold=gold, l_single=l_single, l_shard=l_shard, err2=err2,
                              world=dist.get_world_size())))
    dist.destroy_process_group()
''')


import pytest


@pytest.mark.parametrize("ranks", [2, 3])
def test_sharded_evaluation_gloo_ranks(tmp_path, ranks):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, tests=os.path.join(ROOT, "tests")))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks,
                          "--master-addr", "127.0.0.1", "--master-port", str(29615 + ranks), str(script)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == ranks
    assert abs(r["loss"] - r["ref_loss"]) < 1e-10 * abs(r["ref_loss"]) and r["err"] < 1e-9      # sharded == single process
    assert r["gold"] < 1e-8                                                                      # == reference autograd
    assert abs(r["l_shard"] - r["l_single"]) < 1e-10 * abs(r["l_single"]) and r["err2"] < 1e-8


TITSIAS_WORKER = textwrap.dedent('''
    import sys, json
    sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
    import numpy as np
    import torch.distributed as dist
    import mogptk_amd, mogptk_amd._lib as L
    from mogptk_amd import gpr, synth
    from mogptk_amd import dist as D
    from oracle.table_model import TableDevice
    from helpers import load, fixture_params, load_raw
    L.ExactHandle = TableDevice
    dist.init_process_group("gloo")
    # the reference's Titsias golden (titsias.npz, case 0) evaluated DATA-PARALLEL: every rank holds every world-th point
    fx = load("titsias.npz")
    pre = "c0_"
    C, Q, Dm, _ = [int(v) for v in fx[pre + "meta"]]
    fp = fixture_params(fx, pre)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=Dm)
    Zspec = fx[pre + "Zspec"]
    Zspec = int(Zspec[0]) if bool(fx[pre + "Zspec_is_int"]) else [int(z) for z in Zspec]
    m = gpr.Titsias(k, fx[pre + "X"], fx[pre + "y"], Z=Zspec, variance=float(fp[-1]["cons"]) ** 2, jitter=float(fx[pre + "jitter"]))
    load_raw(m.parameters(), fp)
    l0 = float(m.loss()); g0 = [None if p.grad is None else p.grad.copy() for p in m.parameters()]
    Xs = fx[pre + "Xs"]
    mu0, var0 = m.predict_f(Xs)
    comm = D.Comm(None, native=True)                      # the twin has no native library: its reductions go through the group directly
    comm.force = True
    TableDevice.reduce = staticmethod(lambda a: comm.all_reduce_host(a))
    gpr.config.comm = comm
    l1 = float(m.loss()); g1 = [None if p.grad is None else p.grad.copy() for p in m.parameters()]
    n_local = m._handle.X.shape[0]
    mu1, var1 = m.predict_f(Xs)
    err = max(float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) for a, b in zip(g1, g0) if b is not None)
    gold = max(float(np.max(np.abs(p.grad - f["grad"])) / max(1.0, np.max(np.abs(f["grad"])))) for p, f in zip(m.parameters(), fp) if f["grad"] is not None)
    perr = max(float(np.max(np.abs(mu1 - mu0))), float(np.max(np.abs(var1 - var0))))
    gpr.config.comm = None
    l2 = float(m.loss())                                  # and back: the handle is rebuilt on all points
    # the Snelson (FITC) model of snelson.npz, case 0 (per-channel noise), the same way
    fs = load("snelson.npz")
    C, Q, Dm, _ = [int(v) for v in fs["c0_meta"]]
    ks = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=Dm)
    ms = gpr.Snelson(ks, fs["c0_X"], fs["c0_y"], Z=fs["c0_Z"], variance=(fs["c0_variance"] if fs["c0_variance"].ndim else float(fs["c0_variance"])), jitter=float(fs["c0_jitter"]))
    fps = fixture_params(fs, "c0_")
    load_raw(ms.parameters(), fps)
    s0 = float(ms.loss()); sg0 = [None if p.grad is None else p.grad.copy() for p in ms.parameters()]
    smu0, svar0 = ms.predict_f(fs["c0_Xs"])
    gpr.config.comm = comm
    s1 = float(ms.loss()); sg1 = [None if p.grad is None else p.grad.copy() for p in ms.parameters()]
    smu1, svar1 = ms.predict_f(fs["c0_Xs"])
    gpr.config.comm = None
    serr = max(float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) for a, b in zip(sg1, sg0) if b is not None)
    sgold = max(float(np.max(np.abs(p.grad - f["grad"])) / max(1.0, np.max(np.abs(f["grad"])))) for p, f in zip(ms.parameters(), fps) if f["grad"] is not None)
    sperr = max(float(np.max(np.abs(smu1 - smu0))), float(np.max(np.abs(svar1 - svar0))))
    if dist.get_rank() == 0:
        print(json.dumps(dict(l0=l0, l1=l1, l2=l2, err=err, gold=gold, perr=perr, n_local=n_local, n=int(fx[pre + "X"].shape[0]),
                              world=dist.get_world_size(), ref=float(fx[pre + "loss"]),
                              snelson=dict(l0=s0, l1=s1, err=serr, gold=sgold, perr=sperr, ref=float(fs["c0_loss"])))))
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("ranks", [2, 3])
def test_data_parallel_titsias_gloo_ranks(tmp_path, ranks):
    """SURVEY 8e for the sparse bound: the training points split over the ranks, the sums over points all-reduced (numpy twin of
    mogp_titsias_eval_sharded / _predict_sharded through the product's own gpr.Titsias): equal to one process, and to the reference"""
    script = tmp_path / "worker_titsias.py"
    script.write_text(TITSIAS_WORKER % dict(root=ROOT, tests=os.path.join(ROOT, "tests")))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks,
                          "--master-addr", "127.0.0.1", "--master-port", str(29640 + ranks), str(script)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == ranks and r["n_local"] == (r["n"] + ranks - 1) // ranks          # rank 0 holds every world-th point
    assert abs(r["l1"] - r["l0"]) < 1e-10 * abs(r["l0"]) and r["err"] < 1e-9 and r["perr"] < 1e-9
    assert abs(r["l0"] - r["ref"]) < 1e-8 * abs(r["ref"]) and r["gold"] < 1e-6
    assert r["l2"] == r["l0"]
    t = r["snelson"]                                      # the FITC model data-parallel (twin of mogp_snelson_eval_sharded / _predict_sharded)
    assert abs(t["l1"] - t["l0"]) < 1e-10 * abs(t["l0"]) and t["err"] < 1e-9 and t["perr"] < 1e-9, t
    assert abs(t["l0"] - t["ref"]) < 1e-8 * abs(t["ref"]) and t["gold"] < 1e-6, t


SVGP_WORKER = textwrap.dedent('''
    import sys, json
    sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
    import numpy as np
    import torch.distributed as dist
    import mogptk_amd, mogptk_amd._lib as L
    from mogptk_amd import gpr
    from mogptk_amd import dist as D
    from oracle.table_model import TableDevice
    from helpers import load, fixture_params, load_raw
    L.ExactHandle = TableDevice
    dist.init_process_group("gloo")
    # the reference's SparseHensman goldens with non-Gaussian likelihoods (likelihoods.npz) evaluated DATA-PARALLEL
    fx = load("likelihoods.npz")
    res = {}
    for tag in ("svgp_studentt", "svgp_multi"):
        pre = tag + "_"
        C, Q, Dm, _ = [int(v) for v in fx[pre + "meta"]]
        lik = gpr.StudentTLikelihood(dof=4, scale=0.4) if tag == "svgp_studentt" else gpr.MultiOutputLikelihood(gpr.PoissonLikelihood(), gpr.GaussianLikelihood(0.3))
        k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=Dm)
        m = gpr.SparseHensman(k, fx[pre + "X"], fx[pre + "y"], Z=fx[pre + "Z"], likelihood=lik, jitter=1e-6)
        fp = fixture_params(fx, pre)
        load_raw(m.parameters(), fp)
        gpr.config.comm = None
        l0 = float(m.loss()); g0 = [None if p.grad is None else p.grad.copy() for p in m.parameters()]
        comm = D.Comm(None, native=True)
        comm.force = True
        TableDevice.reduce = staticmethod(lambda a: comm.all_reduce_host(a))
        gpr.config.comm = comm
        l1 = float(m.loss()); g1 = [None if p.grad is None else p.grad.copy() for p in m.parameters()]
        e1 = float(m.log_marginal_likelihood())
        mu, var = m.predict_f(fx[pre + "Xs"])
        res[tag] = dict(l0=l0, l1=l1, e1=e1, ref=float(fx[pre + "loss"]), n_local=int(m._handle.X.shape[0]),
                        err=max(float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) for a, b in zip(g1, g0) if b is not None),
                        gold=max(float(np.max(np.abs(p.grad - f["grad"])) / max(1.0, np.max(np.abs(f["grad"])))) for p, f in zip(m.parameters(), fp) if f["grad"] is not None),
                        perr=float(max(np.max(np.abs(mu - fx[pre + "mu_f"])), np.max(np.abs(var - fx[pre + "var_f"])))))
    if dist.get_rank() == 0:
        print(json.dumps(dict(world=dist.get_world_size(), **res)))
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("ranks", [2, 3])
def test_data_parallel_sparse_hensman_gloo_ranks(tmp_path, ranks):
    """the variational sparse model with non-Gaussian likelihoods, its training points split over the ranks: the device algebra's sums over
    points (numpy twin of mogp_svgp_backward_sharded) and the likelihood's expectation / parameter gradients all-reduced -- equal to one
    process and to the reference's autograd"""
    script = tmp_path / "worker_svgp.py"
    script.write_text(SVGP_WORKER % dict(root=ROOT, tests=os.path.join(ROOT, "tests")))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks,
                          "--master-addr", "127.0.0.1", "--master-port", str(29650 + ranks), str(script)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == ranks
    for tag in ("svgp_studentt", "svgp_multi"):
        t = r[tag]
        assert t["n_local"] == (60 + ranks - 1) // ranks
        assert abs(t["l1"] - t["l0"]) < 1e-10 * abs(t["l0"]) and t["err"] < 1e-9 and abs(t["e1"] + t["l1"]) < 1e-10 * abs(t["l1"]), t
        assert abs(t["l0"] - t["ref"]) < 1e-9 * abs(t["ref"]) and t["gold"] < 1e-7 and t["perr"] < 1e-8, t
