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
