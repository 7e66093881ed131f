"""The task graph of the dataflow schedule (csrc/flow.hip), replayed on the CPU from the numbers mogp_flow_plan returns:
  * every queue is sorted by the sequential algorithm's key and every dependency is produced by tasks with a smaller key (=> the head-of-queue
    rule can never deadlock, however many workgroups are resident);
  * executed with many concurrent "workgroups" in random order under exactly the device's rule (take the first queue, by priority, whose HEAD has
    its counters satisfied), no two tasks in flight ever touch a tile one of them writes (the dependencies cover every hazard);
  * the numbers that come out are the Cholesky factor's inverse and the inverse of the matrix (numpy tiles standing in for the MFMA products).
No device work: the plan is a host function of the tile count."""
import ctypes
import numpy as np
import pytest

from mogptk_amd import _lib

T = 128
W = 24


def plan(nb, rhs=0):
    """rhs > 0: the prediction's plan -- factorisation + forward substitution of rhs tile rows of right-hand sides (T in buffer 2, X in buffer 4)"""
    l = _lib.lib()
    cnt = ctypes.c_int64(0)
    assert l.mogp_flow_plan_rhs(nb, rhs, None, 0, ctypes.byref(cnt)) == 0
    out = np.zeros((cnt.value, W), dtype=np.int64)
    assert l.mogp_flow_plan_rhs(nb, rhs, out.ctypes.data_as(_lib.c_i64p), out.size, ctypes.byref(cnt)) == 0
    return out


def split(rows):
    chains = [r for r in rows if r[0] < 0]                  # the private stream's launches, in stream order
    nq = int(rows[:, 0].max()) + 1
    queues = [[r for r in rows if r[0] == q] for q in range(nq)]
    return chains, queues


def deps_of(r):
    return [(int(r[14 + d]), int(r[18 + d])) for d in range(int(r[13]))]


def signals_of(r):
    if r[0] == -1:                                           # chain kernel: its workgroups count themselves off
        return [(int(r[22]), int(r[23]))]
    if r[0] == -2:                                           # mini-panel: 2 nk tiles per panel row of the next block
        return [(int(r[22]) + i, int(r[23])) for i in range(int(r[6]))]
    if r[0] == -3:
        return []
    return [(int(r[22 + s]), 1) for s in range(2) if r[22 + s] >= 0]


@pytest.mark.parametrize("nb,rhs", [(3, 0), (4, 0), (5, 0), (9, 0), (12, 0), (14, 0), (96, 0), (112, 0),      # 96, 112: more deadline queues than the kernel has -- the farthest are folded into one
                                    (3, 1), (5, 2), (9, 3), (14, 5), (128, 33)])                                   # with right-hand sides (128, 33: BASELINE.json configs[3])
def test_keys_give_a_topological_order_and_queues_follow_it(nb, rhs):
    rows = plan(nb, rhs)
    chains, queues = split(rows)
    for q in queues:
        keys = [int(r[1]) for r in q]
        assert keys == sorted(keys)
    flags = {}
    order = sorted(range(len(rows)), key=lambda k: int(rows[k][1]))
    i = 0
    while i < len(order):                       # tasks of equal key: checked against the state BEFORE any of them ran
        j = i
        while j < len(order) and rows[order[j]][1] == rows[order[i]][1]:
            j += 1
        for k in order[i:j]:
            for idx, need in deps_of(rows[k]):
                assert flags.get(idx, 0) >= need, (nb, rhs, rows[k])
        for k in order[i:j]:
            for idx, inc in signals_of(rows[k]):
                flags[idx] = flags.get(idx, 0) + inc
        i = j


def tile(buf, r, c, nr=1, nc=1):
    return buf[r * T:(r + nr) * T, c * T:(c + nc) * T]


def footprint(r):
    """(reads, write): sets of (buffer, tile row, tile col); pseudo-buffers 5 = z (per tile row), 6 = alpha shares (row block, column group)"""
    var, kt = int(r[11]), int(r[12])
    if var & 32:
        if var & 1:                                  # alpha share of row block br: rows ar .. ar + kt - 1, column group ac
            cols = range(4 * int(r[4]), 4 * int(r[4]) + 4)
            reads = {(3, int(r[3]) + k, c) for k in range(kt) for c in cols if c <= int(r[3]) + k} | {(5, int(r[3]) + k, 0) for k in range(kt)}
            return reads, (6, int(r[6]), int(r[4]))
        i = int(r[3])
        return {(3, i, c) for c in range(i + 1)}, (5, i, 0)
    lay, nk = var & 3, (kt * 16 + T - 1) // T
    reads = set()
    if lay in (0, 1):
        reads |= {(int(r[2]), int(r[3]), int(r[4]) + k) for k in range(nk)}
    else:
        reads |= {(int(r[2]), int(r[3]) + k, int(r[4])) for k in range(nk)}
    if lay == 0:
        reads |= {(int(r[5]), int(r[6]), int(r[7]) + k) for k in range(nk)}
    else:
        reads |= {(int(r[5]), int(r[6]) + k, int(r[7])) for k in range(nk)}
    wr = (int(r[8]), int(r[9]), int(r[10]))
    if not var & 4:
        reads.add(wr)
    return reads, wr


def run_task(bufs, r):
    var, kt = int(r[11]), int(r[12])
    if var & 32:
        Wm, y, z, part = bufs[3], bufs[7], bufs[5], bufs[6]
        if var & 1:
            r0, nr, rb, jg = int(r[3]) * T, kt * T, int(r[6]), int(r[4])
            c0, c1 = jg * 512, min(jg * 512 + 512, Wm.shape[1])
            part[rb, c0:c1] = Wm[r0:r0 + nr, c0:c1].T @ z[r0:r0 + nr]
        else:
            i = int(r[3])
            z[i * T:(i + 1) * T] = Wm[i * T:(i + 1) * T, :(i + 1) * T] @ y[:(i + 1) * T]
        return
    lay, K = var & 3, kt * 16
    a, b = bufs[int(r[2])], bufs[int(r[5])]
    ar, ac, br, bc = int(r[3]) * T, int(r[4]) * T, int(r[6]) * T, int(r[7]) * T
    if lay == 0:
        prod = a[ar:ar + T, ac:ac + K] @ b[br:br + T, bc:bc + K].T
    elif lay == 1:
        prod = a[ar:ar + T, ac:ac + K] @ b[br:br + K, bc:bc + T]
    else:
        prod = a[ar:ar + K, ac:ac + T].T @ b[br:br + K, bc:bc + T]
    alpha = -1.0 if var & 8 else 1.0
    c = tile(bufs[int(r[8])], int(r[9]), int(r[10]))
    if var & 4:
        c[...] = alpha * prod
    else:
        c += alpha * prod


def priv_footprint(r):
    """(reads, writes) of a launch of the private stream"""
    k0, nk = int(r[3]), int(r[4])
    diag = {(k0 + i, k0 + j) for i in range(nk) for j in range(i + 1)}
    if r[0] == -1:
        return {(0, i, j) for i, j in diag}, {(3, i, j) for i, j in diag} | {(0, i, j) for i, j in diag}
    k1, na = int(r[5]), int(r[6])
    panel = {(1, k1 + i, k0 + c) for i in range(na) for c in range(nk)}
    if r[0] == -2:
        return {(0, k1 + i, k0 + c) for i in range(na) for c in range(nk)} | {(3, i, j) for i, j in diag}, panel
    nxt = {(0, k1 + i, k1 + j) for i in range(na) for j in range(i + 1)}
    return panel | nxt, nxt


def run_priv(bufs, r):
    """mini-panel L[rows of the next block][K] = A[rows][K] W_KK^T, or the next diagonal block's last update"""
    k0, nk, k1, na = int(r[3]), int(r[4]), int(r[5]), int(r[6])
    A, Lm, Wm = bufs[0], bufs[1], bufs[3]
    if r[0] == -2:
        tile(Lm, k1, k0, na, nk)[...] = tile(A, k1, k0, na, nk) @ np.tril(tile(Wm, k0, k0, nk, nk)).T
    else:
        P = tile(Lm, k1, k0, na, nk)
        tile(A, k1, k1, na, na)[...] -= P @ P.T


def run_chain(bufs, r):
    k0, nk = int(r[3]), int(r[4])
    A, Wm = bufs[0], bufs[3]
    D = np.tril(tile(A, k0, k0, nk, nk))
    D = D + np.tril(D, -1).T
    L = np.linalg.cholesky(D)
    tile(Wm, k0, k0, nk, nk)[...] = np.linalg.inv(L)
    return L


def replay(rows, bufs, nwg, rng):
    """the device's rule with nwg concurrent workgroups finishing in random order; asserts that nothing in flight is written or overwritten; returns
    the chain kernels' log-determinant parts"""
    chains, queues = split(rows)
    flags = {}
    heads = [0] * len(queues)
    chain_next = 0
    running = []                     # (row, reads, write)
    logdet_parts = []

    def ready(r):
        return all(flags.get(i, 0) >= n for i, n in deps_of(r))

    def start(r):
        if r[0] < 0:
            reads, wr_set = priv_footprint(r)
        else:
            reads, wr = footprint(r)
            wr_set = {wr}
        for _, oreads, owr in running:           # nobody in flight writes what this one touches, nobody reads what this one writes
            assert not (owr & (reads | wr_set)), ("write in flight", r)
            assert not (oreads & wr_set), ("read in flight", r)
        running.append((r, reads, wr_set))

    def finish(k):
        r, _, _ = running.pop(k)
        if r[0] == -1:
            L = run_chain(bufs, r)
            logdet_parts.append(np.log(np.diag(L)).sum())
        elif r[0] < 0:
            run_priv(bufs, r)
        else:
            run_task(bufs, r)
        for idx, inc in signals_of(r):
            flags[idx] = flags.get(idx, 0) + inc

    total = len(rows)
    done = 0
    chain_busy = False
    while done < total:
        progressed = False
        # the private stream: one chain kernel at a time, in order
        if not chain_busy and chain_next < len(chains) and ready(chains[chain_next]) and rng.random() < 0.7:
            start(chains[chain_next]); chain_next += 1; chain_busy = True; progressed = True
        # free workgroups take the first ready head, by priority
        nflow = sum(1 for r, _, _ in running if r[0] >= 0)
        while nflow < nwg:
            took = False
            for q, tasks in enumerate(queues):
                if heads[q] < len(tasks) and ready(tasks[heads[q]]):
                    start(tasks[heads[q]]); heads[q] += 1; nflow += 1; took = True; progressed = True
                    break
            if not took:
                break
        if running:                              # somebody finishes (random order)
            nfin = int(rng.integers(1, max(2, len(running) // 2 + 1)))
            for _ in range(min(nfin, len(running))):
                k = int(rng.integers(0, len(running)))
                if running[k][0][0] < 0:
                    chain_busy = False
                finish(k); done += 1
            progressed = True
        if not progressed and not chain_busy and chain_next < len(chains) and ready(chains[chain_next]):
            start(chains[chain_next]); chain_next += 1; chain_busy = True; progressed = True       # (the coin above said "not yet")
        assert progressed, "deadlock: nothing ready, nothing running, %d of %d tasks done" % (done, total)
    return logdet_parts


def spd(N, rng):
    G = rng.standard_normal((N, N // 2))
    return G @ G.T / (N // 2) + 0.5 * np.eye(N)


@pytest.mark.parametrize("nb,nwg,seed", [(5, 7, 0), (9, 40, 1), (12, 96, 2), (14, 1, 3), (10, 480, 4)])
def test_replay_with_concurrent_workgroups(nb, nwg, seed):
    rng = np.random.default_rng(seed)
    N = nb * T
    K = spd(N, rng)
    nan = np.full((N, N), np.nan)
    yv = rng.standard_normal(N)
    nblk = (nb + 3) // 4
    bufs = [np.tril(K).copy(), nan.copy(), nan.copy(), np.zeros((N, N)), nan.copy(),      # A lower, L, Wt (never read before written), Wm zero, B
            np.full(N, np.nan), np.zeros((nblk, N)), yv]                                   # z, alpha shares per row block, y
    logdet_parts = replay(plan(nb), bufs, nwg, rng)
    Lref = np.linalg.cholesky(K)
    Wref = np.linalg.inv(Lref)
    Kinv = np.linalg.inv(K)
    assert np.max(np.abs(np.tril(bufs[3]) - Wref)) < 1e-9 * np.max(np.abs(Wref))
    assert np.max(np.abs(np.triu(bufs[3], 1))) == 0.0
    assert np.max(np.abs(np.tril(bufs[4]) - np.tril(Kinv))) < 1e-9 * np.max(np.abs(Kinv))
    assert abs(sum(logdet_parts) - np.log(np.diag(Lref)).sum()) < 1e-8
    alpha = np.array([bufs[6][(c // T) // 4:, c].sum() for c in range(N)])            # k_alpha_sum: the shares of the row blocks from the column's own on
    assert np.max(np.abs(bufs[5] - Wref @ yv)) < 1e-9 * np.max(np.abs(Wref @ yv))
    assert np.max(np.abs(alpha - Kinv @ yv)) < 1e-8 * np.max(np.abs(Kinv @ yv))


@pytest.mark.parametrize("nb,rhs,nwg,seed", [(5, 2, 7, 0), (9, 3, 40, 1), (12, 5, 96, 2), (14, 2, 1, 3), (10, 4, 480, 4), (3, 1, 16, 5)])
def test_replay_of_the_prediction_plan(nb, rhs, nwg, seed):
    """factorisation + forward substitution X L^T = T of rhs tile rows of right-hand sides: the same rule, the same hazard checks; X = T L^-T comes out"""
    rng = np.random.default_rng(100 + seed)
    N = nb * T
    K = spd(N, rng)
    T0 = rng.standard_normal((rhs * T, N))
    nan = np.full((N, N), np.nan)
    Tb = nan.copy(); Tb[:rhs * T] = T0
    bufs = [np.tril(K).copy(), nan.copy(), Tb, np.zeros((N, N)), nan.copy(), np.full(N, np.nan), np.zeros(((nb + 3) // 4, N)), np.zeros(N)]
    rows = plan(nb, rhs)
    assert not any(int(r[11]) & 32 for r in rows if r[0] >= 0)                       # no vector tasks, and nothing of the inverse:
    assert not any(r[0] >= 0 and int(r[8]) == 4 and int(r[2]) == 3 and int(r[5]) == 3 for r in rows)      # no W^T W accumulations
    logdet_parts = replay(rows, bufs, nwg, rng)
    Lref = np.linalg.cholesky(K)
    Xref = np.linalg.solve(Lref, T0.T).T
    assert np.max(np.abs(bufs[4][:rhs * T] - Xref)) < 1e-9 * np.max(np.abs(Xref))
    for b in range((nb + 3) // 4):                                                  # the panels L[below block b][block b], where the substitution read them
        k0, k1 = 4 * b, min(nb, 4 * b + 4)
        if k1 < nb:
            assert np.max(np.abs(bufs[1][k1 * T:, k0 * T:k1 * T] - Lref[k1 * T:, k0 * T:k1 * T])) < 1e-9 * np.max(np.abs(Lref))
    assert abs(sum(logdet_parts) - np.log(np.diag(Lref)).sum()) < 1e-8


def happens_before(rows):
    """ancestors[k] = bitset of the tasks that must have FINISHED before task k may start: producers of its counters (a counter that needs n
    has, by construction, exactly its n earliest producers in key order behind it), and the previous chain kernel (same stream)."""
    producers = {}
    for k, r in enumerate(rows):
        for idx, inc in signals_of(r):
            producers.setdefault(idx, []).append((int(r[1]), k, inc))
    for v in producers.values():
        v.sort()
    order = sorted(range(len(rows)), key=lambda k: int(rows[k][1]))
    anc = [0] * len(rows)
    prev_chain = None
    for k in order:
        r = rows[k]
        a = 0
        for idx, need in deps_of(r):
            got = 0
            for key, p, inc in producers.get(idx, []):
                if got >= need:
                    break
                assert key < int(r[1])
                a |= anc[p] | (1 << p)
                got += inc
            assert got >= need
        if r[0] < 0:                                         # the private stream runs its launches one after the other
            if prev_chain is not None:
                a |= anc[prev_chain] | (1 << prev_chain)
            prev_chain = k
        anc[k] = a
    return anc


@pytest.mark.parametrize("nb,rhs", [(6, 0), (13, 0), (22, 0), (6, 2), (13, 4), (22, 3)])
def test_every_conflicting_pair_of_tasks_is_ordered_by_the_counters(nb, rhs):
    rows = plan(nb, rhs)
    anc = happens_before(rows)
    touch = {}
    for k, r in enumerate(rows):
        if r[0] < 0:
            reads, writes = priv_footprint(r)
        else:
            reads, wr = footprint(r)
            writes = {wr}
        for t in reads - writes:
            touch.setdefault(t, ([], []))[0].append(k)
        for t in writes:
            touch.setdefault(t, ([], []))[1].append(k)
    pairs = 0
    for t, (rd, wr) in touch.items():
        for w in wr:
            for o in rd + wr:
                if o == w:
                    continue
                pairs += 1
                assert (anc[w] >> o) & 1 or (anc[o] >> w) & 1, ("unordered tasks on tile", t, rows[w], rows[o])
    assert pairs > 0
