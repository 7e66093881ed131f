"""
Sharded exact-GP evaluation and prediction across the GPUs of one node: one process per GPU.

    import torch.distributed as dist, mogptk_amd
    dist.init_process_group("nccl")       # or "gloo": only used here to hand 128 bytes around and, with gloo, as the transport
    mogptk_amd.use_distributed()          # every gpr.Exact.loss() / predict_f() of this process is now sharded over the group
    model.train(...)

The collectives of the hot path are issued by the native library itself (mogptk_amd/csrc/comm.hip), not by torch:
  * group backend "nccl"  -> the library opens its own RCCL communicator (unique id from rank 0, broadcast once through the group) and
    enqueues ncclAllGather / ncclAllReduce on its own HIP streams: no host round trip inside an evaluation;
  * any other backend     -> two callbacks that stage the device buffers through the host and call the group's collectives (tests with
    several ranks sharing one GPU, which RCCL refuses; machines without a device transport).
What is exchanged per LML+gradient evaluation of an N-point model (Npad = N rounded up to 128, 512-wide pivot blocks): per pivot block
ONE all-gather -- the block's column panel, each tile row from its owner, plus the part left of the block of the pivot tile rows (N^2
doubles per evaluation in total); once: all-reduce of alpha (Npad doubles), of the gradient moments (C(C+1)/2 x T x (2+3D)) and of the
diagonal sums (C).  Every rank holds the full training set but only ITS tile rows of the work matrix (the 128-row tile rows i % world == rank:
1 / world of 8 Npad^2 bytes, physical memory under those rows only -- mogp_model_work_bytes; what it needs of the other ranks' rows arrives with the
exchange and goes straight into the pivot block's work buffers), BUILDS only the Gram / moment tiles of those rows, repeats the cheap serial chain
(512 x 512 block inversions, panels) and applies the rank-512 updates to its own rows only: the O(N^3) work and the O(N^2) memory are divided by
`world`.  A sharded prediction keeps it that way: every rank forms its rows' share of the quadratic form K_s. Kj^-1 K_.s for all test points, one all-reduce of
S doubles makes the variances (no all-gather of the inverse).

`sharded_eval` below is the same protocol spelt out stage by stage over the `mogp_shard_*` entry points; the numpy twin of the device
stages (oracle/table_model.py) runs it under gloo on CPU ranks in tests/test_dist_cpu.py.
"""
import ctypes
import numpy as np

from . import _lib


class Comm:
    """The process group as the rest of the package sees it: rank / world, and whether the native library owns the collectives
    (`native`), in which case ExactHandle calls mogp_exact_eval_sharded / mogp_exact_predict_sharded directly."""

    def __init__(self, group=None, native=False):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised: call dist.init_process_group first")
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.native = native
        self.force = False          # route even a 1-rank group through the sharded path (validation of the plumbing)
        self._keep = []             # ctypes callbacks must outlive the communicator

    # ---- host-staged collectives on raw device pointers (external back end of comm.hip, and the stage-by-stage protocol below) ----
    def _get(self, ptr, count):
        h = np.empty(int(count), dtype=np.float64)
        _lib.check(_lib.lib().mogp_dev_copy(h.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), 8 * int(count), 0))
        return h

    def _put(self, ptr, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        _lib.check(_lib.lib().mogp_dev_copy(ctypes.c_void_p(ptr), arr.ctypes.data_as(ctypes.c_void_p), 8 * arr.size, 1))

    def _cb_allgather(self, user, send, recv, count):
        try:
            s = self.torch.from_numpy(self._get(send, count))
            r = self.torch.empty(int(count) * self.world, dtype=self.torch.float64)
            self.dist.all_gather_into_tensor(r, s, group=self.group)
            self._put(recv, r.numpy())
            return 0
        except Exception:           # a Python exception must not unwind through the C frame
            import traceback
            traceback.print_exc()
            return 1

    def _cb_allreduce(self, user, buf, count):
        try:
            t = self.torch.from_numpy(self._get(buf, count))
            self.dist.all_reduce(t, group=self.group)
            self._put(buf, t.numpy())
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    # ---- collectives of the stage-by-stage protocol (device handle or numpy twin) ----
    def all_gather(self, h, send, recv, count):
        s = self.torch.from_numpy(h.mem_get(send, count))
        r = self.torch.empty(count * self.world, dtype=self.torch.float64)
        self.dist.all_gather_into_tensor(r, s, group=self.group)
        h.mem_put(recv, r.numpy())

    def all_reduce_buf(self, h, buf, count):
        t = self.torch.from_numpy(h.mem_get(buf, count))
        self.dist.all_reduce(t, group=self.group)
        h.mem_put(buf, t.numpy())

    def all_reduce_host(self, arr):
        """sum a small host array over the ranks, in place (a device round trip when the group's backend only takes device tensors)"""
        t = self.torch.from_numpy(arr)
        if self.dist.get_backend(self.group) == "nccl":
            d = t.cuda()
            self.dist.all_reduce(d, group=self.group)
            t.copy_(d.cpu())
        else:
            self.dist.all_reduce(t, group=self.group)
        return arr


def _broadcast_bytes(comm, payload):
    """rank 0's bytes to every rank of the group, through whatever backend the group has"""
    torch, dist = comm.torch, comm.dist
    on_gpu = dist.get_backend(comm.group) == "nccl"
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
    if on_gpu:
        t = t.cuda()
    src = dist.get_global_rank(comm.group, 0) if comm.group is not None else 0
    dist.broadcast(t, src=src, group=comm.group)
    return bytes(t.cpu().numpy().tobytes())


_native = {}        # (group id, transport, device) -> Comm whose communicator lives in the native context


def use_distributed(group=None, transport="auto"):
    """Shard every exact LML+gradient evaluation and prediction of this process over the ranks of `group` (default: the world).
    transport: "rccl" (the library's own RCCL communicator), "host" (callbacks staging through the host over the group's backend), or
    "auto" (rccl when the group's backend is nccl, else host).  The communicator is created once per (group, transport) and reused."""
    from .gpr.config import config
    comm = Comm(group, native=True)
    if transport == "auto":
        transport = "rccl" if comm.dist.get_backend(group) == "nccl" else "host"
    key = (id(group), transport, config.device)
    if key in _native:
        config.comm = _native[key]
        return config.comm
    l = _lib.lib()
    ctx = _lib.context(config.device)
    if transport == "rccl":
        ident = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        if comm.rank == 0:
            _lib.check(l.mogp_comm_unique_id(ident))
        payload = _broadcast_bytes(comm, ident.raw)
        buf = ctypes.create_string_buffer(payload, _lib.COMM_ID_BYTES)
        _lib.check(l.mogp_comm_init_rccl(ctx, buf, comm.rank, comm.world))
    elif transport == "host":
        ag, ar = _lib.ALLGATHER_CB(comm._cb_allgather), _lib.ALLREDUCE_CB(comm._cb_allreduce)
        comm._keep = [ag, ar]
        _lib.check(l.mogp_comm_init_external(ctx, comm.rank, comm.world, ctypes.cast(ag, ctypes.c_void_p), ctypes.cast(ar, ctypes.c_void_p), None))
    else:
        raise ValueError("transport must be 'auto', 'rccl' or 'host'")
    comm.transport = transport
    _native.clear()             # a context holds one communicator: a new one replaces whatever was there
    _native[key] = comm
    config.comm = comm
    return comm


def use_protocol(group=None):
    """Route evaluations through the stage-by-stage protocol `sharded_eval` below (collectives issued here, buffers staged through the
    host).  This is what the CPU tests drive with the numpy twin of the device stages; on a device it is the slow, explicit form."""
    from .gpr.config import config
    config.comm = Comm(group, native=False)
    return config.comm


def use_single_device():
    """evaluations run on this process's GPU alone again (the native communicator, if any, stays alive for the next use_distributed)"""
    from .gpr.config import config
    config.comm = None


def shutdown_distributed():
    """destroy the native communicator (call before dist.destroy_process_group)"""
    from .gpr.config import config
    config.comm = None
    if _native:
        _native.clear()
        _lib.check(_lib.lib().mogp_comm_destroy(_lib.context(config.device)))


def sharded_eval(h, comm, noise_var, jitter, data_var=None):
    """mogp_exact_eval(..., MOGP_EVAL_GRAD) sharded over comm.world ranks, one stage at a time; same return dict on every rank.
    `h` is a device handle (mogptk_amd._lib.ExactHandle) or its numpy twin exposing the shard_* stages."""
    jit, nblocks = h.shard_begin(comm.rank, comm.world, noise_var, jitter, data_var)
    for kb in range(nblocks):
        send, recv, count = h.shard_pack(kb)
        comm.all_gather(h, send, recv, count)
        h.shard_unpack(kb)
        h.shard_block(kb)
    buf, count = h.shard_alpha()
    comm.all_reduce_buf(h, buf, count)
    lml, moments, diagG = h.shard_finish()
    comm.all_reduce_host(moments)
    comm.all_reduce_host(diagG)
    return dict(lml=lml, moments=moments, diagG=diagG, trG=float(np.sum(diagG)), jitter_abs=jit)
