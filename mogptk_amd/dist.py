"""
Sharded exact-GP evaluation across the GPUs of one node: one process per GPU, `torch.distributed`
collectives (backend "nccl" = RCCL over xGMI in production) issued between the stage calls of the C ABI
(`mogp_shard_*`, include/mogp_hip.h).

    import torch.distributed as dist, mogptk_amd
    dist.init_process_group("nccl")
    mogptk_amd.use_distributed()          # every gpr.Exact.loss() of this process is now sharded over the group
    model.train(...)

What is exchanged per LML+gradient evaluation of an N-point model (Npad = N rounded up to 128, 512-wide pivot blocks):
  per pivot block: ONE all-gather of the block's column panel ((Npad - k0) x 512 doubles in total) and up to four broadcasts
  of the pivot tile rows (128 x k0 doubles each)  -- N^2 doubles per evaluation in total;
  once: all-reduce of alpha (Npad doubles), of the gradient moments (C(C+1)/2 x T x (2+3D)) and of diag sums (C).
Every rank holds the full training set and a full-size work matrix (8.6 GB at N = 32768: nothing against 288 GB), owns the
128-row tile rows i with i % world == rank, repeats the cheap serial chain (512 x 512 block inversions, panels) and applies
the rank-512 updates to its own rows only -- the O(N^3) work is divided by `world`.

With backend "gloo" (CPU tests, or several ranks sharing one GPU for validation) buffers are staged through the host.
"""
import ctypes
import numpy as np

from . import _lib


class Comm:
    """thin adapter over torch.distributed.  Buffers are opaque references handed out by the device handle; the handle's
    `mem_tensor / mem_get / mem_put` turn them into a device tensor (RCCL path) or move them through the host (gloo)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised: call dist.init_process_group first")
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device_path = dist.get_backend(group) == "nccl"
        self.force = False          # route even a 1-rank group through the sharded stages (validation of the RCCL plumbing)

    def _wait(self):
        # the collective is ordered after torch's current stream; wait for THAT stream only, so the library's bulk stream (still
        # applying the previous pivot block's update) keeps running underneath the exchange
        self.torch.cuda.current_stream().synchronize()

    def all_gather(self, h, send, recv, count):
        if self.device_path:
            self.dist.all_gather_into_tensor(h.mem_tensor(recv, count * self.world), h.mem_tensor(send, count), group=self.group)
            self._wait()
        else:
            s = self.torch.from_numpy(h.mem_get(send, count))
            r = self.torch.empty(count * self.world, dtype=self.torch.float64)
            self.dist.all_gather_into_tensor(r, s, group=self.group)
            h.mem_put(recv, r.numpy())

    def broadcast(self, h, buf, count, src):
        if count == 0:
            return
        if self.device_path:
            self.dist.broadcast(h.mem_tensor(buf, count), src=src, group=self.group)
            self._wait()
        else:
            t = self.torch.from_numpy(h.mem_get(buf, count) if self.rank == src else np.empty(int(count)))
            self.dist.broadcast(t, src=src, group=self.group)
            if self.rank != src:
                h.mem_put(buf, t.numpy())

    def all_reduce_buf(self, h, buf, count):
        if self.device_path:
            self.dist.all_reduce(h.mem_tensor(buf, count), group=self.group)
            self._wait()
        else:
            t = self.torch.from_numpy(h.mem_get(buf, count))
            self.dist.all_reduce(t, group=self.group)
            h.mem_put(buf, t.numpy())

    def all_reduce_host(self, arr):
        t = self.torch.from_numpy(arr)
        if self.device_path:
            t = t.cuda()
            self.dist.all_reduce(t, group=self.group)
            arr[...] = t.cpu().numpy()
        else:
            self.dist.all_reduce(t, group=self.group)
        return arr


def use_distributed(group=None):
    """shard every exact LML+gradient evaluation of this process over the ranks of `group` (default: the world)"""
    from .gpr.config import config
    config.comm = Comm(group)
    return config.comm


def use_single_device():
    from .gpr.config import config
    config.comm = None


def sharded_eval(h, comm, noise_var, jitter, data_var=None):
    """mogp_exact_eval(..., MOGP_EVAL_GRAD) sharded over comm.world ranks; same return dict on every rank.
    `h` is a device handle (mogptk_amd._lib.ExactHandle, or its numpy twin in the tests) exposing the shard_* stages."""
    import os, time
    prof = os.environ.get("MOGP_SHARD_PROFILE")
    tm = dict(begin=0.0, pack=0.0, gather=0.0, unpack_block=0.0, alpha=0.0, finish=0.0)
    t0 = time.perf_counter()
    jit, nblocks = h.shard_begin(comm.rank, comm.world, noise_var, jitter, data_var)
    t1 = time.perf_counter(); tm["begin"] += t1 - t0
    for kb in range(nblocks):
        t0 = time.perf_counter()
        send, recv, count = h.shard_pack(kb)
        t1 = time.perf_counter(); tm["pack"] += t1 - t0
        comm.all_gather(h, send, recv, count)
        t2 = time.perf_counter(); tm["gather"] += t2 - t1
        h.shard_unpack(kb)
        h.shard_block(kb)
        tm["unpack_block"] += time.perf_counter() - t2
    t0 = time.perf_counter()
    buf, count = h.shard_alpha()
    comm.all_reduce_buf(h, buf, count)
    t1 = time.perf_counter(); tm["alpha"] += t1 - t0
    lml, moments, diagG = h.shard_finish()
    tm["finish"] += time.perf_counter() - t1
    if prof and comm.rank == 0:
        print("sharded_eval ms:", {k: round(1e3 * v, 2) for k, v in tm.items()}, flush=True)
    comm.all_reduce_host(moments)
    comm.all_reduce_host(diagG)
    return dict(lml=lml, moments=moments, diagG=diagG, trG=float(np.sum(diagG)), jitter_abs=jit)
