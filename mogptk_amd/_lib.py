"""
ctypes binding of libmogp_hip.so (include/mogp_hip.h).  This is the only place the package touches the
native library.  There is NO fallback: if the library is missing, or no gfx950 device is visible when a
compute entry point is called, the call fails loudly.
"""
import atexit
import ctypes
import os
import threading
import weakref
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOGP_LIB_PATH") or os.path.join(_HERE, "csrc", "libmogp_hip.so")      # (MOGP_LIB_PATH: another build of the same library, for A/B runs)

MOGP_OK, MOGP_EINVAL, MOGP_EHIP, MOGP_ENOTPD, MOGP_ENONFINITE, MOGP_ENODEVICE = 0, -1, -2, -3, -4, -5
MOGP_EVAL_GRAD = 1
COMM_ID_BYTES = 128
ALLGATHER_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)
ALLREDUCE_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)
ST_GRAM, ST_POTRF, ST_TRTRI, ST_LAUUM, ST_SOLVE, ST_MOMENTS, ST_TOTAL, ST_GEMM_KERNEL, ST_GRAM_KERNEL, ST_MOMENT_KERNEL, ST_COUNT = range(11)

_lib = None
_lock = threading.RLock()
_ctx = {}
_live = weakref.WeakSet()             # open ExactHandles (shutdown() closes them before their contexts go)

c_dp = ctypes.POINTER(ctypes.c_double)
c_i64p = ctypes.POINTER(ctypes.c_int64)

# name -> (restype, argtypes); mirrors include/mogp_hip.h one to one (tests check every symbol is exported)
SIGNATURES = {
    "mogp_version": (ctypes.c_char_p, []),
    "mogp_last_error": (ctypes.c_char_p, []),
    "mogp_device_count": (ctypes.c_int, []),
    "mogp_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "mogp_ctx_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mogp_ctx_device_name": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]),
    "mogp_model_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, c_dp, c_dp,
                                         ctypes.POINTER(ctypes.c_void_p)]),
    "mogp_model_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mogp_model_set_y": (ctypes.c_int, [ctypes.c_void_p, c_dp]),
    "mogp_model_set_terms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_dp]),
    "mogp_model_set_terms_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_dp]),
    "mogp_model_set_point_diag": (ctypes.c_int, [ctypes.c_void_p, c_dp]),
    "mogp_gram_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_dp,
                                    ctypes.c_int64, c_dp, ctypes.c_int64, c_dp, c_dp]),
    "mogp_gram": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_dp,
                                 ctypes.c_int64, c_dp, ctypes.c_int64, c_dp, c_dp]),
    "mogp_exact_eval": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, ctypes.c_double, ctypes.c_int,
                                       c_dp, c_dp, c_dp, c_dp, c_dp, c_i64p]),
    "mogp_exact_predict": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, ctypes.c_double, c_dp, ctypes.c_int64, c_dp,
                                          ctypes.c_int, c_dp, c_dp, c_i64p]),
    "mogp_titsias_eval": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp, ctypes.c_double, ctypes.c_double, c_dp, ctypes.c_int,
                                         c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_i64p]),
    "mogp_titsias_fetch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, c_dp]),
    "mogp_titsias_predict": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp, ctypes.c_double, ctypes.c_double, c_dp,
                                            ctypes.c_int64, c_dp, c_dp, c_dp, c_i64p]),
    "mogp_titsias_eval_sharded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp, ctypes.c_double, ctypes.c_double, c_dp, ctypes.c_int,
                                                 c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_i64p]),
    "mogp_titsias_predict_sharded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp, ctypes.c_double, ctypes.c_double, c_dp,
                                                    ctypes.c_int64, c_dp, c_dp, c_dp, c_i64p]),
    "mogp_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "mogp_comm_init_rccl": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "mogp_comm_init_external": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "mogp_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mogp_comm_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "mogp_comm_selftest": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "mogp_shard_stage_ms": (ctypes.c_int, [ctypes.c_void_p, c_dp]),
    "mogp_model_inverse_fraction": (ctypes.c_int, [ctypes.c_void_p, c_dp]),
    "mogp_model_pivot_range": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp]),
    "mogp_model_set_accurate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "mogp_exact_eval_sharded": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, ctypes.c_double, c_dp, c_dp, c_dp, c_dp, c_dp, c_i64p]),
    "mogp_exact_predict_sharded": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, ctypes.c_double, c_dp, ctypes.c_int64, c_dp, c_dp, c_dp, c_i64p]),
    "mogp_shard_config": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "mogp_model_work_bytes": (ctypes.c_int, [ctypes.c_void_p, c_i64p, c_i64p]),
    "mogp_shard_begin": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, ctypes.c_double, c_dp, ctypes.POINTER(ctypes.c_int)]),
    "mogp_shard_pack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_i64p]),
    "mogp_shard_unpack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "mogp_shard_block": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "mogp_shard_alpha": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), c_i64p]),
    "mogp_shard_finish": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, c_i64p]),
    "mogp_dev_copy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
    "mogp_set_profiling": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "mogp_stage_ms": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_i64p, c_dp]),
    "mogp_model_fetch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_dp]),
    "mogp_model_schedule": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "mogp_model_flow_replay": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "mogp_model_flow_diag": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint)]),
    "mogp_flow_plan": (ctypes.c_int, [ctypes.c_int, c_i64p, ctypes.c_int64, c_i64p]),
    "mogp_flow_plan_rhs": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_i64p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    "mogp_flow_trace": (ctypes.c_int, [ctypes.c_void_p, c_i64p, ctypes.c_int64, c_i64p]),
    "mogp_snelson_eval": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp, c_dp, ctypes.c_double, c_dp, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_double), c_dp, c_dp, c_dp, ctypes.POINTER(ctypes.c_double), c_dp,
                                         ctypes.POINTER(ctypes.c_double), c_i64p]),
    "mogp_snelson_predict": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp, c_dp, ctypes.c_double, c_dp, c_dp, ctypes.c_int64, c_dp,
                                            c_dp, c_dp, c_i64p]),
    "mogp_snelson_eval_sharded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp, c_dp, ctypes.c_double, c_dp, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_double), c_dp, c_dp, c_dp, ctypes.POINTER(ctypes.c_double), c_dp,
                                         ctypes.POINTER(ctypes.c_double), c_i64p]),
    "mogp_snelson_predict_sharded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp, c_dp, ctypes.c_double, c_dp, c_dp, ctypes.c_int64, c_dp,
                                            c_dp, c_dp, c_i64p]),
    "mogp_svgp_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp, c_dp, c_dp, ctypes.c_double, c_dp, ctypes.c_int, ctypes.c_int64,
                                         c_dp, c_dp, c_dp, c_dp, ctypes.POINTER(ctypes.c_double), c_i64p]),
    "mogp_svgp_backward": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp, ctypes.POINTER(ctypes.c_double), c_dp, c_dp]),
    "mogp_svgp_backward_sharded": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp, ctypes.POINTER(ctypes.c_double), c_dp, c_dp]),
    "mogp_oa_forward": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, c_dp, ctypes.POINTER(ctypes.c_double), c_i64p]),
    "mogp_oa_backward": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "mogp_oa_predict": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, ctypes.c_int64, c_dp, ctypes.c_int, c_dp, c_dp, c_i64p]),
    "mogp_sparse_predict_cov": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_dp]),
    "mogp_mosm_terms": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, ctypes.c_double, ctypes.c_double, c_dp]),
    "mogp_mosm_terms_backward": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, ctypes.c_double,
                                                ctypes.c_double, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
}


class MogpError(RuntimeError):
    def __init__(self, code, message, info=0):
        super().__init__(message)
        self.code = code
        self.info = info


def lib():
    """Load libmogp_hip.so (built in-tree by __graft_entry__.build()).  Raises if it is absent."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    "libmogp_hip.so not found at %s: build it with `python __graft_entry__.py` "
                    "(hipcc --offload-arch=gfx950).  mogptk_amd has no CPU fallback." % LIB_PATH)
            l = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(l, name)
                fn.restype = res
                fn.argtypes = args
            _lib = l
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def check(code, info=0):
    if code != MOGP_OK:
        msg = lib().mogp_last_error()
        raise MogpError(code, (msg.decode() if msg else "mogp error %d" % code), info)


def context(device=0):
    """One C-ABI context per device ordinal, created on first use."""
    l = lib()
    with _lock:
        if device not in _ctx:
            h = ctypes.c_void_p()
            check(l.mogp_ctx_create(int(device), ctypes.byref(h)))
            _ctx[device] = h
        return _ctx[device]


def comm_selftest(device=0):
    """-> (ranks_seen, rank_sum) of one library-issued all-reduce over the context's communicator (mogp_comm_selftest)"""
    seen, rsum = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().mogp_comm_selftest(context(device), ctypes.byref(seen), ctypes.byref(rsum)))
    return seen.value, rsum.value


def device_name(device=0):
    buf = ctypes.create_string_buffer(256)
    check(lib().mogp_ctx_device_name(context(device), buf, 256))
    return buf.value.decode()


def gram(device, C, D, table, X1, X2=None):
    """K(X1[,X2]) through mogp_gram."""
    table = _f64(table)
    X1 = _f64(X1)
    X2 = _f64(X2)
    T = table.shape[2]
    M1 = X1.shape[0]
    M2 = M1 if X2 is None else X2.shape[0]
    out = np.empty((M1, M2), dtype=np.float64)
    check(lib().mogp_gram_ex(context(device), C, D, T, int(table.shape[3]), _dp(table), M1, _dp(X1),
                             0 if X2 is None else M2, _dp(X2), _dp(out)))
    return out


def shutdown():
    """Release every model and context of this process while the HIP runtime is still up.  Registered with atexit: left to the interpreter's
    own teardown, the handles' destructors ran after the runtime's (harmless on its own, a crash inside rocprofv3's exit hooks under the
    profiler).  Calling it earlier is allowed; the next use of the library makes a new context."""
    with _lock:
        for h in list(_live):
            try:
                h.close()
            except Exception:
                pass
        if _lib is not None:
            for c in list(_ctx.values()):
                try:
                    _lib.mogp_ctx_destroy(c)
                except Exception:
                    pass
        _ctx.clear()


atexit.register(shutdown)


class ExactHandle:
    """Owner of one mogp_model (device workspaces for a fixed training set)."""

    def __init__(self, device, X, y, C):
        X = _f64(X)
        y = _f64(np.asarray(y).reshape(-1))
        self.N = X.shape[0]
        self.D = X.shape[1] - 1
        self.C = C
        self.device = device
        h = ctypes.c_void_p()
        self._h = ctypes.c_void_p()             # stays null if creation fails: close() / __del__ then have nothing to destroy
        check(lib().mogp_model_create(context(device), self.N, self.D, C, _dp(X), _dp(y), ctypes.byref(h)))
        self._h = h
        _live.add(self)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value is not None:
            lib().mogp_model_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_y(self, y):
        y = _f64(np.asarray(y).reshape(-1))
        check(lib().mogp_model_set_y(self._h, _dp(y)))

    def set_terms(self, table):
        table = _f64(table)
        assert table.shape[0] == self.C and table.shape[1] == self.C and table.shape[3] in (2 + 3 * self.D, 2 + 5 * self.D)
        self.T = table.shape[2]
        self.W = int(table.shape[3])                 # 2 + 3 D, or 2 + 5 D for terms with an envelope (MOHSM)
        check(lib().mogp_model_set_terms_ex(self._h, self.T, self.W, _dp(table)))

    def set_point_diag(self, kdiag):
        """K_diag per training point for kernels whose diagonal is not constant per channel (None: back to the table's constant)"""
        check(lib().mogp_model_set_point_diag(self._h, _dp(_f64(kdiag))))

    def eval(self, noise_var, jitter, grad=True, data_var=None):
        """-> dict(lml, moments[P,T,W], diagG[C], trG, jitter_abs)"""
        noise_var = _f64(noise_var)
        data_var = _f64(data_var)
        from .gpr.config import config as _cfg
        comm = getattr(_cfg, "comm", None)
        if grad and comm is not None and (comm.world > 1 or comm.force):
            if getattr(comm, "native", False):
                return self.eval_sharded(noise_var, jitter, data_var)
            from . import dist as _dist
            return _dist.sharded_eval(self, comm, noise_var, jitter, data_var)
        C, T, W = self.C, self.T, self.W
        lml = ctypes.c_double()
        trG = ctypes.c_double()
        jit = ctypes.c_double()
        info = ctypes.c_int64(0)
        moments = np.zeros((C * (C + 1) // 2, T, W)) if grad else None
        diagG = np.zeros(C) if grad else None
        code = lib().mogp_exact_eval(self._h, _dp(noise_var), _dp(data_var), float(jitter),
                                     MOGP_EVAL_GRAD if grad else 0, ctypes.byref(lml), _dp(moments), _dp(diagG),
                                     ctypes.byref(trG), ctypes.byref(jit), ctypes.byref(info))
        check(code, info.value)
        return dict(lml=lml.value, moments=moments, diagG=diagG, trG=trG.value, jitter_abs=jit.value)

    def eval_sharded(self, noise_var, jitter, data_var=None):
        """mogp_exact_eval_sharded: the evaluation spread over the ranks of the context's communicator (collectives inside the library)"""
        C, T, W = self.C, self.T, self.W
        lml, trG, jit, info = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64(0)
        moments, diagG = np.zeros((C * (C + 1) // 2, T, W)), np.zeros(C)
        code = lib().mogp_exact_eval_sharded(self._h, _dp(_f64(noise_var)), _dp(_f64(data_var)), float(jitter), ctypes.byref(lml), _dp(moments),
                                             _dp(diagG), ctypes.byref(trG), ctypes.byref(jit), ctypes.byref(info))
        check(code, info.value)
        return dict(lml=lml.value, moments=moments, diagG=diagG, trG=trG.value, jitter_abs=jit.value)

    def predict(self, noise_var, jitter, kss_diag, Xs, full=False, data_var=None):
        noise_var = _f64(noise_var)
        kss_diag = _f64(kss_diag)
        data_var = _f64(data_var)
        Xs = _f64(Xs)
        S = Xs.shape[0]
        from .gpr.config import config as _cfg
        comm = getattr(_cfg, "comm", None)
        if not full and comm is not None and getattr(comm, "native", False) and (comm.world > 1 or comm.force):
            mu, var, info = np.empty(S), np.empty(S), ctypes.c_int64(0)
            code = lib().mogp_exact_predict_sharded(self._h, _dp(noise_var), _dp(data_var), float(jitter), _dp(kss_diag), S, _dp(Xs),
                                                    _dp(mu), _dp(var), ctypes.byref(info))
            check(code, info.value)
            return mu.reshape(-1, 1), var.reshape(-1, 1)
        mu = np.empty(S)
        var = np.empty((S, S) if full else S)
        info = ctypes.c_int64(0)
        code = lib().mogp_exact_predict(self._h, _dp(noise_var), _dp(data_var), float(jitter), _dp(kss_diag), S, _dp(Xs),
                                        1 if full else 0, _dp(mu), _dp(var), ctypes.byref(info))
        check(code, info.value)
        return mu.reshape(-1, 1), (var if full else var.reshape(-1, 1))

    # -- sharded evaluation stages (mogp_shard_*): buffers are raw device pointers, counts are in doubles ------------------
    def shard_begin(self, rank, world, noise_var, jitter, data_var=None):
        check(lib().mogp_shard_config(self._h, int(rank), int(world)))
        jit, nblocks = ctypes.c_double(), ctypes.c_int()
        check(lib().mogp_shard_begin(self._h, _dp(_f64(noise_var)), _dp(_f64(data_var)), float(jitter), ctypes.byref(jit), ctypes.byref(nblocks)))
        return jit.value, nblocks.value

    def shard_pack(self, kb):
        send, recv, count = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        check(lib().mogp_shard_pack(self._h, kb, ctypes.byref(send), ctypes.byref(recv), ctypes.byref(count)))
        return send.value, recv.value, count.value

    def shard_unpack(self, kb):
        check(lib().mogp_shard_unpack(self._h, kb))

    def shard_block(self, kb):
        check(lib().mogp_shard_block(self._h, kb))

    def shard_alpha(self):
        buf, count = ctypes.c_void_p(), ctypes.c_int64()
        check(lib().mogp_shard_alpha(self._h, ctypes.byref(buf), ctypes.byref(count)))
        return buf.value, count.value

    def shard_finish(self):
        C, T, W = self.C, self.T, self.W
        lml, info = ctypes.c_double(), ctypes.c_int64(0)
        moments, diagG = np.zeros((C * (C + 1) // 2, T, W)), np.zeros(C)
        check(lib().mogp_shard_finish(self._h, ctypes.byref(lml), _dp(moments), _dp(diagG), ctypes.byref(info)), info.value)
        return lml.value, moments, diagG

    def mem_get(self, ptr, count):
        h = np.empty(int(count), dtype=np.float64)
        check(lib().mogp_dev_copy(h.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), 8 * int(count), 0))
        return h

    def mem_put(self, ptr, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        check(lib().mogp_dev_copy(ctypes.c_void_p(ptr), arr.ctypes.data_as(ctypes.c_void_p), 8 * arr.size, 1))

    def titsias_fetch(self, which, M):
        """intermediates of the last Titsias gradient evaluation in the device's channel-sorted order (mogp_titsias_fetch): 0 dELBO/dKuu (M x M),
        1 dELBO/dKuf without its rank-one part (M x N), 2 beta (M), 3 r (N), 4 v = L^-1 Kuf (M x N), 5 L (M x M), 6 Qs, 7 Pq = Qs^-1, 8 t1 = Pq v y"""
        shape = {0: (M, M), 1: (M, self.N), 2: (M,), 3: (self.N,), 4: (M, self.N), 5: (M, M), 6: (M, M), 7: (M, M), 8: (M,)}[which]
        out = np.empty(shape)
        check(lib().mogp_titsias_fetch(self._h, int(which), int(M), _dp(out)))
        return out

    def titsias_eval(self, Z, sigma, jitter, kff_diag, grad=True, sharded=False):
        """Titsias bound (+ gradient outputs) through mogp_titsias_eval, or -- this handle holding one shard of the data -- through
        mogp_titsias_eval_sharded (the same result on every rank: the full model's)"""
        Z = _f64(Z)
        kff_diag = _f64(kff_diag)
        M = Z.shape[0]
        C, T, W, D = self.C, self.T, self.W, self.D      # rows of the width set_terms took (2 + 3 D, or 2 + 5 D with an envelope)
        self._check_diag("kff_diag", kff_diag, self.N)
        elbo, trGA, dsig, jit = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        info = ctypes.c_int64(0)
        mom_uu = np.zeros((C * (C + 1) // 2, T, W)) if grad else None
        mom_uf = np.zeros((C * C, T, W)) if grad else None
        gZ = np.zeros((M, D)) if grad else None
        fn = lib().mogp_titsias_eval_sharded if sharded else lib().mogp_titsias_eval
        code = fn(self._h, M, _dp(Z), float(sigma), float(jitter), _dp(kff_diag),
                  MOGP_EVAL_GRAD if grad else 0, ctypes.byref(elbo), _dp(mom_uu), _dp(mom_uf), _dp(gZ),
                  ctypes.byref(trGA), ctypes.byref(dsig), ctypes.byref(jit), ctypes.byref(info))
        check(code, info.value)
        return dict(elbo=elbo.value, mom_uu=mom_uu, mom_uf=mom_uf, gZ=gZ, trGA=trGA.value, dsigma=dsig.value,
                    jitter_abs=jit.value)

    def titsias_predict(self, Z, sigma, jitter, Xs, kss_diag, sharded=False):
        Z, Xs, kss_diag = _f64(Z), _f64(Xs), _f64(kss_diag)
        S = Xs.shape[0]
        self._check_diag("kss_diag", kss_diag, S)
        mu, var = np.empty(S), np.empty(S)
        info = ctypes.c_int64(0)
        fn = lib().mogp_titsias_predict_sharded if sharded else lib().mogp_titsias_predict
        code = fn(self._h, Z.shape[0], _dp(Z), float(sigma), float(jitter), _dp(kss_diag), S, _dp(Xs), _dp(mu), _dp(var), ctypes.byref(info))
        check(code, info.value)
        return mu.reshape(-1, 1), var.reshape(-1, 1)

    def _check_diag(self, name, diag, npoints):
        """K_diag arrays of the sparse / variational entry points: one value per CHANNEL, or -- with enveloped terms (rows of width 2 + 5 D),
        whose diagonal follows the points -- one per point; the C side indexes them accordingly, so the length is checked here"""
        want = npoints if self.W > 2 + 3 * self.D else self.C
        if diag is None or diag.size != want:
            raise ValueError("%s must have %d entries (%s), got %s" % (name, want, "per point: the terms carry an envelope" if want != self.C else "per channel",
                                                                       None if diag is None else diag.size))

    def snelson_eval(self, Z, noise_var, jitter, kff_diag, grad=True, sharded=False):
        """Snelson (FITC) marginal likelihood (+ gradient outputs) through mogp_snelson_eval"""
        Z, noise_var, kff_diag = _f64(Z), _f64(noise_var), _f64(kff_diag)
        M = Z.shape[0]
        C, T, W, D = self.C, self.T, self.W, self.D       # W = 2 + 5 D with enveloped terms: kff_diag per point in, dp/dKff per point out
        env = W > 2 + 3 * D
        if kff_diag.size != (self.N if env else C):
            raise ValueError("kff_diag must have %d entries" % (self.N if env else C))
        lml, trGA, jit = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        info = ctypes.c_int64(0)
        mom_uu = np.zeros((C * (C + 1) // 2, T, W)) if grad else None
        mom_uf = np.zeros((C * C, T, W)) if grad else None
        gZ = np.zeros((M, D)) if grad else None
        hsum = np.zeros(self.N if env else C) if grad else None
        fn = lib().mogp_snelson_eval_sharded if sharded else lib().mogp_snelson_eval
        code = fn(self._h, M, _dp(Z), _dp(noise_var), float(jitter), _dp(kff_diag), MOGP_EVAL_GRAD if grad else 0,
                  ctypes.byref(lml), _dp(mom_uu), _dp(mom_uf), _dp(gZ), ctypes.byref(trGA), _dp(hsum), ctypes.byref(jit), ctypes.byref(info))
        check(code, info.value)
        return dict(lml=lml.value, mom_uu=mom_uu, mom_uf=mom_uf, gZ=gZ, trGA=trGA.value, hsum=hsum, jitter_abs=jit.value)

    def snelson_predict(self, Z, noise_var, jitter, Xs, kff_diag, kss_diag, sharded=False):
        Z, Xs, noise_var, kff_diag, kss_diag = _f64(Z), _f64(Xs), _f64(noise_var), _f64(kff_diag), _f64(kss_diag)
        S = Xs.shape[0]
        self._check_diag("kff_diag", kff_diag, self.N)
        self._check_diag("kss_diag", kss_diag, S)
        mu, var = np.empty(S), np.empty(S)
        info = ctypes.c_int64(0)
        fn = lib().mogp_snelson_predict_sharded if sharded else lib().mogp_snelson_predict
        code = fn(self._h, Z.shape[0], _dp(Z), _dp(noise_var), float(jitter), _dp(kff_diag), _dp(kss_diag), S, _dp(Xs),
                  _dp(mu), _dp(var), ctypes.byref(info))
        check(code, info.value)
        return mu.reshape(-1, 1), var.reshape(-1, 1)

    def svgp_forward(self, Z, q_mu, q_sqrt, jitter, kff_diag, Xs=None, kss_diag=None, dense=False):
        """mean / variance of q(f) of the Hensman models at the training inputs (Xs None) or at Xs, through mogp_svgp_forward"""
        Z, q_mu, q_sqrt, kff_diag = _f64(Z), _f64(np.reshape(q_mu, -1)), _f64(q_sqrt), _f64(kff_diag)
        M = Z.shape[0]
        S = 0 if Xs is None else int(np.shape(Xs)[0])
        Xs = None if Xs is None else _f64(Xs)
        kss = None if kss_diag is None else _f64(kss_diag)
        n = self.N if Xs is None else S
        if Xs is None:
            self._check_diag("kff_diag", kff_diag, self.N)
        else:
            self._check_diag("kss_diag", kss, S)
        mu, var = np.empty(n), np.empty(n)
        jit = ctypes.c_double()
        info = ctypes.c_int64(0)
        code = lib().mogp_svgp_forward(self._h, M, _dp(Z), _dp(q_mu), _dp(q_sqrt), float(jitter), _dp(kff_diag), 1 if dense else 0, S,
                                       _dp(Xs), _dp(kss), _dp(mu), _dp(var), ctypes.byref(jit), ctypes.byref(info))
        check(code, info.value)
        self._svgp_M = M
        return dict(mu=mu, var=var, jitter_abs=jit.value)

    def svgp_backward(self, e, f, sharded=False):
        e, f = _f64(np.reshape(e, -1)), _f64(np.reshape(f, -1))
        C, T, W, D, M = self.C, self.T, self.W, self.D, self._svgp_M       # W = 2 + 5 D with enveloped terms
        mom_uu, mom_uf = np.zeros((C * (C + 1) // 2, T, W)), np.zeros((C * C, T, W))
        gZ, g_qmu, g_S = np.zeros((M, D)), np.zeros(M), np.zeros((M, M))
        trGA = ctypes.c_double()
        fn = lib().mogp_svgp_backward_sharded if sharded else lib().mogp_svgp_backward
        check(fn(self._h, _dp(e), _dp(f), _dp(mom_uu), _dp(mom_uf), _dp(gZ), ctypes.byref(trGA), _dp(g_qmu), _dp(g_S)))
        return dict(mom_uu=mom_uu, mom_uf=mom_uf, gZ=gZ, trGA=trGA.value, g_qmu=g_qmu, g_qsqrt=g_S)

    def oa_forward(self, q_nu, q_lambda):
        """mean / variance of q(f) of the Opper-Archambeau model at the training inputs and its `kl` term, through mogp_oa_forward"""
        q_nu, q_lambda = _f64(np.reshape(q_nu, -1)), _f64(np.reshape(q_lambda, -1))
        mu, var = np.empty(self.N), np.empty(self.N)
        kl = ctypes.c_double()
        info = ctypes.c_int64(0)
        check(lib().mogp_oa_forward(self._h, _dp(q_nu), _dp(q_lambda), _dp(mu), _dp(var), ctypes.byref(kl), ctypes.byref(info)), info.value)
        return dict(mu=mu, var=var, kl=kl.value)

    def oa_backward(self, e, f):
        e, f = _f64(np.reshape(e, -1)), _f64(np.reshape(f, -1))
        mom = np.zeros((self.C * (self.C + 1) // 2, self.T, 2 + 3 * self.D))
        g_nu, g_lambda = np.zeros(self.N), np.zeros(self.N)
        check(lib().mogp_oa_backward(self._h, _dp(e), _dp(f), _dp(mom), _dp(g_nu), _dp(g_lambda)))
        return dict(mom=mom, g_nu=g_nu, g_lambda=g_lambda)

    def oa_predict(self, q_nu, q_lambda, kss_diag, Xs, full=False):
        q_nu, q_lambda, kss_diag, Xs = _f64(np.reshape(q_nu, -1)), _f64(np.reshape(q_lambda, -1)), _f64(kss_diag), _f64(Xs)
        S = Xs.shape[0]
        mu, var = np.empty(S), np.empty((S, S) if full else S)
        info = ctypes.c_int64(0)
        check(lib().mogp_oa_predict(self._h, _dp(q_nu), _dp(q_lambda), _dp(kss_diag), S, _dp(Xs), 1 if full else 0, _dp(mu), _dp(var),
                                    ctypes.byref(info)), info.value)
        return mu.reshape(-1, 1), (var if full else var.reshape(-1, 1))

    def sparse_predict_cov(self, S):
        """full S x S covariance of the last sparse prediction on this handle (Titsias / Hensman), through mogp_sparse_predict_cov"""
        cov = np.empty((int(S), int(S)))
        check(lib().mogp_sparse_predict_cov(self._h, int(S), _dp(cov)))
        return cov

    def set_profiling(self, on=True):
        check(lib().mogp_set_profiling(self._h, 1 if on else 0))

    def stage_ms(self):
        ms = np.zeros(ST_COUNT)
        n = ctypes.c_int64(0)
        fl = ctypes.c_double(0)
        check(lib().mogp_stage_ms(self._h, _dp(ms), ctypes.byref(n), ctypes.byref(fl)))
        return ms, n.value, fl.value

    def shard_stage_ms(self):
        """(exchange on the critical stream, serial, next_cols, bulk, exchange on the communication stream, the critical stream's wait for it) ms of the
        last profiled sharded evaluation"""
        ms = np.zeros(6)
        check(lib().mogp_shard_stage_ms(self._h, _dp(ms)))
        return ms

    def schedule(self):
        """-> dict(dataflow, chain_kernel, dataflow_fell_back, chain_fell_back, dataflow_timeouts): how the last gradient evaluation was scheduled (mogp_model_schedule)"""
        f = ctypes.c_int(0)
        check(lib().mogp_model_schedule(self._h, ctypes.byref(f)))
        return dict(dataflow=bool(f.value & 1), chain_kernel=bool(f.value & 2), dataflow_fell_back=bool(f.value & 4), chain_fell_back=bool(f.value & 8),
                    dataflow_timeouts=(f.value >> 8) & 0xffff)

    def flow_diag(self):
        """-> dict of the dataflow schedule's deep-look counters over this model's life (mogp_model_flow_diag)"""
        out = (ctypes.c_uint * 8)()
        check(lib().mogp_model_flow_diag(self._h, out))
        return dict(deep_looks=out[0], stale_heads=out[1], stale_counters=out[2], deep_polls=out[3], stale_polls=out[4], last=(out[5], out[6], out[7] & 0xffff, out[7] >> 16))

    def flow_replay(self, on):
        """measurement mode (mogp_model_flow_replay): the next gradient evaluations run the dataflow kernel alone on the replay plan"""
        check(lib().mogp_model_flow_replay(self._h, 1 if on else 0))

    accurate_mode = False          # what the last set_accurate asked for (a NEW handle starts in the fast mode, as the device object does)

    def set_accurate(self, on):
        """gradient evaluations in the backward-stable form (mogp_model_set_accurate): slower, for ill-conditioned Kj"""
        check(lib().mogp_model_set_accurate(self._h, 1 if on else 0))
        self.accurate_mode = bool(on)

    def condition_estimate(self):
        """(largest / smallest diagonal entry of L)^2 of the last factorisation: a lower estimate of cond(Kj); nan when it was not reported"""
        lo, hi = np.zeros(1), np.zeros(1)
        check(lib().mogp_model_pivot_range(self._h, _dp(lo), _dp(hi)))
        return float((hi[0] / lo[0]) ** 2) if lo[0] > 0.0 else float("nan")

    def work_bytes(self):
        """(bytes of the N x N work matrix this handle has physical memory for, bytes of the whole matrix): a rank of a sharded evaluation holds its own
        tile rows only (mogp_model_work_bytes)"""
        b, w = ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib().mogp_model_work_bytes(self._h, ctypes.byref(b), ctypes.byref(w)))
        return int(b.value), int(w.value)

    def inverse_fraction(self):
        """fraction of the lower tiles of Kj^-1 the last gradient evaluation formed (1.0 = all; see include/mogp_hip.h)"""
        f = np.zeros(1)
        check(lib().mogp_model_inverse_fraction(self._h, _dp(f)))
        return float(f[0])

    def fetch(self, which):
        out = np.empty(self.N if which == 2 else (self.N, self.N))
        check(lib().mogp_model_fetch(self._h, which, _dp(out)))
        return out
