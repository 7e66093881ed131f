// comm.hip -- collectives of the sharded evaluation, issued BY THE LIBRARY on its own HIP streams.
//
// Two back ends behind the same two calls (all-gather, all-reduce-sum of doubles in device memory):
//   RCCL      ncclAllGather / ncclAllReduce enqueued on the library's critical stream: no host round trip between a pivot block's pack,
//             its exchange, its unpack and its arithmetic, and the previous block's trailing update keeps running on the bulk stream
//             underneath.  librccl is opened at run time (dlopen) -- a single-GPU process never loads it -- and the communicator is
//             bootstrapped from a 128-byte unique id the caller distributes by whatever means it has (mogp_comm_unique_id on rank 0,
//             mogp_comm_init_rccl everywhere; mogptk_amd/dist.py uses a torch.distributed broadcast, MPI or a file would do).
//   external  two caller-supplied callbacks on device pointers (the stream is drained before each call).  This is how the N > 1 path
//             is exercised where RCCL cannot run: several ranks sharing ONE GPU in the test suite (RCCL refuses duplicate devices),
//             with gloo moving the data through the host.
#include "mogp_model.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace mogp;

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return fail(MOGP_ENODEVICE, std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "not found"));
#define SYM(field, name)                                                                            \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                        \
    if (!g_rccl.field) return fail(MOGP_ENODEVICE, std::string("librccl lacks the symbol ") + name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = h;
    return 0;
}

int rccl_fail(ncclResult_t r, const char* what) {
    return fail(MOGP_EHIP, std::string("RCCL error '") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?") + "' in " + what);
}

}  // namespace

namespace mogp {

int comm_allgather(mogp_ctx* ctx, const double* send, double* recv, int64_t count, hipStream_t st) {
    mogp_comm& c = ctx->comm;
    if (c.kind == MOGP_COMM_RCCL) {
        ncclResult_t r = g_rccl.AllGather(send, recv, (size_t)count, ncclDouble, reinterpret_cast<ncclComm_t>(c.nccl), st);
        if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather");
        return 0;
    }
    if (c.kind == MOGP_COMM_EXTERNAL) {
        HIP_TRY(hipStreamSynchronize(st));
        if (c.allgather(c.user, send, recv, count)) return fail(MOGP_EHIP, "the external all-gather callback failed");
        return 0;
    }
    if (c.n == 1 || c.kind == MOGP_COMM_NONE) {             // a group of one: the gather is a copy
        HIP_TRY(hipMemcpyAsync(recv, send, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    return fail(MOGP_EINVAL, "no communicator on this context");
}

int comm_allreduce(mogp_ctx* ctx, double* buf, int64_t count, hipStream_t st) {
    mogp_comm& c = ctx->comm;
    if (c.kind == MOGP_COMM_RCCL) {
        ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, reinterpret_cast<ncclComm_t>(c.nccl), st);
        if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce");
        return 0;
    }
    if (c.kind == MOGP_COMM_EXTERNAL) {
        HIP_TRY(hipStreamSynchronize(st));
        if (c.allreduce(c.user, buf, count)) return fail(MOGP_EHIP, "the external all-reduce callback failed");
        return 0;
    }
    return 0;                                               // a group of one
}

}  // namespace mogp

extern "C" {

int mogp_comm_unique_id(void* id128) {
    if (!id128) return fail(MOGP_EINVAL, "mogp_comm_unique_id: null argument");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    static_assert(sizeof(ncclUniqueId) == MOGP_COMM_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof(id));
    return MOGP_OK;
}

int mogp_comm_init_rccl(mogp_ctx* ctx, const void* id128, int rank, int nranks) {
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(MOGP_EINVAL, "mogp_comm_init_rccl: bad argument");
    int rc = rccl_load();
    if (rc) return rc;
    if ((rc = use_device(ctx))) return rc;
    mogp_comm_destroy(ctx);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    ctx->comm.kind = MOGP_COMM_RCCL; ctx->comm.rank = rank; ctx->comm.n = nranks; ctx->comm.nccl = comm;
    return MOGP_OK;
}

int mogp_comm_init_external(mogp_ctx* ctx, int rank, int nranks, mogp_allgather_cb allgather, mogp_allreduce_cb allreduce, void* user) {
    if (!ctx || !allgather || !allreduce || nranks < 1 || rank < 0 || rank >= nranks) return fail(MOGP_EINVAL, "mogp_comm_init_external: bad argument");
    mogp_comm_destroy(ctx);
    ctx->comm.kind = MOGP_COMM_EXTERNAL; ctx->comm.rank = rank; ctx->comm.n = nranks;
    ctx->comm.allgather = allgather; ctx->comm.allreduce = allreduce; ctx->comm.user = user;
    return MOGP_OK;
}

int mogp_comm_destroy(mogp_ctx* ctx) {
    if (!ctx) return MOGP_OK;
    if (ctx->comm.kind == MOGP_COMM_RCCL && ctx->comm.nccl && g_rccl.CommDestroy) {
        hipError_t e = hipSetDevice(ctx->device); (void)e;
        g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(ctx->comm.nccl));
    }
    ctx->comm = mogp_comm();
    return MOGP_OK;
}

int mogp_comm_info(mogp_ctx* ctx, int* kind, int* rank, int* nranks) {
    if (!ctx) return fail(MOGP_EINVAL, "mogp_comm_info: ctx is null");
    if (kind) *kind = ctx->comm.kind;
    if (rank) *rank = ctx->comm.rank;
    if (nranks) *nranks = ctx->comm.n;
    return MOGP_OK;
}

}  // extern "C"
