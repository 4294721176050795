// comm.hip -- the path's one exchange step in the C ABI: an all-reduce of the integer counter blocks over RCCL.
//
// realizations shard embarrassingly over the GPUs of a node (one process and one mcle_ctx
// per GPU); what leaves a GPU is the 8-word mcle_counters block per parameter variation, summed over ranks once.
// The reference's only multi-process mechanism is ipyparallel (simulations/runner.py:1836-1846: one parameter
// variation per engine, results gathered through pickles); here every rank takes a slice of EVERY variation and the
// exact integer sums are reduced over xGMI.
//
// RCCL is bound at run time (dlopen) so that libmcle.so loads on hosts without it and shares the copy a host
// process has already loaded (PyTorch bundles one; the soname is the same as /opt/rocm's).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.hpp"

namespace mcle {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

static int rccl_load(const char* path) {
    if (g_rccl.handle) return MCLE_OK;
    const char* candidates[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* c : candidates) {
        if (!c || !*c) continue;
        h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        set_error("cannot load RCCL (librccl.so.1): %s", dlerror());
        return MCLE_E_STATE;
    }
#define MCLE_SYM(field, name)                                               \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
    if (!g_rccl.field) {                                                    \
        set_error("RCCL symbol %s not found", name);                        \
        return MCLE_E_STATE;                                                \
    }
    MCLE_SYM(GetUniqueId, "ncclGetUniqueId")
    MCLE_SYM(CommInitRank, "ncclCommInitRank")
    MCLE_SYM(CommDestroy, "ncclCommDestroy")
    MCLE_SYM(AllReduce, "ncclAllReduce")
    MCLE_SYM(GroupStart, "ncclGroupStart")
    MCLE_SYM(GroupEnd, "ncclGroupEnd")
    MCLE_SYM(GetErrorString, "ncclGetErrorString")
#undef MCLE_SYM
    g_rccl.handle = h;
    return MCLE_OK;
}

#define MCLE_NCCL(expr)                                                                              \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess) {                                                                     \
            set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
            return MCLE_E_STATE;                                                                     \
        }                                                                                            \
    } while (0)

// counters [n][8] <-> packed sums [6 n] (words 0..5) and maxima [2 n] (n_symbols, n_bits: per-realization
// constants, zero on a rank whose shard was empty)
__global__ void k_counters_pack(const mcle_counters* __restrict__ c, int n, unsigned long long* __restrict__ sums,
                                unsigned long long* __restrict__ maxs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(c + i);
#pragma unroll
    for (int k = 0; k < 6; ++k) sums[6 * i + k] = w[k];
    maxs[2 * i] = w[6];
    maxs[2 * i + 1] = w[7];
}
__global__ void k_counters_unpack(mcle_counters* __restrict__ c, int n, const unsigned long long* __restrict__ sums,
                                  const unsigned long long* __restrict__ maxs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long* w = reinterpret_cast<unsigned long long*>(c + i);
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = sums[6 * i + k];
    w[6] = maxs[2 * i];
    w[7] = maxs[2 * i + 1];
}

}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_comm_load(const char* rccl_path) { return rccl_load(rccl_path); }

int mcle_comm_unique_id(void* id_out, size_t bytes) {
    MCLE_REQUIRE(id_out != nullptr && bytes >= sizeof(ncclUniqueId), "unique id buffer must hold %zu bytes",
                 sizeof(ncclUniqueId));
    int rc = rccl_load(nullptr);
    if (rc) return rc;
    ncclUniqueId id;
    MCLE_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return MCLE_OK;
}

int mcle_comm_init(mcle_ctx* ctx, const void* unique_id, int rank, int world) {
    MCLE_REQUIRE(ctx != nullptr && unique_id != nullptr, "null argument");
    MCLE_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d outside a world of %d", rank, world);
    MCLE_REQUIRE(ctx->comm == nullptr, "the context already has a communicator (mcle_comm_destroy first)");
    int rc = rccl_load(nullptr);
    if (rc) return rc;
    if ((rc = ctx->bind())) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    MCLE_NCCL(g_rccl.CommInitRank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return MCLE_OK;
}

int mcle_comm_destroy(mcle_ctx* ctx) {
    MCLE_REQUIRE(ctx != nullptr, "null argument");
    if (ctx->comm) {
        MCLE_HIP(hipStreamSynchronize(ctx->stream));
        MCLE_NCCL(g_rccl.CommDestroy(static_cast<ncclComm_t>(ctx->comm)));
        ctx->comm = nullptr;
        ctx->comm_rank = 0;
        ctx->comm_world = 1;
    }
    return MCLE_OK;
}

int mcle_comm_info(mcle_ctx* ctx, int* rank, int* world) {
    MCLE_REQUIRE(ctx != nullptr, "null argument");
    if (rank) *rank = ctx->comm_rank;
    if (world) *world = ctx->comm_world;
    return MCLE_OK;
}

int mcle_counters_allreduce(mcle_ctx* ctx, mcle_counters* d_counters, int n) {
    MCLE_REQUIRE(ctx != nullptr && d_counters != nullptr && n >= 1, "bad argument");
    // no communicator: single process, nothing to exchange.  WITH a communicator of one rank the whole path still runs
    // (pack, the grouped in-place SUM / MAX all-reduces, unpack): RCCL treats it as a copy, and it is the only way a
    // one-GPU box can execute these kernels and calls (tests/test_gpu_distributed.py).
    if (ctx->comm == nullptr) return MCLE_OK;
    int rc;
    if ((rc = ctx->bind())) return rc;
    if (!ctx->d_comm_buf || ctx->comm_buf_words < (size_t)8 * n) {
        if (ctx->d_comm_buf) MCLE_HIP(hipFree(ctx->d_comm_buf));
        MCLE_HIP(hipMalloc(&ctx->d_comm_buf, (size_t)8 * n * sizeof(unsigned long long)));
        ctx->comm_buf_words = (size_t)8 * n;
    }
    unsigned long long* sums = static_cast<unsigned long long*>(ctx->d_comm_buf);
    unsigned long long* maxs = sums + (size_t)6 * n;
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_counters_pack, dim3(blocks), dim3(256), 0, ctx->stream, d_counters, n, sums, maxs);
    MCLE_LAUNCH_CHECK();
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    MCLE_NCCL(g_rccl.GroupStart());
    MCLE_NCCL(g_rccl.AllReduce(sums, sums, (size_t)6 * n, ncclUint64, ncclSum, comm, ctx->stream));
    MCLE_NCCL(g_rccl.AllReduce(maxs, maxs, (size_t)2 * n, ncclUint64, ncclMax, comm, ctx->stream));
    MCLE_NCCL(g_rccl.GroupEnd());
    hipLaunchKernelGGL(k_counters_unpack, dim3(blocks), dim3(256), 0, ctx->stream, d_counters, n, sums, maxs);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_allreduce_f64(mcle_ctx* ctx, double* d_values, size_t n) {
    MCLE_REQUIRE(ctx != nullptr && d_values != nullptr, "null argument");
    if (ctx->comm == nullptr || n == 0) return MCLE_OK;
    int rc;
    if ((rc = ctx->bind())) return rc;
    MCLE_NCCL(g_rccl.AllReduce(d_values, d_values, n, ncclFloat64, ncclSum, static_cast<ncclComm_t>(ctx->comm),
                               ctx->stream));
    return MCLE_OK;
}

}  // extern "C"
