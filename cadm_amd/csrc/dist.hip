// RCCL glue for candidate-sharded planning: one ncclAllGather of per-candidate returns per CEM iteration
// over xGMI.  librccl is dlopen'ed lazily (it resolves to the copy PyTorch already loaded when there is one),
// so single-GPU use has no RCCL dependency.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {
struct IdBlob { char b[128]; };     // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
typedef int (*fn_get_id)(void*);
typedef int (*fn_init_rank2)(void**, int, IdBlob, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_comm_int)(void*, int*);

struct Rccl {
    void* h = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank2 init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr;
    fn_errstr errstr = nullptr;
    fn_comm_int count = nullptr, user_rank = nullptr;
} g_rccl;

int load_rccl() {
    if (g_rccl.h) return CADM_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { cadm_set_error("cannot dlopen librccl: %s", dlerror()); return CADM_EINVAL; }
    g_rccl.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank2)dlsym(h, "ncclCommInitRank");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    g_rccl.count = (fn_comm_int)dlsym(h, "ncclCommCount");
    g_rccl.user_rank = (fn_comm_int)dlsym(h, "ncclCommUserRank");
    if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allgather) {
        cadm_set_error("librccl lacks the expected nccl* symbols");
        return CADM_EINVAL;
    }
    g_rccl.h = h;
    return CADM_OK;
}

int check_nccl(int rc, const char* what) {
    if (rc == 0) return CADM_OK;
    cadm_set_error("%s failed: %s (ncclResult %d)", what, g_rccl.errstr ? g_rccl.errstr(rc) : "?", rc);
    return CADM_EHIP;
}
}  // namespace

static int dist_flag_alloc(cadm_ctx* ctx) {
    if (ctx->dist_flag) return CADM_OK;
    CADM_CHECK_HIP(hipMalloc(&ctx->dist_flag, sizeof(unsigned)));
    CADM_CHECK_HIP(hipMemset(ctx->dist_flag, 0, sizeof(unsigned)));
    return CADM_OK;
}

extern "C" int cadm_dist_unique_id(char out_id[128]) {
    CADM_REQUIRE(out_id, "cadm_dist_unique_id: null argument");
    int rc = load_rccl();
    if (rc) return rc;
    return check_nccl(g_rccl.get_id(out_id), "ncclGetUniqueId");
}

extern "C" int cadm_dist_init(cadm_ctx* ctx, const char id[128], int nranks, int rank) {
    CADM_REQUIRE(ctx && id && nranks >= 1 && rank >= 0 && rank < nranks, "cadm_dist_init: bad arguments");
    CADM_REQUIRE(!cadm_sharded(ctx), "cadm_dist_init: communicator already initialised");
    int rc = load_rccl();
    if (rc) return rc;
    CADM_CHECK_HIP(hipSetDevice(ctx->device));
    IdBlob blob;
    memcpy(blob.b, id, 128);
    void* comm = nullptr;
    if ((rc = check_nccl(g_rccl.init_rank(&comm, nranks, blob, rank), "ncclCommInitRank"))) return rc;
    // the communicator RCCL built must be the one asked for (the id travels by value through ncclCommInitRank)
    if (g_rccl.count && g_rccl.user_rank) {
        int cnt = -1, ur = -1;
        rc = check_nccl(g_rccl.count(comm, &cnt), "ncclCommCount");
        if (!rc) rc = check_nccl(g_rccl.user_rank(comm, &ur), "ncclCommUserRank");
        if (!rc && (cnt != nranks || ur != rank)) {
            cadm_set_error("cadm_dist_init: RCCL communicator has %d ranks / rank %d, expected %d / %d", cnt, ur, nranks, rank);
            rc = CADM_EINVAL;
        }
        if (rc) { g_rccl.destroy(comm); return rc; }
    }
    if ((rc = dist_flag_alloc(ctx))) { g_rccl.destroy(comm); return rc; }
    ctx->comm = comm;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return CADM_OK;
}

extern "C" int cadm_dist_init_external(cadm_ctx* ctx, int nranks, int rank, cadm_allgather_fn fn, void* user) {
    CADM_REQUIRE(ctx && fn && nranks >= 1 && rank >= 0 && rank < nranks, "cadm_dist_init_external: bad arguments");
    CADM_REQUIRE(!cadm_sharded(ctx), "cadm_dist_init_external: the ctx already has a communicator");
    CADM_ON_DEVICE(ctx);
    const int rc = dist_flag_alloc(ctx);
    if (rc) return rc;
    ctx->ext_allgather = fn;
    ctx->ext_user = user;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return CADM_OK;
}

extern "C" int cadm_dist_destroy(cadm_ctx* ctx) {
    if (!ctx) return CADM_OK;
    if (ctx->comm) g_rccl.destroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->ext_allgather = nullptr;
    ctx->ext_user = nullptr;
    ctx->nranks = 1;
    ctx->rank = 0;
    return CADM_OK;
}

extern "C" int cadm_dist_mismatch(cadm_ctx* ctx, int* mismatch_out, void* stream) {
    CADM_REQUIRE(ctx && mismatch_out, "cadm_dist_mismatch: null argument");
    *mismatch_out = 0;
    if (!ctx->dist_flag) return CADM_OK;
    CADM_ON_DEVICE(ctx);
    unsigned v = 0;
    CADM_CHECK_HIP(hipMemcpyAsync(&v, ctx->dist_flag, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    CADM_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (v) CADM_CHECK_HIP(hipMemsetAsync(ctx->dist_flag, 0, sizeof(unsigned), (hipStream_t)stream));
    *mismatch_out = v ? 1 : 0;
    return CADM_OK;
}

extern "C" int cadm_dist_info(cadm_ctx* ctx, int* nranks_out, int* rank_out) {
    CADM_REQUIRE(ctx && nranks_out && rank_out, "cadm_dist_info: null argument");
    *nranks_out = 1;
    *rank_out = 0;
    if (ctx->ext_allgather) { *nranks_out = ctx->nranks; *rank_out = ctx->rank; return CADM_OK; }      // host-supplied collective
    if (!ctx->comm) return CADM_OK;                         // single-GPU planner: no communicator
    CADM_REQUIRE(g_rccl.count && g_rccl.user_rank, "cadm_dist_info: librccl lacks ncclCommCount / ncclCommUserRank");
    int rc = check_nccl(g_rccl.count(ctx->comm, nranks_out), "ncclCommCount");
    if (rc) return rc;
    return check_nccl(g_rccl.user_rank(ctx->comm, rank_out), "ncclCommUserRank");
}

// [count] floats per rank -> [nranks * count] on every rank (ncclFloat32 = 7)
int cadm_dist_allgather(cadm_ctx* ctx, const float* send, float* recv, size_t count, hipStream_t s) {
    if (ctx->ext_allgather) {
        const int rc = ctx->ext_allgather(ctx->ext_user, send, recv, count, (void*)s);
        if (rc) { cadm_set_error("cadm_dist_allgather: the host-supplied all-gather failed (code %d)", rc); return CADM_EHIP; }
        return CADM_OK;
    }
    if (!ctx->comm) { cadm_set_error("cadm_dist_allgather: no communicator"); return CADM_ESTATE; }
    return check_nccl(g_rccl.allgather(send, recv, count, 7, ctx->comm, s), "ncclAllGather");
}
