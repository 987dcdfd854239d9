// Shared host/device definitions for libcadm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/cadm_hip.h"
#include "xdl_geo.h"

// ---------------------------------------------------------------------------------------------
// error plumbing (no exceptions cross the C ABI)
// ---------------------------------------------------------------------------------------------
#ifdef CADM_JIT_MODULE      // side modules (rollout_jit.hip) report through the pointer the ctx carries
extern void (*cadm_jit_set_error)(const char*, ...);
#define cadm_set_error(...) cadm_jit_set_error(__VA_ARGS__)
#else
void cadm_set_error(const char* fmt, ...);
#endif
// (hidden width, context width) lists compiled into the library (Makefile HIDS / CTXS)
#ifndef CADM_CTX_LIST
#define CADM_CTX_LIST 0, 10            // 0 = vanilla PE-TS, 10 = the reference default --context_out_dim (run_cadm_pets.py:135)
#endif
#ifndef CADM_HID_LIST
#define CADM_HID_LIST 200              // the reference default --hidden_size (run_cadm_pets.py:129)
#endif
// bumped whenever cadm_ctx / RolloutArgs change: a side module built against another layout is refused
#define CADM_CTX_LAYOUT_TAG 3004

#define CADM_CHECK_HIP(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            cadm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return CADM_EHIP;                                                                  \
        }                                                                                      \
    } while (0)

#define CADM_REQUIRE(cond, ...)        \
    do {                               \
        if (!(cond)) {                 \
            cadm_set_error(__VA_ARGS__); \
            return CADM_EINVAL;        \
        }                              \
    } while (0)

// ---------------------------------------------------------------------------------------------
// env tables
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int env_D(int k) { return k == 0 ? 18 : k == 1 ? 28 : k == 2 ? 45 : k == 3 ? 4 : 3; }
__host__ __device__ constexpr int env_A(int k) { return k == 0 ? 6 : k == 1 ? 8 : k == 2 ? 17 : k == 3 ? 2 : 1; }
__host__ __device__ constexpr int env_P(int k) { return k == 0 ? 18 : k == 1 ? 27 : k == 2 ? 45 : k == 3 ? 4 : 3; }

// RNG stream tags (oracle/philox.py)
#define CADM_STREAM_EPS 1u
#define CADM_STREAM_ACT 2u
#define CADM_STREAM_UNI 3u

// ---------------------------------------------------------------------------------------------
// planner weight-stream geometry (see DESIGN.md "weight streams")
//   A dense layer with K inputs and a set of 16-wide output tiles is evaluated as
//   OUT^T[unit][row] = sum_k W^T[unit][k] * IN^T[k][row] with v_mfma_f32_16x16x4_f32:
//   the weights are the A operand, 16 data rows are the B/D columns.  K is consumed in
//   "chunks" of 4 k-steps (16 input features).  A workgroup has 4 waves; wave w owns
//   NFO = ntiles/4 "full" output tiles (tiles w*NFO .. w*NFO+NFO-1) over all of K, and every one of
//   the NSO = ntiles%4 "split" tiles (tiles 4*NFO ..) for k-step r == w of every chunk.
//   Per (member, layer, wave) the stream is: for each chunk: NFO blocks of float4[64 lanes]
//   followed by NSO blocks of float[64 lanes], in exactly the order the wave consumes them.
// ---------------------------------------------------------------------------------------------
struct LayerGeo {
    int K;        // input features
    int nch;      // chunks = ceil(K/16)
    int ntiles;   // output tiles
    int nfo, nso; // full tiles per wave / split tiles
    int head;     // 0: hidden-type outputs (HID units), 1: (mu,logvar) head tiles of 8 dims
    int nout;     // HID or D
    int csplit;   // 1: all tiles K-split by CHUNK (wave w owns chunks w, w+4, ..; float4 blocks), nfo == 0
    int nslots() const { return csplit ? (nch + 3) / 4 : nch; }                 // chunk slots per wave
    size_t slot_floats() const { return csplit ? (size_t)nso * 256 : (size_t)(nfo * 4 + nso) * 64; }
    size_t wave_floats() const { return slot_floats() * nslots(); }
    size_t layer_floats() const { return wave_floats() * 4; }
    size_t bias_floats() const { return (size_t)ntiles * 256; }
};

struct DenseRef {  // registered master weights (caller-owned device memory)
    float* W = nullptr;
    float* b = nullptr;
    int din = 0, dout = 0;
};

struct NormStats {  // device copies of the 12 stat vectors + derived
    float* buf = nullptr;  // one allocation
    float *obs_mean, *obs_std, *act_mean, *act_std, *delta_mean, *delta_std;
    float *cp_obs_mean, *cp_obs_std, *cp_act_mean, *cp_act_std, *back_delta_mean, *back_delta_std;
    bool set = false;
};

struct TrainState;  // train.hip
struct RolloutArgs;  // rollout_args.h

struct cadm_ctx {
    cadm_config cfg;
    int device = 0;
    int D, A, P, C, E, p, H, HID, NH, K0;
    // master weights
    std::vector<DenseRef> ff, back, cp;   // ff/back: NH hidden + mu + logvar; cp: n_cp_hidden + out
    float *ff_maxlv = nullptr, *ff_minlv = nullptr, *back_maxlv = nullptr, *back_minlv = nullptr;
    bool packed = false;
    bool train_packs_stale = true;  // the training chains' packed operand streams (train.hip) must be rebuilt from the master weights
    // planner weight stream (ff net only): split-f16 fragments of the rollout kernel (xdl_geo.h)
    XdlGeo xg;
    unsigned short* xw = nullptr;   // [E][member_frags] fragments of 2 KB
    float* xb = nullptr;            // [E][bias_tiles][64][4] D-layout bias tiles
    int* xflag = nullptr;           // device flag: a weight did not fit the f16 range
    // Developer hooks.  The product library never sets them (no entry point does, and it reads no environment variable);
    // libcadm_hip_dev.so adds dev/dev_api.hip, whose cadm_dev_set_rollout installs the fp32-MFMA comparison kernel of
    // round 1 (dev/rollout_f32.h) or forces a row-tile flavour of the production kernel.
    // side modules for geometries that are not compiled in (rollout_jit.hip, registered by cadm_register_rollout), per noise mode
    int (*jit_rollout[3])(cadm_ctx*, const struct RolloutArgs*, int rows_per_member, void* stream) = {nullptr, nullptr, nullptr};
    void (*set_error)(const char*, ...) = nullptr;      // = cadm_set_error of the library that owns the ctx
    int (*dev_rollout)(cadm_ctx*, const struct RolloutArgs&, int rows_per_member, hipStream_t) = nullptr;
    int (*dev_pack)(cadm_ctx*, hipStream_t) = nullptr;
    void (*dev_free)(cadm_ctx*) = nullptr;
    int dev_force_mt = 0;           // 0: launcher's plan; 1 / 2: ONE launch of the cooperative kernel with that many row tiles per workgroup; 3 / 4: of the wave-tile kernel (8 / 4 tiles per workgroup)
    LayerGeo g0, gh, go;            // fp32 fragment stream of the comparison kernel (allocated and packed by dev_pack only)
    float* wstream = nullptr;
    float* bstream = nullptr;
    size_t wstream_member_floats = 0, bstream_member_floats = 0;
    int n_cus = 256;
    std::unordered_set<const void*> attr_done;   // kernels whose dynamic-LDS attribute is set on THIS ctx's device
    size_t chain_attr_lds = 0;      // training chain kernel: the dynamic-LDS size its attribute was last raised to on this ctx's device
    size_t chain_attr_lds4 = 0;     // (its 4-wave throughput flavour)
    int train_force_nw = 0;         // developer library only: 4 / 8 = force that flavour of the chain kernel (0: by work items)
    int train_force_spread = 0;     // developer library only: 1 / 2 = force the chain kernel's work items spread over all XCDs / member-affine
    int train_force_merge = 0;      // developer library only: 1 / 2 = force / forbid the one-pass context backward of large batches
    NormStats st;
    TrainState* train = nullptr;
    // scratch for the context encoder
    float* cp_scratch = nullptr;
    size_t cp_scratch_floats = 0;
    // optional hipEvent bracketing of the rollout launches (cadm_profile_*)
    bool prof = false;
    std::vector<hipEvent_t> prof_ev;   // start/stop pairs around the rollout launches
    size_t prof_used = 0;
    std::vector<hipEvent_t> prof_ag;   // start/stop pairs around the per-iteration ncclAllGather (sharded planner)
    size_t prof_ag_used = 0;
    unsigned long long* tbuf = nullptr;   // cadm_dev_set_timing_buffer (developer library only)
    // completion flags of a staged planner call (cadm_cem_plan_staged): [m] words in pinned host memory, released by the last refit
    unsigned* plan_done = nullptr;
    unsigned plan_done_val = 0;
    // candidate-sharded planning (dist.hip): the RCCL communicator the ctx owns, OR an all-gather supplied by the host
    // (cadm_dist_init_external: torch.distributed over any backend) -- the planner loop is the same, only the collective differs
    void* comm = nullptr;
    int nranks = 1, rank = 0;
    cadm_allgather_fn ext_allgather = nullptr;
    void* ext_user = nullptr;
    unsigned* dist_flag = nullptr;   // device word: set by the sharded refit when the ranks' input checksums differ (cadm_dist_mismatch)
    // second packed copy of the planner weights for the wave-tile kernel (rollout_wt.h): ONE consumption order for every wave
    XdlGeo xg1;
    unsigned short* xw1 = nullptr;  // [E][member_frags of xg1] fragments of 2 KB
};

// Entry points launch on the ctx's device whatever device the caller's thread has current (two engines on different GPUs in
// one process; the ctor's `device=` kwarg), and restore the caller's device on return.
struct CadmDeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit CadmDeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~CadmDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    CadmDeviceGuard(const CadmDeviceGuard&) = delete;
    CadmDeviceGuard& operator=(const CadmDeviceGuard&) = delete;
};
#define CADM_ON_DEVICE(ctx) CadmDeviceGuard cadm_device_guard_((ctx)->device)

// kernels' host launchers (one per translation unit)
int cadm_pack_streams(cadm_ctx* ctx, hipStream_t s);
int cadm_pack_xdl(cadm_ctx* ctx, hipStream_t s);
int cadm_launch_rollout(cadm_ctx* ctx, const float* obs, const float* obs_rows, const float* ctx_vec,
                        const float* actions, const float* eps, int norm_actions, uint32_t seed,
                        uint32_t call, int it, int cand_offset, int n_global, int m, int n_local,
                        float* returns_rows, float* traj_out, hipStream_t s, int dry_run = 0, int force_deterministic = -1);
// Sharded planner (capi.hip: cem_plan_impl): what the refit needs to REGENERATE the elites' action sequences by global
// candidate id instead of reading them (a rank draws only its own shard), and to check the input checksums at the end of every rank's
// all-gather payload.
struct RefitRegen {
    int on;                    // 1: elite actions are drawn again from (seed, call, it); the actions pointer may be null
    uint32_t seed, call; int it;
    float lb, ub;
    int gstride;               // floats per rank in the gathered buffer (0: m * n_local); m * n_local + 1 with the trailing checksum
    int my_rank;               // >= 0: compare every rank's checksum with this rank's; mismatch -> NaN plan
    unsigned* mismatch;        // device word raised on a checksum mismatch (ctx->dist_flag), may be null
    unsigned* mismatch_host;   // pinned host words [m] behind the completion flags of a staged call, may be null
};
inline bool cadm_sharded(const cadm_ctx* c) { return c->comm != nullptr || c->ext_allgather != nullptr; }
int cadm_launch_input_checksum(cadm_ctx* ctx, const float* obs, const float* cp_obs, const float* cp_act, const float* mean, const float* var,
                               int m, unsigned* out, hipStream_t s);
int cadm_launch_context(cadm_ctx* ctx, const float* cp_obs, const float* cp_act, int m, int bs,
                        float* out, hipStream_t s);
// Per-call inputs of a small planner call travel as KERNEL ARGUMENTS (cadm_cem_plan_staged, capi.hip): up to CADM_INGEST_MAX floats.
#define CADM_INGEST_MAX 960
struct IngestBlock { float v[CADM_INGEST_MAX]; };
// the fused head's copy of the block shares the 4 KB kernel-argument segment with the encoder's and the sampler's arguments: a smaller cap
// (blocks between the two caps take the plain ingest kernel + the unfused head)
#define CADM_HEAD_INGEST_MAX 896
struct HeadBlock { float v[CADM_HEAD_INGEST_MAX]; };
#define CADM_CONTEXT_BATCHED_MIN_ROWS 48    // histories per member from which the context encoder runs as a GEMM chain (context.hip)
// The head of a staged planner call as ONE launch (context.hip: plan_head_kernel): unpack the ingest block into the device block, the
// context encoder on the block's history (C > 0), and the candidates of CEM iteration 0.  off[5] = float offsets of obs, cp_obs,
// cp_act, init_mean, init_var inside the block (-1: absent).
int cadm_launch_plan_head(cadm_ctx* ctx, const float* host_block, int nfloats, const int32_t off[5], float* dev_block, int m, int n,
                          uint32_t seed, uint32_t call, float* ctx_out, float* actions_out, hipStream_t s);
void cadm_train_free(cadm_ctx* ctx);
// (developer library: dev/dev_api.hip) Adam moment buffers of one trained tensor; layer as in cadm_set_weights, is_bias 0 / 1;
// layer == -1 / -2: max_logvar / min_logvar of the forward net.  Returns null pointers before cadm_train_configure.
int cadm_train_adam_slot(cadm_ctx* ctx, int net, int layer, int is_bias, float** m, float** v, size_t* n);
int cadm_dist_allgather(cadm_ctx* ctx, const float* send, float* recv, size_t count, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) {
    return (float)(x >> 8) * 5.9604644775390625e-08f + 2.98023223876953125e-08f;  // 2^-24, 2^-25
}

__device__ __forceinline__ void box_muller(float u1, float u2, float& z0, float& z1) {
    // hardware transcendentals: v_log_f32, v_sqrt_f32, v_cos_f32 / v_sin_f32 (argument in revolutions)
    const float r = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

// one candidate action element L = ((mi * n + c) * H + t) * A + a from the CEM distribution (mu, var) of its (t, a)
__device__ __forceinline__ float sample_action(float mu, float var, const float* __restrict__ z, size_t L, uint32_t seed, uint32_t call,
                                               int it, float lb, float ub) {
    const float lbd = mu - lb, ubd = ub - mu;                                  // :425
    const float a1 = lbd / 2.0f, a2 = ubd / 2.0f;
    const float cv = fminf(fminf(a1 * a1, a2 * a2), var);                      // :426
    float zz;
    if (z) {
        zz = z[L];
    } else {
        // TF TruncatedNormalDistribution: reject |x| >= 2 (kTruncateValue)
        zz = 0.0f;
        for (uint32_t attempt = 0; attempt < 64; ++attempt) {
            uint32_t r[4];
            philox4x32_10((uint32_t)(L & 0xFFFFFFFFull), attempt, (uint32_t)(L >> 32),
                          CADM_STREAM_ACT | ((uint32_t)it << 8), seed, call, r);
            float c0, c1, c2, c3;
            box_muller(u01(r[0]), u01(r[1]), c0, c1);
            if (fabsf(c0) < 2.0f) { zz = c0; break; }
            if (fabsf(c1) < 2.0f) { zz = c1; break; }
            box_muller(u01(r[2]), u01(r[3]), c2, c3);      // (the call's second pair: needed by 0.2 % of the draws)
            if (fabsf(c2) < 2.0f) { zz = c2; break; }
            if (fabsf(c3) < 2.0f) { zz = c3; break; }
        }
    }
    return mu + sqrtf(cv) * zz;                                                // :429
}


// (return desc, index asc) as one ascending 64-bit key: reproduces tf.nn.top_k's order, ties -> lower index (core/utils.py:475)
__device__ __forceinline__ uint64_t make_key(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-orderable
    return ((uint64_t)(~u) << 32) | idx;               // ascending key == descending value, ties -> lower idx
}


// tf.nn.softplus (TF 1.15 Eigen functor): threshold = log(eps) + 2
__device__ __forceinline__ float tf_softplus(float x) {
    const float thr = -13.942385f;  // logf(FLT_EPSILON) + 2
    if (x > -thr) return x;
    const float ex = expf(x);
    if (x < thr) return ex;
    return log1pf(ex);
}

// swish(x) = x * sigmoid(x) (dynamics.py:23); v_exp_f32 / v_rcp_f32 based, ~3 ulp
__device__ __forceinline__ float swish_f(float x) {
    const float e = __expf(-x);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
