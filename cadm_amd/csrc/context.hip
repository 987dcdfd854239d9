// Context encoder inference (reference core/utils.py:401-406 + :614-617; get_context_pred,
// dynamics.py:369-380): normalise the history window, 3 ReLU layers, linear output.
// Two kernels: `context_kernel` for the planner's call (once per get_action on E*m rows, 5 at m = 1, ~1 MFLOP: a latency
// kernel, one workgroup per (member, row), K split across its waves and reduced through LDS) and `context_batched_kernel` for
// many histories per call (get_context_pred from the PPO consumer, SURVEY 8f-3: a GEMM chain on the fp32 matrix pipe).
#include "common.h"

#define CP_MAX_WIDTH 1024
#define CADM_CONTEXT_BATCHED_MIN_ROWS 48    // per member: below, one latency-tuned workgroup per row (the planner's m = 1..10)

struct CpArgs {
    const float* W[CADM_MAX_CP_LAYERS + 1];
    const float* b[CADM_MAX_CP_LAYERS + 1];
    int dims[CADM_MAX_CP_LAYERS + 2];  // in, h0, h1, ..., out
    int nlayers;                       // dense layers including the output
    const float *cp_obs, *cp_act;
    const float *obs_mean, *obs_std, *act_mean, *act_std;
    int n_obs, n_act;                  // D*Hh, A*Hh
    int m, bs, E;
    float* out;
};

// 16 waves per (member, env): a layer is a latency chain of weight rows (one workgroup reads the member's whole 430 KB),
// so the K dimension is cut 16 ways -- every wave has all of its (at most 15) rows in flight at once.
#define CP_THREADS 1024
#define CP_KQ (CP_THREADS / 64)
__global__ __launch_bounds__(CP_THREADS) void context_kernel(const CpArgs a) {
    __shared__ float xa[CP_MAX_WIDTH];
    __shared__ float xb[CP_MAX_WIDTH];
    __shared__ float red[CP_KQ][256];
    const int e = blockIdx.x / a.m, mi = blockIdx.x % a.m;
    const int tid = threadIdx.x;
    const size_t in_row = a.bs ? ((size_t)e * a.m + mi) : (size_t)mi;   // tile(.., [E,1,1]) unless already [E,m,.]
    for (int i = tid; i < a.n_obs; i += CP_THREADS)
        xa[i] = (a.cp_obs[in_row * a.n_obs + i] - a.obs_mean[i]) / (a.obs_std[i] + 1e-10f);          // :403
    for (int i = tid; i < a.n_act; i += CP_THREADS)
        xa[a.n_obs + i] = (a.cp_act[in_row * a.n_act + i] - a.act_mean[i]) / (a.act_std[i] + 1e-10f);  // :404
    __syncthreads();
    float* xin = xa;
    float* xout = xb;
    for (int l = 0; l < a.nlayers; ++l) {
        const int K = a.dims[l], N = a.dims[l + 1];
        const float* W = a.W[l] + (size_t)e * K * N;
        const float* b = a.b[l] + (size_t)e * N;
        // thread (kq = tid >> 6, lane = tid & 63): 4 consecutive columns per lane (float4 weight loads when
        // N % 4 == 0), K split in CP_KQ contiguous parts across the waves, partials reduced through LDS in a fixed order.
        const int kq = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (uniform: row addresses stay in SGPRs)
        const int k0 = (K * kq) / CP_KQ, k1 = (K * (kq + 1)) / CP_KQ;
        const bool vec = (N & 3) == 0;
        for (int nb = 0; nb < N; nb += 256) {
            const int n = nb + lane * 4;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (n < N) {
                if (vec) {
                    // 16 rows in flight per pass; rows past k1 are re-reads of the last row weighted by 0 (loads stay
                    // unconditional, so they are issued together)
                    for (int k = k0; k < k1; k += 16) {
                        floatx4 w[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int kk = k + u < k1 ? k + u : k1 - 1;
                            w[u] = *reinterpret_cast<const floatx4*>(W + (size_t)kk * N + n);
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const float xv = k + u < k1 ? xin[k + u] : 0.0f;
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[c] = fmaf(xv, w[u][c], acc[c]);
                        }
                    }
                } else {
                    // (columns past N read the last column: unconditional loads; their sums are never consumed)
#pragma unroll 4
                    for (int k = k0; k < k1; ++k) {
                        const float xv = xin[k];
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = fmaf(xv, W[(size_t)k * N + (n + c < N ? n + c : N - 1)], acc[c]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) red[kq][lane * 4 + c] = acc[c];
            __syncthreads();
            {
                const int nn = nb + tid;
                if (tid < 256 && nn < N) {
                    float v = red[0][tid];
#pragma unroll
                    for (int q = 1; q < CP_KQ; ++q) v += red[q][tid];
                    v += b[nn];
                    if (l + 1 < a.nlayers) v = fmaxf(v, 0.0f);   // ReLU hidden (layers.py:34), identity output
                    xout[nn] = v;
                }
            }
            __syncthreads();
        }
        float* t = xin; xin = xout; xout = t;
    }
    const int C = a.dims[a.nlayers];
    for (int i = tid; i < C; i += CP_THREADS) a.out[((size_t)e * a.m + mi) * C + i] = xin[i];
}


// ---------------------------------------------------------------------------------------------------------------------------
// Batched context inference (the PPO consumer calls get_context_pred on nenvs x nsteps histories per update,
// model_free/ppo_cadm.py:155-162; dynamics.py:369-380): thousands of rows per member are a GEMM chain, not a GEMV -- a member's
// 412 KB of weights must be reused across rows instead of re-read per row (context_kernel above: one workgroup per row).
//
// One workgroup = 4 waves = 16 * RT rows of ONE member through all layers, activations resident in LDS, exact fp32 arithmetic
// on v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain over k: the same numbers as the per-row kernel up to summation order).
// A layer is evaluated transposed, OUT^T = W^T IN^T: the weights are the A operand, read STRAIGHT from the row-major master
// tensor W[k][n] -- a "tile" of 16 output units is the STRIDED set {u0 + 4 i + t : i = 0..15}, so one 16-byte load per lane
// (4 consecutive units of row k) is the A operand of FOUR tiles at once and a wave covers 64 consecutive units with 16
// accumulator registers per row tile; the 16 * RT data rows are the B / D columns.  Weight fragments are requested CP_PF k-steps
// ahead into a register ring (the stream comes out of the member's L2-resident tensor: ~61 KB per wave and layer).
// LDS: the network input as [row][k] (odd row stride: the coalesced global read stores conflict-free), every later activation
// as [unit][row] with the row index XOR-swizzled by the unit's parity (a B operand = 4 k x 16 rows then touches 32 distinct banks).
// ---------------------------------------------------------------------------------------------------------------------------
#define CPB_THREADS 256
#define CPB_WAVES 4
#define CPB_PF 6                       // k-steps of weight fragments in flight per wave
struct CpbArgs {
    CpArgs a;
    int rows_per_member;               // m
    int region0, region1;              // LDS floats of the two activation regions
    int in_stride;                     // row stride of the input tile (odd)
};

template <int RT>
__device__ __forceinline__ void cpb_layer(const float* __restrict__ W, const float* __restrict__ bias, int K, int N, bool relu, bool in_rowmajor,
                                          int in_stride, const float* xin, float* xout, float* gout, int gld, int grow0, int grows,
                                          int wave, int lane) {
    constexpr int R = 16 * RT;
    const int i = lane & 15, q = lane >> 4;
    const int NG = (N + 63) >> 6;                                 // groups of 64 output units
    const int split = NG >= CPB_WAVES ? 1 : RT;                   // few groups: the row tiles of a group go to different waves
    const int nitems = NG * split;
    const int nsteps = (K + 3) >> 2;
    const bool vec = (N & 3) == 0;
    for (int item = wave; item < nitems; item += CPB_WAVES) {
        const int g = item % NG, rt0 = split == 1 ? 0 : item / NG, nrt = split == 1 ? RT : 1;
        const int u0 = g * 64 + 4 * i;                            // this lane's 4 consecutive units (tile t: unit u0 + t)
        floatx4 acc[4][RT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[t][r] = floatx4{0.f, 0.f, 0.f, 0.f};
        // A operand of k-step s: row k = 4 s + q of W, units u0 .. u0 + 3 (rows / units past the matrix: clamped loads -- the
        // B operand is zero for k >= K, and units >= N are never stored)
        auto fetch = [&](int s) -> floatx4 {
            int k = 4 * s + q;
            k = k < K ? k : K - 1;
            if (vec) {
                const int u = u0 < N ? u0 : N - 4;
                return *reinterpret_cast<const floatx4*>(W + (size_t)k * N + u);
            }
            floatx4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = W[(size_t)k * N + (u0 + t < N ? u0 + t : N - 1)];
            return v;
        };
        floatx4 ring[CPB_PF];
#pragma unroll
        for (int u = 0; u < CPB_PF; ++u) ring[u] = fetch(u < nsteps ? u : nsteps - 1);
        for (int s0 = 0; s0 < nsteps; s0 += CPB_PF) {
#pragma unroll
            for (int u = 0; u < CPB_PF; ++u) {
                const int s = s0 + u;
                if (s < nsteps) {                                 // (uniform)
                    const floatx4 av = ring[u];
                    const int sn = s + CPB_PF;
                    ring[u] = fetch(sn < nsteps ? sn : nsteps - 1);
                    const int k = 4 * s + q;
                    float bv[RT];
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        if (r < nrt) {
                            const int row = 16 * (rt0 + r) + i;
                            bv[r] = k < K ? (in_rowmajor ? xin[row * in_stride + k] : xin[k * R + (row ^ (16 * (k & 1) * (RT - 1)))]) : 0.0f;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < RT; ++r)
                            if (r < nrt) acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[r], acc[t][r], 0, 0, 0);
                }
            }
        }
        // D layout: lane (col = i -> data row, q), register rr -> A-row 4 q + rr -> unit g * 64 + 4 (4 q + rr) + t
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int unit = g * 64 + 4 * (4 * q + rr) + t;
                const float bsv = bias[unit < N ? unit : N - 1];
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (r < nrt && unit < N) {
                        float v = acc[t][r][rr] + bsv;
                        if (relu) v = fmaxf(v, 0.0f);                                    // ReLU hidden (layers.py:34), identity output
                        const int row = 16 * (rt0 + r) + i;
                        if (gout) { if (row < grows) gout[(size_t)(grow0 + row) * gld + unit] = v; }
                        else xout[unit * R + (row ^ (16 * (unit & 1) * (RT - 1)))] = v;
                    }
                }
            }
    }
}

template <int RT>
__global__ __launch_bounds__(CPB_THREADS) void context_batched_kernel(const CpbArgs p) {
    extern __shared__ __attribute__((aligned(16))) float cpb_smem[];
    constexpr int R = 16 * RT;
    const CpArgs& a = p.a;
    const int tiles = (p.rows_per_member + R - 1) / R;
    const int e = blockIdx.x / tiles, row0 = (blockIdx.x - e * tiles) * R;
    const int rows = p.rows_per_member - row0 < R ? p.rows_per_member - row0 : R;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* reg0 = cpb_smem;
    float* reg1 = cpb_smem + p.region0;
    // input tile [row][k], normalised on the way in (core/utils.py:403-404); rows past the batch are zeros
    const int K0 = a.dims[0], ist = p.in_stride;
    for (int idx = tid; idx < R * K0; idx += CPB_THREADS) {
        const int row = idx / K0, k = idx - row * K0;
        float v = 0.0f;
        if (row < rows) {
            const size_t in_row = a.bs ? ((size_t)e * a.m + row0 + row) : (size_t)(row0 + row);     // tile(.., [E,1,1]) unless already [E,m,.]
            v = k < a.n_obs ? (a.cp_obs[in_row * a.n_obs + k] - a.obs_mean[k]) / (a.obs_std[k] + 1e-10f)
                            : (a.cp_act[in_row * a.n_act + (k - a.n_obs)] - a.act_mean[k - a.n_obs]) / (a.act_std[k - a.n_obs] + 1e-10f);
        }
        reg0[row * ist + k] = v;
    }
    __syncthreads();
    const float* xin = reg0;
    for (int l = 0; l < a.nlayers; ++l) {
        const int K = a.dims[l], N = a.dims[l + 1];
        const bool last = l + 1 == a.nlayers;
        float* xout = (l & 1) ? reg0 : reg1;
        cpb_layer<RT>(a.W[l] + (size_t)e * K * N, a.b[l] + (size_t)e * N, K, N, !last, l == 0, ist, xin, xout,
                      last ? a.out : nullptr, N, e * p.rows_per_member + row0, rows, wave, lane);
        __syncthreads();
        xin = xout;
    }
}

static int launch_context_batched(cadm_ctx* ctx, const CpArgs& a, int m, hipStream_t s, bool* launched) {
    *launched = false;
    CpbArgs p{};
    p.a = a;
    p.rows_per_member = m;
    p.in_stride = a.dims[0] | 1;
    for (int RT = 2; RT >= 1; --RT) {
        const int R = 16 * RT;
        // region 0: the input tile, later the outputs of the odd layers; region 1: the outputs of the even layers
        size_t r0 = (size_t)R * p.in_stride, r1 = 0;
        for (int l = 0; l + 1 < a.nlayers; ++l) {
            const size_t need = (size_t)R * a.dims[l + 1];
            if (l & 1) r0 = need > r0 ? need : r0; else r1 = need > r1 ? need : r1;
        }
        r0 = (r0 + 3) & ~(size_t)3;
        const size_t lds = (r0 + r1) * sizeof(float);
        if (lds > 160 * 1024) continue;
        p.region0 = (int)r0; p.region1 = (int)r1;
        const void* fn = RT == 2 ? reinterpret_cast<const void*>(&context_batched_kernel<2>) : reinterpret_cast<const void*>(&context_batched_kernel<1>);
        if (lds > 64 * 1024 && !ctx->attr_done.count(fn)) {
            CADM_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->attr_done.insert(fn);
        }
        const int tiles = (m + R - 1) / R;
        if (RT == 2) hipLaunchKernelGGL(context_batched_kernel<2>, dim3(ctx->E * tiles), dim3(CPB_THREADS), lds, s, p);
        else hipLaunchKernelGGL(context_batched_kernel<1>, dim3(ctx->E * tiles), dim3(CPB_THREADS), lds, s, p);
        CADM_CHECK_HIP(hipGetLastError());
        *launched = true;
        return CADM_OK;
    }
    return CADM_OK;      // layers too wide for an LDS-resident tile: the per-row kernel takes the call
}

int cadm_launch_context(cadm_ctx* ctx, const float* cp_obs, const float* cp_act, int m, int bs, float* out,
                        hipStream_t s) {
    CpArgs a{};
    const int nl = ctx->cfg.n_cp_hidden + 1;
    a.nlayers = nl;
    a.dims[0] = (ctx->D + ctx->A) * ctx->cfg.history_length;
    for (int l = 0; l < nl; ++l) {
        if (!ctx->cp[l].W || !ctx->cp[l].b) {
            cadm_set_error("cadm_context_forward: context_model layer %d has no registered weights", l);
            return CADM_ESTATE;
        }
        a.W[l] = ctx->cp[l].W;
        a.b[l] = ctx->cp[l].b;
        a.dims[l + 1] = ctx->cp[l].dout;
        if (ctx->cp[l].dout > CP_MAX_WIDTH || ctx->cp[l].din > CP_MAX_WIDTH) {
            cadm_set_error("cadm_context_forward: layer width > %d unsupported", CP_MAX_WIDTH);
            return CADM_EINVAL;
        }
    }
    a.cp_obs = cp_obs; a.cp_act = cp_act;
    a.obs_mean = ctx->st.cp_obs_mean; a.obs_std = ctx->st.cp_obs_std;
    a.act_mean = ctx->st.cp_act_mean; a.act_std = ctx->st.cp_act_std;
    a.n_obs = ctx->D * ctx->cfg.history_length;
    a.n_act = ctx->A * ctx->cfg.history_length;
    a.m = m; a.bs = bs; a.E = ctx->E; a.out = out;
    if (m >= CADM_CONTEXT_BATCHED_MIN_ROWS) {       // many histories per member: the GEMM-shaped path (weights reused across rows)
        bool launched = false;
        const int rc = launch_context_batched(ctx, a, m, s, &launched);
        if (rc || launched) return rc;
    }
    hipLaunchKernelGGL(context_kernel, dim3(ctx->E * m), dim3(CP_THREADS), 0, s, a);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}
