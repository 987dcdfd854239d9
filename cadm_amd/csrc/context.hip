// Context encoder inference (reference core/utils.py:401-406 + :614-617; get_context_pred,
// dynamics.py:369-380): normalise the history window, 3 ReLU layers, linear output.
// Runs once per get_action on E*m rows (5 at m = 1), ~1 MFLOP: a latency kernel, not a throughput one.
// One workgroup per (member, row); K is split 4 ways across the block and reduced through LDS.
#include "common.h"

#define CP_MAX_WIDTH 1024

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
    hipLaunchKernelGGL(context_kernel, dim3(ctx->E * m), dim3(CP_THREADS), 0, s, a);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}
