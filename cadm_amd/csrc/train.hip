// Training step of the ensemble (reference cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:269-317;
// vanilla twin mlp_ensemble_cem_dynamics.py:148-170): forward of the context / forward / backward
// nets on one [E,B,.] bootstrap batch, the losses, hand-written backward GEMMs and TF1-semantics Adam.
//
// One batched fp32-MFMA GEMM kernel (v_mfma_f32_16x16x4_f32, 64x64 tile per workgroup, LDS-staged
// 32-deep K slabs) serves all three products through strides, with the layer's pointwise work fused
// into its epilogue:
//   FWD  H = act(X W + b)            stores z (pre-activation) and h
//   DX   dZ_prev = (dZ W^T) * act'(z_prev)   (optionally accumulating: the context vector feeds 2 nets)
//   DW   W <- Adam(W, X^T dZ + c*wd*W), b <- Adam(b, colsum dZ)   -- the gradient never touches HBM
// Everything is launch-ordered on one stream: a layer's DX (which reads W) runs before its DW (which
// overwrites W).
#include <math.h>

#include "common.h"

namespace {

enum { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2 };
enum { MODE_FWD = 0, MODE_DX = 1, MODE_DW = 2 };

struct GemmP {
    const float *A, *B;
    long sAe, sBe;                 // member strides (elements)
    int M, N, K, E;                // C[e] is M x N, reduction over K
    long sam, sak, sbk, sbn;       // A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
    int a_mcontig;                 // 1: A is contiguous along m (load mapping for coalescing)
    int mode, act;
    // FWD
    const float* bias; long sbe;   // bias[e][n]
    float* Zout; float* Hout; long ldo, sOe;   // z / h outputs [E][M][ldo] (+ column offset baked into the pointer)
    // DX
    const float* Zprev; long ldzp, sZpe;       // pre-activation of the producing layer [E][M][ldzp]
    float* DXout; long lddx, sDXe; int accumulate;
    // DW (+ Adam)
    float *W, *Mw, *Vw; long ldw, sWe;         // W[e][M][N] row-major (ldw = N)
    float *bW, *bM, *bV; long sbWe;            // bias[e][N] (null: no bias update)
    float wdc;                                 // weight_decay_coeff * wd (d l2 / dW = wdc * W)
    float lr_t, b1, b2, eps;
};

__device__ __forceinline__ float act_fwd(int act, float z) {
    if (act == ACT_SWISH) return z * (1.0f / (1.0f + expf(-z)));
    if (act == ACT_RELU) return fmaxf(z, 0.0f);
    return z;
}
__device__ __forceinline__ float act_bwd(int act, float z) {   // d act / d z
    if (act == ACT_SWISH) {
        const float s = 1.0f / (1.0f + expf(-z));
        return s * (1.0f + z * (1.0f - s));
    }
    if (act == ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
    return 1.0f;
}

__device__ __forceinline__ void adam_update(float& w, float& m, float& v, float g, float lr_t, float b1, float b2,
                                            float eps) {
    // tf.compat.v1.train.AdamOptimizer (training_ops ApplyAdam): m,v EMA; w -= lr_t * m / (sqrt(v) + eps)
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    w -= lr_t * m / (sqrtf(v) + eps);
}

#define TN 64
#define TK 32
#define NSLAB 8                    // slabs per K panel: the whole panel (256 deep) is in flight at once
#define LDB (TN + 4)

// These GEMMs are tiny (<= 0.1 GFLOP over <= 160 workgroups): a launch is bound by the serial
// load -> LDS -> MFMA latency chain of one workgroup, not by throughput.  So a workgroup issues the
// global loads of a whole 256-deep K panel up front (registers), and then walks the panel slab by
// slab -- stash slab s into its own LDS region as its loads land (the compiler's vmcnt ladder),
// barrier, 32-deep MFMA sweep -- so the only exposed memory latency is the first slab's.
// TM = 32*MI rows x 64 columns per workgroup; 2 x 2 waves, each (16*MI) x 32.
template <int MI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
    constexpr int TM = 32 * MI;
    constexpr int LDA = TM + 4;
    extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
    float* const As = gemm_smem;                          // [NSLAB*TK][LDA]
    float* const Bs = gemm_smem + NSLAB * TK * LDA;       // [NSLAB*TK][LDB]
    const int e = blockIdx.z;
    const int mb = blockIdx.y * TM, nb = blockIdx.x * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float* A = p.A + (long)e * p.sAe;
    const float* B = p.B + (long)e * p.sBe;
    floatx4 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    float colsum = 0.0f;                                // DW: bias gradient (threads < 64 of the m-tile-0 blocks)

    constexpr int NLA = TM * TK / 256, NLB = TN * TK / 256;   // elements per thread per slab
    float ra[NSLAB][NLA], rb[NSLAB][NLB];
    // Per-thread element bases are fixed across slabs.  Loads are UNCONDITIONAL (addresses clamped into the matrix,
    // out-of-range elements zeroed afterwards): a `cond ? *p : 0` select makes hipcc branch around every load and
    // wait for each one in turn (cdna_hip_programming.md, "three .s-level traps" (c)).
    const float* pa[NLA];
    const float* pb[NLB];
    int ka[NLA], kb[NLB], la[NLA], lb[NLB];
    bool va[NLA], vb[NLB];
#pragma unroll
    for (int it = 0; it < NLA; ++it) {
        const int idx = tid + it * 256;
        int am, ak;
        if (p.a_mcontig) { am = idx & (TM - 1); ak = idx / TM; } else { ak = idx & (TK - 1); am = idx / TK; }
        ka[it] = ak; la[it] = ak * LDA + am;
        va[it] = mb + am < p.M;
        pa[it] = A + (long)(va[it] ? mb + am : 0) * p.sam;
    }
#pragma unroll
    for (int it = 0; it < NLB; ++it) {
        const int idx = tid + it * 256;
        int bn, bk;
        if (p.sbn == 1) { bn = idx & (TN - 1); bk = idx / TN; } else { bk = idx & (TK - 1); bn = idx / TK; }
        kb[it] = bk; lb[it] = bk * LDB + bn;
        vb[it] = nb + bn < p.N;
        pb[it] = B + (long)(vb[it] ? nb + bn : 0) * p.sbn;
    }
    const int kmax = p.K - 1;
    const bool do_colsum = p.mode == MODE_DW && p.bW && blockIdx.y == 0 && tid < TN;

    for (int kp = 0; kp < p.K; kp += NSLAB * TK) {
        if (kp > 0) __syncthreads();               // previous panel fully consumed before its LDS is overwritten
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
            const int k0 = kp + s * TK;
#pragma unroll
            for (int it = 0; it < NLA; ++it) {
                const int k = k0 + ka[it];
                ra[s][it] = pa[it][(long)(k < kmax ? k : kmax) * p.sak];
            }
#pragma unroll
            for (int it = 0; it < NLB; ++it) {
                const int k = k0 + kb[it];
                rb[s][it] = pb[it][(long)(k < kmax ? k : kmax) * p.sbk];
            }
        }
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
            const int k0 = kp + s * TK;
            if (k0 >= p.K) break;
            float* as = As + s * TK * LDA;
            float* bs = Bs + s * TK * LDB;
#pragma unroll
            for (int it = 0; it < NLA; ++it) as[la[it]] = (va[it] && k0 + ka[it] <= kmax) ? ra[s][it] : 0.0f;
#pragma unroll
            for (int it = 0; it < NLB; ++it) bs[lb[it]] = (vb[it] && k0 + kb[it] <= kmax) ? rb[s][it] : 0.0f;
            __syncthreads();
            if (do_colsum) {
#pragma unroll
                for (int kk = 0; kk < TK; ++kk) colsum += bs[kk * LDB + tid];
            }
#pragma unroll
            for (int ks = 0; ks < TK / 4; ++ks) {
                float a[MI], b[2];
                const int kr = ks * 4 + (lane >> 4);
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = as[kr * LDA + wm * (16 * MI) + i * 16 + (lane & 15)];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = bs[kr * LDB + wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: D layout col = lane & 15 -> n, row = (lane >> 4) * 4 + r -> m ----
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + wm * (16 * MI) + i * 16 + (lane >> 4) * 4 + r;
                const int n = nb + wn * 32 + j * 16 + (lane & 15);
                if (m >= p.M || n >= p.N) continue;
                const float c = acc[i][j][r];
                if (p.mode == MODE_FWD) {
                    const float z = c + p.bias[(long)e * p.sbe + n];
                    const long o = (long)e * p.sOe + (long)m * p.ldo + n;
                    if (p.Zout) p.Zout[o] = z;
                    p.Hout[o] = act_fwd(p.act, z);
                } else if (p.mode == MODE_DX) {
                    float g = c;
                    if (p.Zprev) g *= act_bwd(p.act, p.Zprev[(long)e * p.sZpe + (long)m * p.ldzp + n]);
                    const long o = (long)e * p.sDXe + (long)m * p.lddx + n;
                    p.DXout[o] = p.accumulate ? p.DXout[o] + g : g;
                } else {
                    const long o = (long)e * p.sWe + (long)m * p.ldw + n;
                    float w = p.W[o], mo = p.Mw[o], vo = p.Vw[o];
                    adam_update(w, mo, vo, c + p.wdc * w, p.lr_t, p.b1, p.b2, p.eps);
                    p.W[o] = w; p.Mw[o] = mo; p.Vw[o] = vo;
                }
            }
    if (do_colsum && nb + tid < p.N) {
        const long o = (long)e * p.sbWe + nb + tid;
        float w = p.bW[o], mo = p.bM[o], vo = p.bV[o];
        adam_update(w, mo, vo, colsum, p.lr_t, p.b1, p.b2, p.eps);
        p.bW[o] = w; p.bM[o] = mo; p.bV[o] = vo;
    }
}

__global__ void mul_actgrad_kernel(float* g, const float* z, long n, int act) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) g[i] *= act_bwd(act, z[i]);
}

// elementwise Adam for tensors whose gradient is a closed form: g = gscale * gsrc (+ wdc * w)
__global__ void adam_elem_kernel(float* w, float* m, float* v, const float* gsrc, float gscale, float gconst, float wdc,
                                 long n, float lr_t, float b1, float b2, float eps) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float ww = w[i], mm = m[i], vv = v[i];
    const float g = (gsrc ? gscale * gsrc[i] : 0.0f) + gconst + wdc * ww;
    adam_update(ww, mm, vv, g, lr_t, b1, b2, eps);
    w[i] = ww; m[i] = mm; v[i] = vv;
}

// ---------------------------------------------------------------------------------------------
// input assembly (core/utils.py:372-379 and :619-621 of the reference)
// ---------------------------------------------------------------------------------------------
struct AsmP {
    const float *obs, *obs_next, *act, *cp_obs, *cp_act;
    const float *obs_mean, *obs_std, *act_mean, *act_std, *cp_obs_mean, *cp_obs_std, *cp_act_mean, *cp_act_std;
    float *Xff, *Xbk, *Xcp;
    int rows, D, A, P, K0, ncpo, ncpa, env, has_back, has_cp;
};

__device__ __forceinline__ float preproc_at(int env, const float* o, int pf) {
    if (env == CADM_ENV_HALFCHEETAH) {
        if (pf == 0) return o[1];
        if (pf == 1) return sinf(o[2]);
        if (pf == 2) return cosf(o[2]);
        return o[pf];
    }
    if (env == CADM_ENV_ANT) return o[pf + 1];
    return o[pf];
}

__global__ void assemble_kernel(const AsmP p) {
    const int row = blockIdx.x;                       // e * B + b
    for (int f = threadIdx.x; f < p.P + p.A; f += blockDim.x) {
        if (f < p.P) {
            const float inv = p.obs_std[f] + 1e-10f;
            p.Xff[(long)row * p.K0 + f] = (preproc_at(p.env, p.obs + (long)row * p.D, f) - p.obs_mean[f]) / inv;
            if (p.has_back) p.Xbk[(long)row * p.K0 + f] = (preproc_at(p.env, p.obs_next + (long)row * p.D, f) - p.obs_mean[f]) / inv;
        } else {
            const int a = f - p.P;
            const float v = (p.act[(long)row * p.A + a] - p.act_mean[a]) / (p.act_std[a] + 1e-10f);
            p.Xff[(long)row * p.K0 + f] = v;
            if (p.has_back) p.Xbk[(long)row * p.K0 + f] = v;
        }
    }
    if (p.has_cp) {
        const int n = p.ncpo + p.ncpa;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            float v;
            if (i < p.ncpo) v = (p.cp_obs[(long)row * p.ncpo + i] - p.cp_obs_mean[i]) / (p.cp_obs_std[i] + 1e-10f);
            else v = (p.cp_act[(long)row * p.ncpa + (i - p.ncpo)] - p.cp_act_mean[i - p.ncpo]) / (p.cp_act_std[i - p.ncpo] + 1e-10f);
            p.Xcp[(long)row * n + i] = v;
        }
    }
}

__global__ void copy_cols_kernel(const float* src, long lds_, float* dst, long ldd, int cols, long rows) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const long r = i / cols;
    const int c = (int)(i % cols);
    dst[r * ldd + c] = src[r * lds_ + c];
}

// ---------------------------------------------------------------------------------------------
// losses (dynamics.py:269-314) and output-layer gradients
// ---------------------------------------------------------------------------------------------
struct LossP {
    const float *mu, *lv, *bmu;              // head outputs [E*B, D]
    const float *delta, *back_delta;         // raw targets [E*B, D]
    const float *dmean, *dstd, *bdmean, *bdstd, *maxlv, *minlv;
    float *dMu, *dLv, *dBmu;                 // d loss / d head pre-activation
    float *terms;                            // [7][E*B*D]: mse, mu_loss, var_loss, back_mse, g_maxlv, g_minlv, (unused)
    long n;                                  // E*B*D
    int D, B, det, has_back;
    float back_coeff;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void loss_kernel(const LossP p) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const int d = (int)(i % p.D);
    const float s = 1.0f / ((float)p.B * (float)p.D);         // reduce_mean over b then d; reduce_sum over e
    const float t = (p.delta[i] - p.dmean[d]) / (p.dstd[d] + 1e-10f);
    const float mu = p.mu[i];
    const float diff = mu - t;
    p.terms[0 * p.n + i] = diff * diff * s;                                   // mse            (:273-274)
    if (p.det) {
        p.dMu[i] = 2.0f * s * diff;
        p.dLv[i] = 0.0f;
        p.terms[1 * p.n + i] = 0.0f; p.terms[2 * p.n + i] = 0.0f; p.terms[4 * p.n + i] = 0.0f; p.terms[5 * p.n + i] = 0.0f;
    } else {
        const float mx = p.maxlv[d], mn = p.minlv[d], lv0 = p.lv[i];
        const float u = mx - tf_softplus(mx - lv0);                           // core/utils.py:356
        const float lvc = mn + tf_softplus(u - mn);                           // core/utils.py:357
        const float invvar = expf(-lvc);                                      // :303
        p.terms[1 * p.n + i] = diff * diff * invvar * s;                      // mu_loss        (:304-305)
        p.terms[2 * p.n + i] = lvc * s;                                       // var_loss       (:306-307)
        const float g_lvc = s * (1.0f - diff * diff * invvar);
        const float s1 = sigmoidf_(u - mn), s2 = sigmoidf_(mx - lv0);         // softplus' = sigmoid
        p.dMu[i] = 2.0f * s * diff * invvar;
        p.dLv[i] = g_lvc * s1 * s2;
        p.terms[4 * p.n + i] = g_lvc * s1 * (1.0f - s2);                      // d / d max_logvar (without the 0.01 reg)
        p.terms[5 * p.n + i] = g_lvc * (1.0f - s1);                           // d / d min_logvar
    }
    if (p.has_back) {
        const float tb = (p.back_delta[i] - p.bdmean[d]) / (p.bdstd[d] + 1e-10f);
        const float db = p.bmu[i] - tb;
        p.terms[3 * p.n + i] = db * db * s;                                   // back_mse       (:280-281)
        p.dBmu[i] = p.back_coeff * 2.0f * s * db;
    } else {
        p.terms[3 * p.n + i] = 0.0f;
    }
}

// deterministic reductions: block q < 4 sums terms[q] over everything; block 4 + d / 4 + D + d sum the
// max/min logvar gradient terms over rows for dim d.  out: [4 + 2D]
__global__ void reduce_kernel(const float* terms, long n, int D, float* out) {
    __shared__ float sh[256];
    const int q = blockIdx.x;
    float acc = 0.0f;
    if (q < 4) {
        for (long i = threadIdx.x; i < n; i += 256) acc += terms[q * n + i];
    } else {
        const int which = (q - 4) / D, d = (q - 4) % D;
        const float* src = terms + (4 + which) * n;
        for (long r = threadIdx.x; r * D + d < n; r += 256) acc += src[r * D + d];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[q] = sh[0];
}

// losses_out = [mse, back_mse, recon]  (dynamics.py:505-507: recon = loss - reg - coeff*l2)
__global__ void finalize_loss_kernel(const float* red, int det, float back_coeff, int has_back, float* losses_out) {
    const float mse = red[0], mu_loss = red[1], var_loss = red[2], back = red[3];
    float recon = det ? mse : mu_loss + var_loss;
    if (has_back) recon += back_coeff * back;
    losses_out[0] = mse;
    losses_out[1] = has_back ? back : 0.0f;
    losses_out[2] = recon;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct NetBufs {
    std::vector<float*> z, h;     // per hidden layer [E,B,width]
    float *mu = nullptr, *lv = nullptr;
};

struct AdamSlot { float *m = nullptr, *v = nullptr; size_t n = 0; };

struct TrainState {
    cadm_train_hparams hp{};
    bool configured = false;
    int B = 0;                    // workspace capacity (rows per member)
    long step = 0;
    float* ws = nullptr;          // one workspace allocation
    size_t ws_floats = 0;
    // views
    float *Xff = nullptr, *Xbk = nullptr, *Xcp = nullptr, *dCtx = nullptr, *ctxo = nullptr;
    NetBufs ff, bk, cp;
    float *dA = nullptr, *dBuf = nullptr, *dMu = nullptr, *dLv = nullptr, *dBmu = nullptr, *terms = nullptr, *red = nullptr;
    // Adam moments, same order as the registered layers: W then b
    std::vector<AdamSlot> a_ff, a_bk, a_cp;   // 2 per layer
    AdamSlot a_mx, a_mn;
    float* adam_buf = nullptr;
};

void cadm_train_free(cadm_ctx* ctx) {
    if (!ctx->train) return;
    if (ctx->train->ws) (void)hipFree(ctx->train->ws);
    if (ctx->train->adam_buf) (void)hipFree(ctx->train->adam_buf);
    delete ctx->train;
    ctx->train = nullptr;
}

static int alloc_adam(cadm_ctx* ctx) {
    TrainState* t = ctx->train;
    size_t total = 0;
    auto count = [&](const std::vector<DenseRef>& v) { for (auto& d : v) total += 2 * ((size_t)ctx->E * d.din * d.dout + (size_t)ctx->E * d.dout); };
    count(ctx->ff);
    if (ctx->cfg.back_model) count(ctx->back);
    if (ctx->C > 0) count(ctx->cp);
    total += 4 * (size_t)ctx->D;
    CADM_CHECK_HIP(hipMalloc(&t->adam_buf, total * sizeof(float)));
    CADM_CHECK_HIP(hipMemset(t->adam_buf, 0, total * sizeof(float)));
    float* q = t->adam_buf;
    auto carve = [&](const std::vector<DenseRef>& v, std::vector<AdamSlot>& out) {
        out.clear();
        for (auto& d : v) {
            AdamSlot w, b;
            w.n = (size_t)ctx->E * d.din * d.dout; w.m = q; q += w.n; w.v = q; q += w.n;
            b.n = (size_t)ctx->E * d.dout; b.m = q; q += b.n; b.v = q; q += b.n;
            out.push_back(w); out.push_back(b);
        }
    };
    carve(ctx->ff, t->a_ff);
    if (ctx->cfg.back_model) carve(ctx->back, t->a_bk);
    if (ctx->C > 0) carve(ctx->cp, t->a_cp);
    t->a_mx.n = t->a_mn.n = ctx->D;
    t->a_mx.m = q; q += ctx->D; t->a_mx.v = q; q += ctx->D;
    t->a_mn.m = q; q += ctx->D; t->a_mn.v = q; q += ctx->D;
    return CADM_OK;
}

static int ensure_workspace(cadm_ctx* ctx, int B) {
    TrainState* t = ctx->train;
    if (B <= t->B && t->ws) return CADM_OK;
    if (t->ws) { (void)hipFree(t->ws); t->ws = nullptr; }
    const size_t R = (size_t)ctx->E * B;
    const int NH = ctx->NH, HID = ctx->HID, D = ctx->D, K0 = ctx->K0;
    const int ncp = ctx->C > 0 ? ctx->cfg.n_cp_hidden : 0;
    const int cpin = (ctx->D + ctx->A) * ctx->cfg.history_length;
    int maxw = HID > K0 ? HID : K0;
    for (int l = 0; l < ncp; ++l) maxw = ctx->cfg.cp_hidden[l] > maxw ? ctx->cfg.cp_hidden[l] : maxw;
    size_t total = 0;
    auto need = [&](size_t n) { size_t o = total; total += (n + 63) & ~(size_t)63; return o; };
    const size_t oXff = need(R * K0), oXbk = need(R * K0), oXcp = need(R * (cpin > 0 ? cpin : 1));
    const size_t odCtx = need(R * (ctx->C > 0 ? ctx->C : 1)), octxo = need(R * (ctx->C > 0 ? ctx->C : 1));
    std::vector<size_t> oz_ff(NH), oh_ff(NH), oz_bk(NH), oh_bk(NH), oz_cp(ncp), oh_cp(ncp);
    for (int l = 0; l < NH; ++l) { oz_ff[l] = need(R * HID); oh_ff[l] = need(R * HID); oz_bk[l] = need(R * HID); oh_bk[l] = need(R * HID); }
    for (int l = 0; l < ncp; ++l) { oz_cp[l] = need(R * ctx->cfg.cp_hidden[l]); oh_cp[l] = need(R * ctx->cfg.cp_hidden[l]); }
    const size_t omu = need(R * D), olv = need(R * D), obmu = need(R * D), oblv = need(R * D);
    const size_t odA = need(R * maxw), odB = need(R * maxw), odMu = need(R * D), odLv = need(R * D), odBmu = need(R * D);
    const size_t oterms = need(7 * R * D), ored = need(4 + 2 * (size_t)D + 8);
    CADM_CHECK_HIP(hipMalloc(&t->ws, total * sizeof(float)));
    t->ws_floats = total;
    float* w = t->ws;
    t->Xff = w + oXff; t->Xbk = w + oXbk; t->Xcp = w + oXcp; t->dCtx = w + odCtx; t->ctxo = w + octxo;
    t->ff.z.resize(NH); t->ff.h.resize(NH); t->bk.z.resize(NH); t->bk.h.resize(NH); t->cp.z.resize(ncp); t->cp.h.resize(ncp);
    for (int l = 0; l < NH; ++l) { t->ff.z[l] = w + oz_ff[l]; t->ff.h[l] = w + oh_ff[l]; t->bk.z[l] = w + oz_bk[l]; t->bk.h[l] = w + oh_bk[l]; }
    for (int l = 0; l < ncp; ++l) { t->cp.z[l] = w + oz_cp[l]; t->cp.h[l] = w + oh_cp[l]; }
    t->ff.mu = w + omu; t->ff.lv = w + olv; t->bk.mu = w + obmu; t->bk.lv = w + oblv;
    t->dA = w + odA; t->dBuf = w + odB; t->dMu = w + odMu; t->dLv = w + odLv; t->dBmu = w + odBmu;
    t->terms = w + oterms; t->red = w + ored;
    t->B = B;
    return CADM_OK;
}

static int ensure_state(cadm_ctx* ctx) {
    if (ctx->train) return CADM_OK;
    ctx->train = new (std::nothrow) TrainState();
    if (!ctx->train) { cadm_set_error("out of host memory"); return CADM_ENOMEM; }
    return CADM_OK;
}

extern "C" int cadm_train_configure(cadm_ctx* ctx, const cadm_train_hparams* hp, int max_batch) {
    CADM_REQUIRE(ctx && hp, "cadm_train_configure: null argument");
    for (auto& d : ctx->ff) CADM_REQUIRE(d.W && d.b, "cadm_train_configure: ff_model weights not registered");
    if (ctx->cfg.back_model) for (auto& d : ctx->back) CADM_REQUIRE(d.W && d.b, "cadm_train_configure: backward_model weights not registered");
    if (ctx->C > 0) for (auto& d : ctx->cp) CADM_REQUIRE(d.W && d.b, "cadm_train_configure: context_model weights not registered");
    int rc0 = ensure_state(ctx);
    if (rc0) return rc0;
    if (!ctx->train->adam_buf && (rc0 = alloc_adam(ctx))) return rc0;
    ctx->train->hp = *hp;
    ctx->train->configured = true;
    if (max_batch > 0) return ensure_workspace(ctx, max_batch);
    return CADM_OK;
}

extern "C" int cadm_train_reset(cadm_ctx* ctx, void* stream) {
    CADM_REQUIRE(ctx && ctx->train, "cadm_train_reset: training not configured");
    TrainState* t = ctx->train;
    size_t total = 0;
    for (auto* v : {&t->a_ff, &t->a_bk, &t->a_cp}) for (auto& s : *v) total += 2 * s.n;
    total += 4 * (size_t)ctx->D;
    CADM_CHECK_HIP(hipMemsetAsync(t->adam_buf, 0, total * sizeof(float), (hipStream_t)stream));
    t->step = 0;
    return CADM_OK;
}

namespace {

template <int MI>
int launch_gemm_t(const GemmP& p, hipStream_t s) {
    constexpr int TM = 32 * MI;
    dim3 grid((p.N + TN - 1) / TN, (p.M + TM - 1) / TM, p.E);
    const size_t lds = (size_t)NSLAB * TK * ((TM + 4) + LDB) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        CADM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<MI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_kernel<MI>, grid, dim3(256), lds, s, p);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

int launch_gemm(const GemmP& p, hipStream_t s) {
    // 32-row tiles while they still leave the 256 CUs under-subscribed (more, shorter latency chains), else 64-row tiles
    const long wg32 = (long)((p.N + TN - 1) / TN) * ((p.M + 31) / 32) * p.E;
    return wg32 <= 1024 ? launch_gemm_t<1>(p, s) : launch_gemm_t<2>(p, s);
}

// H = act(X W + b): X [E,B,ldx] (first K columns used), W [E,K,N]
int fwd_layer(cadm_ctx* ctx, int B, const float* X, int ldx, const DenseRef& L, int act, float* Z, float* H, int ldo,
              hipStream_t s) {
    GemmP p{};
    p.A = X; p.sAe = (long)B * ldx; p.sam = ldx; p.sak = 1; p.a_mcontig = 0;
    p.B = L.W; p.sBe = (long)L.din * L.dout; p.sbk = L.dout; p.sbn = 1;
    p.M = B; p.N = L.dout; p.K = L.din; p.E = ctx->E;
    p.mode = MODE_FWD; p.act = act;
    p.bias = L.b; p.sbe = L.dout;
    p.Zout = Z; p.Hout = H; p.ldo = ldo; p.sOe = (long)B * ldo;
    return launch_gemm(p, s);
}

// dXsub = (dZ W[k0:k0+kn, :]^T) * act'(zprev)
int dx_layer(cadm_ctx* ctx, int B, const float* dZ, const DenseRef& L, int k0, int kn, const float* Zprev, int act,
             float* DX, int lddx, int accumulate, hipStream_t s) {
    GemmP p{};
    p.A = dZ; p.sAe = (long)B * L.dout; p.sam = L.dout; p.sak = 1; p.a_mcontig = 0;
    p.B = L.W + (long)k0 * L.dout; p.sBe = (long)L.din * L.dout; p.sbk = 1; p.sbn = L.dout;   // B(k=n_out, n=k_in) = W[k_in][n_out]
    p.M = B; p.N = kn; p.K = L.dout; p.E = ctx->E;
    p.mode = MODE_DX; p.act = act;
    p.Zprev = Zprev; p.ldzp = kn; p.sZpe = (long)B * kn;
    p.DXout = DX; p.lddx = lddx; p.sDXe = (long)B * lddx; p.accumulate = accumulate;
    return launch_gemm(p, s);
}

// W <- Adam(W, X^T dZ + wdc W), b <- Adam(b, colsum dZ)
int dw_layer(cadm_ctx* ctx, int B, const float* X, int ldx, const float* dZ, const DenseRef& L, float wdc, AdamSlot& aw,
             AdamSlot& ab, float lr_t, hipStream_t s) {
    const cadm_train_hparams& hp = ctx->train->hp;
    GemmP p{};
    p.A = X; p.sAe = (long)B * ldx; p.sam = 1; p.sak = ldx; p.a_mcontig = 1;      // A(m=k_in, k=b) = X[b][k_in]
    p.B = dZ; p.sBe = (long)B * L.dout; p.sbk = L.dout; p.sbn = 1;
    p.M = L.din; p.N = L.dout; p.K = B; p.E = ctx->E;
    p.mode = MODE_DW;
    p.W = L.W; p.Mw = aw.m; p.Vw = aw.v; p.ldw = L.dout; p.sWe = (long)L.din * L.dout;
    p.bW = L.b; p.bM = ab.m; p.bV = ab.v; p.sbWe = L.dout;
    p.wdc = wdc; p.lr_t = lr_t; p.b1 = hp.beta1; p.b2 = hp.beta2; p.eps = hp.epsilon;
    return launch_gemm(p, s);
}

int adam_elem(float* w, AdamSlot& a, const float* gsrc, float gscale, float gconst, float wdc, const cadm_train_hparams& hp,
              float lr_t, hipStream_t s) {
    hipLaunchKernelGGL(adam_elem_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, w, a.m, a.v, gsrc, gscale,
                       gconst, wdc, (long)a.n, lr_t, hp.beta1, hp.beta2, hp.epsilon);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

}  // namespace

namespace {
// forward of the context / forward (/ backward) nets on one [E,B,.] batch into the workspace
int forward_nets(cadm_ctx* ctx, const float* obs, const float* act, const float* obs_next, const float* cp_obs,
                 const float* cp_act, int B, bool has_back, hipStream_t s) {
    TrainState* t = ctx->train;
    const bool has_cp = ctx->C > 0, det = ctx->cfg.deterministic != 0;
    const int E = ctx->E, NH = ctx->NH, HID = ctx->HID, D = ctx->D, K0 = ctx->K0, C = ctx->C, PA = ctx->P + ctx->A;
    const int ncp = has_cp ? ctx->cfg.n_cp_hidden : 0;
    const int cpin = (ctx->D + ctx->A) * ctx->cfg.history_length;
    const long R = (long)E * B;
    int rc;
    AsmP ap{};
    ap.obs = obs; ap.obs_next = obs_next; ap.act = act; ap.cp_obs = cp_obs; ap.cp_act = cp_act;
    ap.obs_mean = ctx->st.obs_mean; ap.obs_std = ctx->st.obs_std; ap.act_mean = ctx->st.act_mean; ap.act_std = ctx->st.act_std;
    ap.cp_obs_mean = ctx->st.cp_obs_mean; ap.cp_obs_std = ctx->st.cp_obs_std;
    ap.cp_act_mean = ctx->st.cp_act_mean; ap.cp_act_std = ctx->st.cp_act_std;
    ap.Xff = t->Xff; ap.Xbk = t->Xbk; ap.Xcp = t->Xcp;
    ap.rows = (int)R; ap.D = D; ap.A = ctx->A; ap.P = ctx->P; ap.K0 = K0;
    ap.ncpo = D * ctx->cfg.history_length; ap.ncpa = ctx->A * ctx->cfg.history_length;
    ap.env = ctx->cfg.env_kind; ap.has_back = has_back; ap.has_cp = has_cp;
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)R), dim3(64), 0, s, ap);
    CADM_CHECK_HIP(hipGetLastError());
    if (has_cp) {
        const float* x = t->Xcp;
        int ldx = cpin;
        for (int l = 0; l < ncp; ++l) {
            if ((rc = fwd_layer(ctx, B, x, ldx, ctx->cp[l], ACT_RELU, t->cp.z[l], t->cp.h[l], ctx->cp[l].dout, s))) return rc;
            x = t->cp.h[l]; ldx = ctx->cp[l].dout;
        }
        // context vector straight into the ctx columns of the forward net's input; copied for the backward net
        if ((rc = fwd_layer(ctx, B, x, ldx, ctx->cp[ncp], ACT_NONE, nullptr, t->Xff + PA, K0, s))) return rc;
        if (has_back) {
            hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((R * C + 255) / 256)), dim3(256), 0, s, t->Xff + PA, (long)K0,
                               t->Xbk + PA, (long)K0, C, R);
            CADM_CHECK_HIP(hipGetLastError());
        }
    }
    auto net_fwd = [&](const std::vector<DenseRef>& net, const float* X, NetBufs& nb, bool want_lv) -> int {
        const float* x = X;
        int ldx = K0;
        for (int l = 0; l < NH; ++l) {
            int r = fwd_layer(ctx, B, x, ldx, net[l], ACT_SWISH, nb.z[l], nb.h[l], HID, s);
            if (r) return r;
            x = nb.h[l]; ldx = HID;
        }
        int r = fwd_layer(ctx, B, x, HID, net[NH], ACT_NONE, nullptr, nb.mu, D, s);
        if (r) return r;
        if (want_lv) r = fwd_layer(ctx, B, x, HID, net[NH + 1], ACT_NONE, nullptr, nb.lv, D, s);
        return r;
    };
    if ((rc = net_fwd(ctx->ff, t->Xff, t->ff, !det))) return rc;
    if (has_back && (rc = net_fwd(ctx->back, t->Xbk, t->bk, false))) return rc;

    return CADM_OK;
}

__global__ void clamp_logvar_kernel(const float* lv, const float* maxlv, const float* minlv, float* out, long n, int D) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % D);
    const float u = maxlv[d] - tf_softplus(maxlv[d] - lv[i]);      // core/utils.py:356
    out[i] = minlv[d] + tf_softplus(u - minlv[d]);                 // core/utils.py:357
}
}  // namespace

extern "C" int cadm_train_step(cadm_ctx* ctx, const float* obs, const float* act, const float* delta,
                               const float* obs_next, const float* back_delta, const float* cp_obs,
                               const float* cp_act, int B, int train, float* losses_out, void* stream) {
    CADM_REQUIRE(ctx && obs && act && delta && losses_out && B > 0, "cadm_train_step: bad arguments");
    CADM_REQUIRE(ctx->train && ctx->train->configured, "cadm_train_step: call cadm_train_configure first");
    CADM_REQUIRE(ctx->st.set, "cadm_train_step: normalisation stats not set");
    const bool has_back = ctx->cfg.back_model != 0, has_cp = ctx->C > 0, det = ctx->cfg.deterministic != 0;
    CADM_REQUIRE(!has_back || (obs_next && back_delta), "cadm_train_step: obs_next / back_delta required (backward model)");
    CADM_REQUIRE(!has_cp || (cp_obs && cp_act), "cadm_train_step: cp_obs / cp_act required (context model)");
    CADM_REQUIRE(ctx->ff_maxlv && ctx->ff_minlv, "cadm_train_step: logvar bounds not registered");
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_workspace(ctx, B);
    if (rc) return rc;
    TrainState* t = ctx->train;
    const cadm_train_hparams& hp = t->hp;
    const int E = ctx->E, NH = ctx->NH, HID = ctx->HID, D = ctx->D, K0 = ctx->K0, C = ctx->C, PA = ctx->P + ctx->A;
    const int ncp = has_cp ? ctx->cfg.n_cp_hidden : 0;
    const int cpin = (ctx->D + ctx->A) * ctx->cfg.history_length;
    const long R = (long)E * B;

    if ((rc = forward_nets(ctx, obs, act, obs_next, cp_obs, cp_act, B, has_back, s))) return rc;

    // ---- losses + head gradients ----
    LossP lp{};
    lp.mu = t->ff.mu; lp.lv = t->ff.lv; lp.bmu = t->bk.mu; lp.delta = delta; lp.back_delta = back_delta;
    lp.dmean = ctx->st.delta_mean; lp.dstd = ctx->st.delta_std; lp.bdmean = ctx->st.back_delta_mean; lp.bdstd = ctx->st.back_delta_std;
    lp.maxlv = ctx->ff_maxlv; lp.minlv = ctx->ff_minlv;
    lp.dMu = t->dMu; lp.dLv = t->dLv; lp.dBmu = t->dBmu; lp.terms = t->terms;
    lp.n = R * D; lp.D = D; lp.B = B; lp.det = det; lp.has_back = has_back; lp.back_coeff = hp.back_coeff;
    hipLaunchKernelGGL(loss_kernel, dim3((unsigned)((lp.n + 255) / 256)), dim3(256), 0, s, lp);
    hipLaunchKernelGGL(reduce_kernel, dim3(4 + 2 * D), dim3(256), 0, s, t->terms, lp.n, D, t->red);
    hipLaunchKernelGGL(finalize_loss_kernel, dim3(1), dim3(1), 0, s, t->red, (int)det, hp.back_coeff, (int)has_back, losses_out);
    CADM_CHECK_HIP(hipGetLastError());
    if (!train) return CADM_OK;

    // ---- backward + TF1 Adam (lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)) ----
    t->step += 1;
    const float lr_t = (float)(hp.learning_rate * sqrt(1.0 - pow((double)hp.beta2, (double)t->step)) /
                               (1.0 - pow((double)hp.beta1, (double)t->step)));
    const float coeff = hp.weight_decay_coeff;
    auto wd_dyn = [&](int l) { return coeff * (l < NH ? hp.weight_decays[l] : hp.weight_decays[NH]); };
    if (has_cp) CADM_CHECK_HIP(hipMemsetAsync(t->dCtx, 0, (size_t)R * C * sizeof(float), s));

    auto backward_net = [&](std::vector<DenseRef>& net, const float* X, NetBufs& nb, std::vector<AdamSlot>& ad,
                            const float* dMu, const float* dLv, bool lv_l2_only) -> int {
        int r;
        float* dcur = t->dA;     // gradient w.r.t. the pre-activation of hidden layer NH-1
        float* dnext = t->dBuf;
        // d h_{NH-1} = dMu W_mu^T (+ dLv W_lv^T), then * swish'(z_{NH-1})
        if ((r = dx_layer(ctx, B, dMu, net[NH], 0, HID, nullptr, ACT_NONE, dcur, HID, 0, s))) return r;
        if (dLv && (r = dx_layer(ctx, B, dLv, net[NH + 1], 0, HID, nullptr, ACT_NONE, dcur, HID, 1, s))) return r;
        hipLaunchKernelGGL(mul_actgrad_kernel, dim3((unsigned)((R * HID + 255) / 256)), dim3(256), 0, s, dcur, nb.z[NH - 1],
                           R * HID, (int)ACT_SWISH);
        CADM_CHECK_HIP(hipGetLastError());
        // head weights
        if ((r = dw_layer(ctx, B, nb.h[NH - 1], HID, dMu, net[NH], wd_dyn(NH), ad[2 * NH], ad[2 * NH + 1], lr_t, s))) return r;
        if (dLv) {
            if ((r = dw_layer(ctx, B, nb.h[NH - 1], HID, dLv, net[NH + 1], wd_dyn(NH + 1), ad[2 * (NH + 1)], ad[2 * (NH + 1) + 1], lr_t, s))) return r;
        } else if (lv_l2_only) {
            // output_logvar is outside the data path (deterministic / backward net): its weight only sees the L2 term,
            // its bias has no gradient at all and is skipped like TF does
            if ((r = adam_elem(net[NH + 1].W, ad[2 * (NH + 1)], nullptr, 0.f, 0.f, wd_dyn(NH + 1), hp, lr_t, s))) return r;
        }
        for (int l = NH - 1; l >= 0; --l) {
            const float* xin = l == 0 ? X : nb.h[l - 1];
            const int ldx = l == 0 ? K0 : HID;
            if (l > 0) {
                if ((r = dx_layer(ctx, B, dcur, net[l], 0, HID, nb.z[l - 1], ACT_SWISH, dnext, HID, 0, s))) return r;
            } else if (has_cp) {
                // only the context columns of the input carry a gradient; both nets accumulate into dCtx
                if ((r = dx_layer(ctx, B, dcur, net[0], PA, C, nullptr, ACT_NONE, t->dCtx, C, 1, s))) return r;
            }
            if ((r = dw_layer(ctx, B, xin, ldx, dcur, net[l], wd_dyn(l), ad[2 * l], ad[2 * l + 1], lr_t, s))) return r;
            float* tmp = dcur; dcur = dnext; dnext = tmp;
        }
        return CADM_OK;
    };
    if ((rc = backward_net(ctx->ff, t->Xff, t->ff, t->a_ff, t->dMu, det ? nullptr : t->dLv, det))) return rc;
    if (has_back && (rc = backward_net(ctx->back, t->Xbk, t->bk, t->a_bk, t->dBmu, nullptr, true))) return rc;
    if (!det) {   // max/min_logvar of the forward net: data term + 0.01 regulariser (dynamics.py:308)
        if ((rc = adam_elem(ctx->ff_maxlv, t->a_mx, t->red + 4, 1.0f, 0.01f, 0.0f, hp, lr_t, s))) return rc;
        if ((rc = adam_elem(ctx->ff_minlv, t->a_mn, t->red + 4 + D, 1.0f, -0.01f, 0.0f, hp, lr_t, s))) return rc;
    }
    if (has_cp) {
        auto wd_cp = [&](int l) { return coeff * (l < ncp ? hp.context_weight_decays[l] : hp.context_weight_decays[ncp]); };
        float* dcur = t->dCtx;      // gradient w.r.t. the (linear) context output
        float* bufs[2] = {t->dA, t->dBuf};
        int flip = 0;
        for (int l = ncp; l >= 0; --l) {
            const float* xin = l == 0 ? t->Xcp : t->cp.h[l - 1];
            const int ldx = l == 0 ? cpin : ctx->cp[l - 1].dout;
            float* dprev = nullptr;
            if (l > 0) {
                dprev = bufs[flip]; flip ^= 1;
                if ((rc = dx_layer(ctx, B, dcur, ctx->cp[l], 0, ctx->cp[l].din, t->cp.z[l - 1], ACT_RELU, dprev, ctx->cp[l].din, 0, s))) return rc;
            }
            if ((rc = dw_layer(ctx, B, xin, ldx, dcur, ctx->cp[l], wd_cp(l), t->a_cp[2 * l], t->a_cp[2 * l + 1], lr_t, s))) return rc;
            dcur = dprev;
        }
    }
    ctx->packed = false;   // planner streams are stale until cadm_repack
    return CADM_OK;
}

// One-step prediction heads of every member on an [E,B,.] batch (the vanilla reference's `_get_pred`,
// mlp_ensemble_cem_dynamics.py:185-189: [mlp.mu, mlp.logvar]): normalised mean and clamped log-variance.
extern "C" int cadm_predict(cadm_ctx* ctx, const float* obs, const float* act, const float* cp_obs, const float* cp_act,
                            int B, float* mu_out, float* logvar_out, void* stream) {
    CADM_REQUIRE(ctx && obs && act && mu_out && B > 0, "cadm_predict: bad arguments");
    CADM_REQUIRE(ctx->st.set, "cadm_predict: normalisation stats not set");
    CADM_REQUIRE(ctx->C == 0 || (cp_obs && cp_act), "cadm_predict: cp_obs / cp_act required (context model)");
    for (auto& d : ctx->ff) CADM_REQUIRE(d.W && d.b, "cadm_predict: ff_model weights not registered");
    if (ctx->C > 0) for (auto& d : ctx->cp) CADM_REQUIRE(d.W && d.b, "cadm_predict: context_model weights not registered");
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    if ((rc = ensure_workspace(ctx, B))) return rc;
    if ((rc = forward_nets(ctx, obs, act, nullptr, cp_obs, cp_act, B, false, s))) return rc;
    TrainState* t = ctx->train;
    const long n = (long)ctx->E * B * ctx->D;
    CADM_CHECK_HIP(hipMemcpyAsync(mu_out, t->ff.mu, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (logvar_out) {
        CADM_REQUIRE(!ctx->cfg.deterministic, "cadm_predict: a deterministic model has no log-variance head output");
        hipLaunchKernelGGL(clamp_logvar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, t->ff.lv, ctx->ff_maxlv,
                           ctx->ff_minlv, logvar_out, n, ctx->D);
        CADM_CHECK_HIP(hipGetLastError());
    }
    return CADM_OK;
}
