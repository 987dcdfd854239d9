// Context encoder inference (reference core/utils.py:401-406 + :614-617; get_context_pred,
// dynamics.py:369-380): normalise the history window, 3 ReLU layers, linear output.
// Two kernels: `context_kernel` for the planner's call (once per get_action on E*m rows, 5 at m = 1, ~1 MFLOP: a latency
// kernel, one workgroup per (member, row), K split across its waves and reduced through LDS) and `context_batched_kernel` for
// many histories per call (get_context_pred from the PPO consumer, SURVEY 8f-3: a GEMM chain on the fp32 matrix pipe).
#include <string.h>

#include "common.h"

#define CP_MAX_WIDTH 1024
// (CADM_CONTEXT_BATCHED_MIN_ROWS, common.h: below it, one latency-tuned workgroup per row -- the planner's m = 1..10)

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
// the encoder on ONE (member, history) row: `in_obs(i)` / `in_act(i)` read element i of the row's raw history
template <class InObs, class InAct>
__device__ __forceinline__ void context_row(const CpArgs& a, int e, int mi, InObs&& in_obs, InAct&& in_act) {
    __shared__ float xa[CP_MAX_WIDTH];
    __shared__ float xb[CP_MAX_WIDTH];
    __shared__ float red[CP_KQ][256];
    const int tid = threadIdx.x;
    for (int i = tid; i < a.n_obs; i += CP_THREADS)
        xa[i] = (in_obs(i) - a.obs_mean[i]) / (a.obs_std[i] + 1e-10f);          // :403
    for (int i = tid; i < a.n_act; i += CP_THREADS)
        xa[a.n_obs + i] = (in_act(i) - a.act_mean[i]) / (a.act_std[i] + 1e-10f);  // :404
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

__global__ __launch_bounds__(CP_THREADS) void context_kernel(const CpArgs a) {
    const int e = blockIdx.x / a.m, mi = blockIdx.x % a.m;
    const size_t in_row = a.bs ? ((size_t)e * a.m + mi) : (size_t)mi;   // tile(.., [E,1,1]) unless already [E,m,.]
    context_row(a, e, mi, [&](int i) { return a.cp_obs[in_row * a.n_obs + i]; }, [&](int i) { return a.cp_act[in_row * a.n_act + i]; });
}

// Head of a staged planner call (cadm_cem_plan_staged), one launch instead of three (ingest, context encoder, candidates of CEM
// iteration 0: 4.8 us of sampling and two dependent-launch boundaries of a 0.85 ms get_action): the call's inputs arrive as kernel
// arguments; workgroups [0, E m) run the encoder on the histories IN the argument block, workgroup E m unpacks the block into the
// device block (what the later launches of the call read), the rest draw the candidates of iteration 0 from the block's mean / var.
struct PlanHeadArgs {
    int off_cp_obs, off_cp_act, off_mean, off_var, nfloats;
    float* dev_block;
    int ctx_blocks;                    // E * m (0: no context model)
    int n, H, A; float lb, ub; uint32_t seed, call;
    float* actions;
};
__global__ __launch_bounds__(CP_THREADS) void plan_head_kernel(const HeadBlock blk, const CpArgs a, const PlanHeadArgs x) {
    const int b = blockIdx.x;
    if (b < x.ctx_blocks) {
        const int e = b / a.m, mi = b % a.m;
        context_row(a, e, mi, [&](int i) { return blk.v[x.off_cp_obs + mi * a.n_obs + i]; }, [&](int i) { return blk.v[x.off_cp_act + mi * a.n_act + i]; });
    } else if (b == x.ctx_blocks) {
        for (int i = threadIdx.x; i < x.nfloats; i += CP_THREADS) x.dev_block[i] = blk.v[i];
    } else {
        const int HA = x.H * x.A;
        const size_t total = (size_t)a.m * x.n * HA;
        const size_t stride = (size_t)(gridDim.x - x.ctx_blocks - 1) * CP_THREADS;
        for (size_t L = (size_t)(b - x.ctx_blocks - 1) * CP_THREADS + threadIdx.x; L < total; L += stride) {
            const int ta = (int)(L % HA);
            const int mi = (int)(L / ((size_t)x.n * HA));
            x.actions[L] = sample_action(blk.v[x.off_mean + mi * HA + ta], blk.v[x.off_var + mi * HA + ta], nullptr, L, x.seed, x.call, 0, x.lb, x.ub);
        }
    }
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
#ifndef CPB_MAX_RT
#define CPB_MAX_RT 2                   // row tiles per workgroup tried first (developer builds: -DCPB_MAX_RT=1)
#endif
#define CPB_PF 6                       // k-steps of weight fragments in flight per wave
struct CpbArgs {
    CpArgs a;
    int rows_per_member;               // m
    int region0, region1;              // LDS floats of the two activation regions
    int in_stride;                     // row stride of the input tile (odd)
};

// One work item of a layer: 64 output units x NRT row tiles, the whole K.  Everything the k loop branches on is a template
// parameter (hipcc waits for ALL outstanding loads at every use once a uniform branch sits between a load and its use: the
// ring would be worth nothing), the loop body is straight-line code: one ring slot consumed, one refilled, NRT LDS reads,
// 4 NRT MFMAs.  K is walked in steps of 4 up to the next multiple of 4: the LDS tiles are zero-padded there and the weight
// rows clamped, so the tail needs no predicate.
template <int RT, int NRT, bool VEC, bool ROWMAJOR>
__device__ __forceinline__ void cpb_item(const float* __restrict__ W, const float* __restrict__ bias, int K, int N, bool relu, int in_stride,
                                         const float* xin, float* xout, float* gout, int grow0, int grows, int g, int rt0, int lane) {
    constexpr int R = 16 * RT;
    const int i = lane & 15, q = lane >> 4;
    const int u0 = g * 64 + 4 * i;                                // this lane's 4 consecutive units (tile t: unit u0 + t)
    const int nsteps = (K + 3) >> 2;
    // the bias is the accumulators' initial value (requested first, it arrives under the ring's first loads)
    floatx4 acc[4][NRT];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        floatx4 b4;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { const int unit = g * 64 + 4 * (4 * q + rr) + t; b4[rr] = bias[unit < N ? unit : N - 1]; }
#pragma unroll
        for (int r = 0; r < NRT; ++r) acc[t][r] = b4;
    }
    // A operand of k-step s: row k = 4 s + q of W, units u0 .. u0 + 3 (clamped into the matrix: the B operand is zero for
    // k >= K, and units >= N are never stored)
    const int uc = VEC ? (u0 < N ? u0 : N - 4) : 0;
    int ucs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ucs[t] = u0 + t < N ? u0 + t : N - 1;
    auto fetch = [&](int s) -> floatx4 {
        int k = 4 * s + q;
        k = k < K ? k : K - 1;
        const float* row = W + (size_t)k * N;
        if (VEC) return *reinterpret_cast<const floatx4*>(row + uc);
        floatx4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = row[ucs[t]];
        return v;
    };
    // B operand of k-step s for row tile r: X[row = 16 (rt0 + r) + i][k = 4 s + q]
    int boff[NRT];
#pragma unroll
    for (int r = 0; r < NRT; ++r) {
        const int row = 16 * (rt0 + r) + i;
        boff[r] = ROWMAJOR ? row * in_stride + q : q * R + (row ^ (RT == 2 ? 16 * (q & 1) : 0));      // (k & 1 == q & 1: k = 4 s + q)
    }
    const int bstep = ROWMAJOR ? 4 : 4 * R;
    auto bread = [&](float (&bv)[NRT], int s) {
#pragma unroll
        for (int r = 0; r < NRT; ++r) bv[r] = xin[boff[r] + s * bstep];
    };
    auto mfmas = [&](const floatx4& av, const float (&bv)[NRT]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < NRT; ++r) acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[r], acc[t][r], 0, 0, 0);
    };
    floatx4 ring[CPB_PF];
#pragma unroll
    for (int u = 0; u < CPB_PF; ++u) ring[u] = fetch(u < nsteps ? u : nsteps - 1);
    int s0 = 0;
    for (; s0 + CPB_PF <= nsteps; s0 += CPB_PF) {
        float bv[CPB_PF][NRT];
#pragma unroll
        for (int u = 0; u < CPB_PF; ++u) bread(bv[u], s0 + u);    // the B operands of the whole group first: one LDS latency per PF steps
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CPB_PF; ++u) {
            // consume the slot, THEN refill it, and keep hipcc from moving the refill in front of the MFMAs: a load hoisted
            // above the last use of its destination needs a second register and a copy at the loop's end -- behind a vmcnt(0)
            mfmas(ring[u], bv[u]);
            const int sn = s0 + u + CPB_PF;
            ring[u] = fetch(sn < nsteps ? sn : nsteps - 1);      // (past the end: a re-read of the last step, never consumed)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < CPB_PF - 1; ++u)                          // the last nsteps % PF steps are already in the ring
        if (s0 + u < nsteps) { float bv[NRT]; bread(bv, s0 + u); mfmas(ring[u], bv); }
    // D layout: lane (col = i -> data row, q), register rr -> A-row 4 q + rr -> unit g * 64 + 4 (4 q + rr) + t.  Units up to
    // the next multiple of 4 behind N are written as zeros (the next layer walks K in steps of 4).
    const int N4 = (N + 3) & ~3;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int unit = g * 64 + 4 * (4 * q + rr) + t;
#pragma unroll
            for (int r = 0; r < NRT; ++r) {
                float v = acc[t][r][rr];
                if (relu) v = fmaxf(v, 0.0f);                                        // ReLU hidden (layers.py:34), identity output
                const int row = 16 * (rt0 + r) + i;
                if (gout) { if (unit < N && row < grows) gout[(size_t)(grow0 + row) * N + unit] = v; }
                else if (unit < N4) xout[unit * R + (row ^ (RT == 2 ? 16 * (unit & 1) : 0))] = unit < N ? v : 0.0f;
            }
        }
}

template <int RT, bool ROWMAJOR>
__device__ __forceinline__ void cpb_layer(const float* __restrict__ W, const float* __restrict__ bias, int K, int N, bool relu, int in_stride,
                                          const float* xin, float* xout, float* gout, int grow0, int grows, int wave, int lane) {
    const int NG = (N + 63) >> 6;                                 // groups of 64 output units
    const bool vec = (N & 3) == 0;
    if (RT == 1 || NG >= CPB_WAVES) {                             // a wave takes whole groups, all row tiles (weights reused RT times)
        for (int g = wave; g < NG; g += CPB_WAVES) {
            if (vec) cpb_item<RT, RT, true, ROWMAJOR>(W, bias, K, N, relu, in_stride, xin, xout, gout, grow0, grows, g, 0, lane);
            else cpb_item<RT, RT, false, ROWMAJOR>(W, bias, K, N, relu, in_stride, xin, xout, gout, grow0, grows, g, 0, lane);
        }
    } else {                                                      // few groups: the row tiles of a group go to different waves
        for (int item = wave; item < NG * RT; item += CPB_WAVES) {
            const int g = item % NG, rt0 = item / NG;
            if (vec) cpb_item<RT, 1, true, ROWMAJOR>(W, bias, K, N, relu, in_stride, xin, xout, gout, grow0, grows, g, rt0, lane);
            else cpb_item<RT, 1, false, ROWMAJOR>(W, bias, K, N, relu, in_stride, xin, xout, gout, grow0, grows, g, rt0, lane);
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
    // input tile [row][k], normalised on the way in (core/utils.py:403-404); rows past the batch and the columns up to the next
    // multiple of 4 behind K0 are zeros.  A thread takes column k = tid of EVERY row (K0 <= CPB_THREADS: the reference's 240) or
    // walks (row, k) pairs; the loads of a thread are independent and issued together.
    const int K0 = a.dims[0], ist = p.in_stride, K04 = (K0 + 3) & ~3;
    for (int kb = 0; kb < K04; kb += CPB_THREADS) {
        const int k = kb + tid;
        const bool kin = k < K0, isobs = k < a.n_obs;
        const int kk = kin ? (isobs ? k : k - a.n_obs) : 0;
        const float* src = isobs ? a.cp_obs : a.cp_act;
        const int ldsrc = isobs ? a.n_obs : a.n_act;
        const float mu = (isobs ? a.obs_mean : a.act_mean)[kk], sd = (isobs ? a.obs_std : a.act_std)[kk] + 1e-10f;
        const size_t base = a.bs ? (size_t)e * a.m + row0 : (size_t)row0;                 // tile(.., [E,1,1]) unless already [E,m,.]
        if (k < K04) {
            float v[R];
#pragma unroll
            for (int row = 0; row < R; ++row) v[row] = src[(base + (row < rows ? row : rows - 1)) * ldsrc + kk];
#pragma unroll
            for (int row = 0; row < R; ++row) {
                const float x = (v[row] - mu) / sd;
                reg0[row * ist + k] = (kin && row < rows) ? x : 0.0f;
            }
        }
    }
    __syncthreads();
    const float* xin = reg0;
    for (int l = 0; l < a.nlayers; ++l) {
        const int K = a.dims[l], N = a.dims[l + 1];
        const bool last = l + 1 == a.nlayers;
        float* xout = (l & 1) ? reg0 : reg1;
        const float* W = a.W[l] + (size_t)e * K * N;
        const float* bl = a.b[l] + (size_t)e * N;
        float* gout = last ? a.out + (size_t)e * p.rows_per_member * N : nullptr;
        if (l == 0) cpb_layer<RT, true>(W, bl, K, N, !last, ist, xin, xout, gout, row0, rows, wave, lane);
        else cpb_layer<RT, false>(W, bl, K, N, !last, ist, xin, xout, gout, row0, rows, wave, lane);
        __syncthreads();
        xin = xout;
    }
}

static int launch_context_batched(cadm_ctx* ctx, const CpArgs& a, int m, hipStream_t s, bool* launched) {
    *launched = false;
    CpbArgs p{};
    p.a = a;
    p.rows_per_member = m;
    p.in_stride = ((a.dims[0] + 3) & ~3) | 1;      // odd: the coalesced input store and the [row][k] operand reads spread over the banks
    // Two row tiles per workgroup halve the weight stream per row, one tile doubles the workgroups: measured (E = 5, 1 x MI355X)
    // m = 256 / 1024 / 2048 / 4096 / 8192: RT 2 30.7 / 30.9 / 51.0 / 77.7 / 125.6 us, RT 1 23.7 / 33.9 / 42.7 / 77.5 / 146.8 us --
    // one tile wins while the two-tile grid would leave the chip between one and two workgroups per CU (uneven rounds).
    const int wg2 = ctx->E * ((m + 31) / 32);
    const int rt_first = (wg2 > ctx->n_cus && wg2 < 2 * ctx->n_cus + ctx->n_cus / 2) || wg2 <= ctx->n_cus / 4 ? 1 : CPB_MAX_RT;
    for (int RT = rt_first; RT >= 1; --RT) {
        const int R = 16 * RT;
        // region 0: the input tile, later the outputs of the odd layers; region 1: the outputs of the even layers
        size_t r0 = (size_t)R * p.in_stride, r1 = 0;
        for (int l = 0; l + 1 < a.nlayers; ++l) {
            const size_t need = (size_t)R * ((a.dims[l + 1] + 3) & ~3);
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

static int context_args(cadm_ctx* ctx, CpArgs& a) {
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
    a.obs_mean = ctx->st.cp_obs_mean; a.obs_std = ctx->st.cp_obs_std;
    a.act_mean = ctx->st.cp_act_mean; a.act_std = ctx->st.cp_act_std;
    a.n_obs = ctx->D * ctx->cfg.history_length;
    a.n_act = ctx->A * ctx->cfg.history_length;
    a.E = ctx->E;
    return CADM_OK;
}

int cadm_launch_plan_head(cadm_ctx* ctx, const float* host_block, int nfloats, const int32_t off[5], float* dev_block, int m, int n,
                          uint32_t seed, uint32_t call, float* ctx_out, float* actions_out, hipStream_t s) {
    CADM_REQUIRE(nfloats > 0 && nfloats <= CADM_HEAD_INGEST_MAX && off[3] >= 0 && off[4] >= 0, "plan head: bad ingest block");
    static_assert(sizeof(HeadBlock) + sizeof(CpArgs) + sizeof(PlanHeadArgs) <= 4096, "kernel argument block");
    CpArgs a{};
    PlanHeadArgs x{};
    if (ctx->C > 0) {
        CADM_REQUIRE(off[1] >= 0 && off[2] >= 0, "plan head: cp_obs / cp_act required for a context model");
        const int rc = context_args(ctx, a);
        if (rc) return rc;
        a.out = ctx_out;
        x.ctx_blocks = ctx->E * m;
    }
    a.m = m; a.bs = 0;
    x.off_cp_obs = off[1]; x.off_cp_act = off[2]; x.off_mean = off[3]; x.off_var = off[4]; x.nfloats = nfloats;
    x.dev_block = dev_block;
    x.n = n; x.H = ctx->H; x.A = ctx->A; x.lb = ctx->cfg.lower_bound; x.ub = ctx->cfg.upper_bound; x.seed = seed; x.call = call;
    x.actions = actions_out;
    const size_t total = (size_t)m * n * ctx->H * ctx->A;
    size_t sb = (total + CP_THREADS - 1) / CP_THREADS;
    if (sb > 2048) sb = 2048;
    HeadBlock blk;
    memcpy(blk.v, host_block, (size_t)nfloats * sizeof(float));
    hipLaunchKernelGGL(plan_head_kernel, dim3(x.ctx_blocks + 1 + (unsigned)sb), dim3(CP_THREADS), 0, s, blk, a, x);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

int cadm_launch_context(cadm_ctx* ctx, const float* cp_obs, const float* cp_act, int m, int bs, float* out,
                        hipStream_t s) {
    CpArgs a{};
    {
        const int rc = context_args(ctx, a);
        if (rc) return rc;
    }
    a.cp_obs = cp_obs; a.cp_act = cp_act;
    a.m = m; a.bs = bs; a.out = out;
    if (m >= CADM_CONTEXT_BATCHED_MIN_ROWS) {       // many histories per member: the GEMM-shaped path (weights reused across rows)
        bool launched = false;
        const int rc = launch_context_batched(ctx, a, m, s, &launched);
        if (rc || launched) return rc;
    }
    hipLaunchKernelGGL(context_kernel, dim3(ctx->E * m), dim3(CP_THREADS), 0, s, a);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}
