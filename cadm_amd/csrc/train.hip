// Training step of the ensemble (reference cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:269-317;
// vanilla twin mlp_ensemble_cem_dynamics.py:148-170): forward of the context / forward / backward
// nets on one [E,B,.] bootstrap batch, the losses, hand-written backward and TF1-semantics Adam.
//
// The batch is tiny (B = 256 rows x 5 members, < 3 GFLOP per step), so a step is bound by dependent-
// launch overhead (~4.5 us per kernel on this part) and by per-workgroup latency chains, not by
// throughput.  The step is therefore built from THREE fat kernels instead of one GEMM per layer:
//   chain_kernel (forward)   a workgroup owns 16 batch rows of one member of one net and walks the whole
//                            layer chain (context encoder -> dynamics net -> heads) with the activations
//                            resident in LDS; weights stream from L2 straight into the MFMA B operand
//                            through rotating register blocks (no LDS staging: a weight is used by exactly
//                            one wave; 8 waves, two per SIMD).  z / h of every layer are stored for the
//                            backward pass.
//   chain_kernel (backward)  same kernel, transposed weight indexing: dZ_{l-1} = (dZ_l W_l^T) * act'(z_{l-1})
//                            down the chain; once for the forward+backward nets, once for the context net.
//   dw_adam_kernel           every layer's W <- Adam(W, X^T dZ + c*wd*W), b <- Adam(b, colsum dZ) as ONE
//                            grouped launch over a tile table (the gradient never touches HBM).
// A chain is described by a small stage table in device memory (LOAD / GEMM stages, rebuilt only when
// a pointer changes).  Everything is launch-ordered on one stream: the backward chains (which read W)
// run before the grouped DW launch (which overwrites W).
#include <math.h>
#include <string.h>

#include "common.h"

namespace {

enum { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4 };      // stage-table codes (see dyn_act)

// Branch-free on the activation kind (uniform selects): v_exp_f32 / v_rcp_f32 sigmoid like the planner's swish_f.
__device__ __forceinline__ float sigmoid_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float act_fwd(int act, float z) {
    const float sw = z * sigmoid_fast(z);
    const float r = act == ACT_RELU ? fmaxf(z, 0.0f) : z;
    return act == ACT_SWISH ? sw : r;
}
__device__ __forceinline__ float act_bwd(int act, float z) {   // d act / d z
    const float sg = sigmoid_fast(z);
    const float sw = sg * (1.0f + z * (1.0f - sg));
    const float r = act == ACT_RELU ? (z > 0.0f ? 1.0f : 0.0f) : 1.0f;
    return act == ACT_SWISH ? sw : r;
}

// hidden nonlinearity of the dynamics nets (cadm_config.hidden_act, CADM_ACT_*) as a stage-table code
static int dyn_act(const cadm_ctx* ctx) {
    switch (ctx->cfg.hidden_act) {
        case CADM_ACT_RELU: return ACT_RELU;
        case CADM_ACT_TANH: return ACT_TANH;
        case CADM_ACT_SIGMOID: return ACT_SIGMOID;
        case CADM_ACT_NONE: return ACT_NONE;
        default: return ACT_SWISH;
    }
}

__device__ __forceinline__ void adam_update(float& w, float& m, float& v, float g, float lr_t, float b1, float b2,
                                            float eps) {
    // tf.compat.v1.train.AdamOptimizer (training_ops ApplyAdam): m,v EMA; w -= lr_t * m / (sqrt(v) + eps)
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    w -= lr_t * m / (sqrtf(v) + eps);
}

// ---------------------------------------------------------------------------------------------
// chain kernel: a list of LOAD / GEMM stages over a 16-row batch tile held in LDS
// ---------------------------------------------------------------------------------------------
enum { ST_LOAD = 0, ST_GEMM = 1 };
// Pointers read out of a stage table are generic to the compiler (flat_load: slower, and it ties the vector-memory
// counter to the LDS one); they all point to device memory, so say so.
typedef __attribute__((address_space(1))) const float* gcptr;
typedef __attribute__((address_space(1))) float* gptr;
__device__ __forceinline__ gcptr as_global(const float* p) { return (gcptr)p; }
__device__ __forceinline__ gptr as_global(float* p) { return (gptr)p; }
#define CH_ROWS 16
#define CH_THREADS 512        // 8 waves: two per SIMD, so one wave's LDS / load / scalar work overlaps the other's MFMAs
#define CH_GW 32              // output columns per wave and pass (NT <= 2 MFMA tiles)
#define CH_AD 3            // A fragments are read from LDS this many k-steps ahead
#define CH_MAXSTAGE 20

struct ChainPart {         // one product term of a GEMM stage: acc += src[16 x K] * Bop[K x N]
    const float* W;        // [E][.][ldw]
    long sWe;              // member stride (elements)
    int ldw;               // row stride of W
    int wt;                // 0: Bop(k,n) = W[k][n]   1: Bop(k,n) = W[row0 + n][k]  (backward: dZ W^T)
    int row0, K, src, pad;
};
struct ChainStage {
    int kind, N, nparts, act_d, act_o, dst, dk0, K;      // K, ld_in: LOAD only
    ChainPart part[2];
    const float *bias, *zprev, *g0, *g1;                 // g0 (+ g1): LOAD sources [E][B][ld_in]
    float *out0, *out1, *gsum;                           // out0: value before act_o, out1: after; gsum: LOAD echo
    int ldz, ldo, ld_in;
    int zpad;                                            // rows to zero behind the written range of dst: the consumer's k loops run
                                                         // whole 32-deep blocks without masking A (host: round32(width) - width)
};
#define CH_MAXPF 24
struct ChainArgs {
    const ChainStage* prog;
    int first[2], count[2];                              // stage range per chain (y)
    int B, bufsz;                                        // rows per member, floats per LDS activation buffer
    int E, ny, ntiles, G, ips;                           // work decomposition, see chain_kernel
    int npf;                                             // weight tensors to pull into this XCD's L2 up front
    const float* pf_ptr[CH_MAXPF]; int pf_n[CH_MAXPF];   // base, floats per member
    unsigned long long* tbuf;                            // cadm_debug_set_timing_buffer: clocks of member 0's first work item:
                                                         // [0..63] stage boundaries, [64 + 4 si ..] wave 0: group start, k loop end,
                                                         // epilogue end, barrier reached
};

typedef __attribute__((address_space(1))) const char* gcbytes;
typedef __attribute__((address_space(1))) const floatx4* gcptr4;
typedef float floatx2 __attribute__((ext_vector_type(2)));

// acc[j] += src(16 x K) * Bop(K x 16) for NT column tiles of this wave.  A comes from the LDS activation buffer
// (k-major, lds[k * 16 + m]), B straight from global memory through a register ring PF k-steps deep, addressed as
// (uniform base of the k-step) + (per-lane byte offset).  Three load shapes:
//   MODE 0  dword per (tile, k-step); tile j <-> column nb + 16 j + c; any N, K, either weight orientation
//   MODE 1  forward (Bop(k,n) = W[k][n]), N % 2 == 0: one b64 per k-step holds the 2 tiles, tile j <-> column nb + 2 c + j
//   MODE 2  transposed (Bop(k,n) = W[n][k]), K % 4 == 0: one b128 per (tile, 4 k-steps), k-step 4 t + i <-> k = 16 t + 4 kq + i
// The k loop is a compact rolled loop (this kernel runs each piece of code once per stage, so long unrolled
// stretches turn into instruction-cache misses that cost more than the MFMAs: ~300 cycles per k-step were measured
// with 32-step straight-line blocks).  B lives in four register blocks of PF k-steps that rotate: while block b is
// consumed, blocks b+1, b+2 and b+3 are in flight, which covers the ~2000-cycle L2 latency seen when a member's 32
// workgroups stream the same weights.
// hipcc cannot express that pipeline: its s_waitcnt placement gives up across the loop back edge and waits for
// every outstanding load at the first use, i.e. a lookahead of at most one block.  The weight loads are therefore
// issued from inline asm (invisible to the compiler's counters) and ordered with explicit `s_waitcnt vmcnt(n)`:
// loads return in issue order, so "at most n younger loads outstanding" is exact; older compiler-issued accesses
// (epilogue operands, stores of the previous stage) only make a wait conservative.  Every wait is followed by a
// sched_barrier so that no consumer can move above it, and nothing is left in flight when the function returns.
// A block that reaches past K is padded, never predicated: k-step indices are clamped to the last one, whose lanes
// beyond K use offsets clamped into the matrix, and A is zeroed for k >= K.
// (uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset): no vector address arithmetic per load
__device__ __forceinline__ void async_load(float& dst, gcbytes base, unsigned off) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void async_load(floatx2& dst, gcbytes base, unsigned off) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void async_load(floatx4& dst, gcbytes base, unsigned off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// Values read out of the LDS stage table are wave-uniform, but the compiler cannot know: make them scalar.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ gcbytes uni(gcbytes p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (gcbytes)(((unsigned long long)hi << 32) | lo);
}

template <int NT, int MODE>
__device__ __forceinline__ void chain_kloop(floatx4 (&acc)[NT], const ChainPart& pt, const float* src, int e, int nb, int N,
                                            int lane) {
    constexpr int PF = 8;                                  // k-steps per block
    constexpr int NR0 = MODE == 0 ? PF : 1, NR1 = MODE == 1 ? PF : 1, NR2 = MODE == 2 ? PF / 4 : 1;
    constexpr int LPB = MODE == 0 ? PF * NT : MODE == 1 ? PF : (PF / 4) * NT;      // loads per block
    const int c = lane & 15, kq = lane >> 4;
    const int K = uni(pt.K), ldw = uni(pt.ldw), wt = uni(pt.wt), prow0 = uni(pt.row0);
    const int nsteps = MODE == 2 ? ((K + 15) >> 4) * 4 : (K + 3) >> 2, last = nsteps - 1;
    const int nblk = (nsteps + PF - 1) / PF;
    gcbytes Wm = uni((gcbytes)(as_global(pt.W) + (long)e * pt.sWe));
    long step_bytes;                                        // per k-step (MODE 0/1) or per 4 k-steps (MODE 2)
    unsigned boff[NT], boffl[NT];                           // per-lane byte offsets: regular / last (clamped) step
    if (MODE == 0) {
        const int ks = wt ? 1 : ldw, ns = wt ? ldw : 1;
        const int kql = 4 * last + kq < K ? kq : K - 1 - 4 * last;
        step_bytes = 16L * ks;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            int n = nb + 16 * j + c;
            n = n < N ? n : N - 1;
            boff[j] = 4u * (unsigned)((prow0 + n) * ns + kq * ks);
            boffl[j] = 4u * (unsigned)((prow0 + n) * ns + kql * ks);
        }
    } else if (MODE == 1) {
        int n2 = nb + 2 * c;
        n2 = n2 < N ? n2 : N - 2;
        const int kql = 4 * last + kq < K ? kq : K - 1 - 4 * last;
        step_bytes = 16L * ldw;
        boff[0] = 4u * (unsigned)(n2 + kq * ldw);
        boffl[0] = 4u * (unsigned)(n2 + kql * ldw);
    } else {
        const int lastq = last >> 2;
        const int k4l = 16 * lastq + 4 * kq <= K - 4 ? 4 * kq : K - 4 - 16 * lastq;
        step_bytes = 64L;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            int n = nb + 16 * j + c;
            n = n < N ? n : N - 1;
            boff[j] = 4u * (unsigned)((prow0 + n) * ldw + 4 * kq);
            boffl[j] = 4u * (unsigned)((prow0 + n) * ldw + k4l);
        }
    }
    struct Blk {
        float r0[NR0][NT];
        floatx2 r1[NR1];
        floatx4 r2[NR2][NT];
    };
    Blk b0, b1, b2, b3;
    float ar[4];
    // loads of block bi (its PF k-steps) into a register block.  Blocks that end before the last k-step take the
    // fast path: a scalar base walks the k-steps, the per-lane offsets never change.
    auto issue_block = [&](Blk& blk, int bi) {
        const int s0 = bi * PF;
        constexpr int SPL = MODE == 2 ? 4 : 1;              // k-steps per load group
        const int lim = MODE == 2 ? last >> 2 : last;
        const int i0 = s0 / SPL;                            // first load-group index of the block
        if (i0 + PF / SPL - 1 < lim) {
            gcbytes base = Wm + i0 * step_bytes;
#pragma unroll
            for (int u = 0; u < PF; u += SPL) {
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) async_load(blk.r0[u][j], base, boff[j]);
                } else if (MODE == 1) {
                    async_load(blk.r1[u], base, boff[0]);
                } else {
#pragma unroll
                    for (int j = 0; j < NT; ++j) async_load(blk.r2[u >> 2][j], base, boff[j]);
                }
                base += step_bytes;
            }
        } else {
#pragma unroll
            for (int u = 0; u < PF; u += SPL) {
                const int sidx = i0 + u / SPL;
                const int sc = sidx < lim ? sidx : lim;
                gcbytes base = Wm + sc * step_bytes;
                const bool tail = sidx >= lim;
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) async_load(blk.r0[u][j], base, tail ? boffl[j] : boff[j]);
                } else if (MODE == 1) {
                    async_load(blk.r1[u], base, tail ? boffl[0] : boff[0]);
                } else {
#pragma unroll
                    for (int j = 0; j < NT; ++j) async_load(blk.r2[u >> 2][j], base, tail ? boffl[j] : boff[j]);
                }
            }
        }
    };
    // A fragment of k-step sidx: MODE 0/1 lds[(4 sidx + kq) * 16 + c] = lds[64 sidx + lane]; MODE 2 (sidx = 4 t + i)
    // lds[(16 t + 4 kq + i) * 16 + c].  Neither clamped nor masked: the producing stage zeroed rows K .. round32(K)
    // (ChainStage::zpad), and anything a lookahead read fetches beyond that is never used.  (A per-step
    // compare + select feeding the MFMA costs ~30 cycles per k-step that do not overlap with the matrix pipe.)
    const float* abase = MODE == 2 ? src + 64 * kq + c : src + lane;
    auto read_a = [&](int sidx) { return MODE == 2 ? abase[256 * (sidx >> 2) + 16 * (sidx & 3)] : abase[64 * sidx]; };
    auto compute_block = [&](const Blk& blk, int bi) {
        const int s0 = bi * PF;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            float b[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = MODE == 0 ? blk.r0[u][j] : MODE == 1 ? blk.r1[u][j] : blk.r2[u >> 2][j][u & 3];
            const float av = ar[u & 3];
            ar[(u + CH_AD) & 3] = read_a(s0 + u + CH_AD);
            __builtin_amdgcn_sched_barrier(0);     // keep the LDS read CH_AD steps ahead of its use
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // consume block bi from `cur`; `nxt` is the register block that was consumed last (free) and receives block bi + 3
    auto step = [&](Blk& cur, Blk& nxt, int bi) {
        const int rem = nblk - 1 - bi;             // blocks after this one
        if (rem >= 3) { issue_block(nxt, bi + 3); wait_vmcnt<3 * LPB>(); }
        else if (rem == 2) wait_vmcnt<2 * LPB>();
        else if (rem == 1) wait_vmcnt<LPB>();
        else wait_vmcnt<0>();
        compute_block(cur, bi);
    };
    __builtin_amdgcn_sched_barrier(0);
    issue_block(b0, 0);
    if (nblk > 1) issue_block(b1, 1);
    if (nblk > 2) issue_block(b2, 2);
#pragma unroll
    for (int u = 0; u < CH_AD; ++u) ar[u] = read_a(u);
#pragma unroll 1
    for (int bi = 0; bi < nblk; bi += 4) {
        step(b0, b3, bi);
        if (bi + 1 < nblk) step(b1, b0, bi + 1);
        if (bi + 2 < nblk) step(b2, b1, bi + 2);
        if (bi + 3 < nblk) step(b3, b2, bi + 3);
    }
}

// One GEMM stage for NT column tiles of this wave (VECN: tile j <-> column nb + 2 c + j, else nb + 16 j + c): all parts,
// then the epilogue.
__device__ __forceinline__ gcbytes uni_ptr(const float* p) { return uni((gcbytes)as_global(p)); }

template <int NT, bool VECN>
__device__ __forceinline__ void chain_group(const ChainStage& st, float* bufs, int bufsz, int e, int B, int row0, int nb,
                                            int lane, unsigned long long* dbg) {
    if (dbg) dbg[0] = __builtin_readcyclecounter();
    const int c = lane & 15, q = lane >> 4;
    // stage constants -> SGPRs (they come out of LDS): scalar address bases, uniform branches on the activation kinds
    const int N = uni(st.N), ldo = uni(st.ldo), ldz = uni(st.ldz), dk0 = uni(st.dk0), dsti = uni(st.dst);
    const int act_d = uni(st.act_d), act_o = uni(st.act_o), nparts = uni(st.nparts);
    gcbytes p_bias = uni_ptr(st.bias), p_z = uni_ptr(st.zprev), p_o0 = uni_ptr(st.out0), p_o1 = uni_ptr(st.out1);
    const bool has_z = p_z != nullptr, has_b = p_bias != nullptr, s0 = p_o0 != nullptr, s1 = p_o1 != nullptr;
    const long mrow = (long)e * B;                        // first row of this member in the [E][B][.] tensors
    floatx4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // epilogue operands requested before the k loop so that their latency hides under it (clamped, never predicated)
    float zp[NT][4], bv[NT];
    int ncl[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = VECN ? nb + 2 * c + j : nb + 16 * j + c;
        ncl[j] = n < N ? n : N - 1;
    }
    {
        gcbytes bb = has_b ? p_bias + (long)e * N * 4 : (gcbytes)as_global(st.part[0].W);
        gcbytes zb = has_z ? p_z + mrow * ldz * 4 : (gcbytes)as_global(st.part[0].W);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bv[j] = *reinterpret_cast<gcptr>(bb + (has_b ? 4u * (unsigned)ncl[j] : 0u));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = row0 + 4 * q + r;
                row = row < B ? row : B - 1;
                zp[j][r] = *reinterpret_cast<gcptr>(zb + (has_z ? 4u * (unsigned)(row * ldz + ncl[j]) : 0u));
            }
        }
    }
    for (int pi = 0; pi < nparts; ++pi) {
        const ChainPart& pt = st.part[pi];
        const float* src = bufs + pt.src * bufsz;
        if (VECN) chain_kloop<NT, 1>(acc, pt, src, e, nb, N, lane);
        else if (pt.wt && (pt.K & 3) == 0 && (pt.ldw & 3) == 0) chain_kloop<NT, 2>(acc, pt, src, e, nb, N, lane);
        else chain_kloop<NT, 0>(acc, pt, src, e, nb, N, lane);
    }
    if (dbg) dbg[1] = __builtin_readcyclecounter();
    // D layout: col = lane & 15 -> column slot c, row = (lane >> 4) * 4 + r -> batch row
    floatx4 v0[NT], v1[NT];                       // [tile][r]: before / after the output activation
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v0[j][r] = acc[j][r] + (has_b ? bv[j] : 0.0f);
    if (has_z) {
        if (act_d == ACT_SWISH) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = zp[j][r], sg = sigmoid_fast(z);
                    v0[j][r] *= sg * (1.0f + z * (1.0f - sg));
                }
        } else if (act_d == ACT_RELU) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v0[j][r] = zp[j][r] > 0.0f ? v0[j][r] : 0.0f;
        } else if (act_d == ACT_TANH) {          // 1 - tanh(z)^2 = 4 s (1 - s), s = sigmoid(2z)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float sg = sigmoid_fast(2.0f * zp[j][r]); v0[j][r] *= 4.0f * sg * (1.0f - sg); }
        } else if (act_d == ACT_SIGMOID) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float sg = sigmoid_fast(zp[j][r]); v0[j][r] *= sg * (1.0f - sg); }
        }
    }
    if (act_o == ACT_SWISH) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v1[j][r] = v0[j][r] * sigmoid_fast(v0[j][r]);
    } else if (act_o == ACT_RELU) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v1[j][r] = fmaxf(v0[j][r], 0.0f);
    } else if (act_o == ACT_TANH) {              // as the planner: 2 sigmoid(2z) - 1, odd series near 0 where that cancels
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = v0[j][r], x2 = x * x;
                const float ser = x * fmaf(x2, fmaf(x2, fmaf(x2, -0.05396825396825397f, 0.13333333333333333f), -0.3333333333333333f), 1.0f);
                v1[j][r] = fabsf(x) < 0.1f ? ser : fmaf(2.0f, sigmoid_fast(2.0f * x), -1.0f);
            }
    } else if (act_o == ACT_SIGMOID) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v1[j][r] = sigmoid_fast(v0[j][r]);
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) v1[j] = v0[j];
    }
    if (dsti >= 0) {
        float* dst = bufs + dsti * bufsz;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = VECN ? nb + 2 * c + j : nb + 16 * j + c;
            if (n < N) *reinterpret_cast<floatx4*>(dst + (dk0 + n) * CH_ROWS + 4 * q) = v1[j];
        }
    }
    // global stores: (uniform base of this member) + 32-bit byte offset (host checks E * B * ldo * 4 < 2^32)
    gcbytes b0 = p_o0 + mrow * ldo * 4, b1 = p_o1 + mrow * ldo * 4;
    typedef __attribute__((address_space(1))) float* gfp;
    typedef __attribute__((address_space(1))) floatx2* gf2p;
    if (VECN && (ldo & 1) == 0) {                 // a lane's 2 tiles are 2 adjacent columns: b64 stores
        const int n = nb + 2 * c;
        if (n < N) {
            // every store's data pair is built in ITS OWN registers before the first store is issued: a store reads its
            // data registers asynchronously, so re-using a pair makes hipcc wait (vmcnt) for the previous store to complete
            // -- eight full store round trips, ~2.6 k cycles per stage, were measured that way
            floatx2 p0[4], p1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p0[r] = floatx2{v0[0][r], v0[1 % NT][r]};
                p1[r] = floatx2{v1[0][r], v1[1 % NT][r]};
            }
            asm volatile("" : "+v"(p0[0]), "+v"(p0[1]), "+v"(p0[2]), "+v"(p0[3]), "+v"(p1[0]), "+v"(p1[1]), "+v"(p1[2]), "+v"(p1[3]));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * q + r;
                if (row >= B) continue;
                const unsigned o = 4u * (unsigned)(row * ldo + n);
                if (s0) *(gf2p)(b0 + o) = p0[r];
                if (s1) *(gf2p)(b1 + o) = p1[r];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = VECN ? nb + 2 * c + j : nb + 16 * j + c;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * q + r;
                if (row >= B) continue;
                const unsigned o = 4u * (unsigned)(row * ldo + n);
                if (s0) *(gfp)(b0 + o) = v0[j][r];
                if (s1) *(gfp)(b1 + o) = v1[j][r];
            }
        }
    }
    if (dbg) dbg[2] = __builtin_readcyclecounter();
}

// Work decomposition.  Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), and every XCD has
// its own L2, so the launch is 1-D and a member's work items (batch tile x chain) are all sent to the same
// G = 8 / E XCDs (E <= 8; one XCD per member for the 5-member ensemble: 32 items on its 32 CUs): a member's
// weights are then filled into exactly one L2 instead of eight.
__device__ __forceinline__ bool xcd_affine_item(int E, int G, int ips, int per, int& e, int& item) {
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    e = xcd / G + 8 * (j / ips);
    item = (j % ips) * G + xcd % G;
    return e < E && item < per;
}

__global__ __launch_bounds__(CH_THREADS) void chain_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float chain_smem[];
    ChainStage* const stg = reinterpret_cast<ChainStage*>(chain_smem);
    float* const bufs = chain_smem + (CH_MAXSTAGE * sizeof(ChainStage)) / sizeof(float);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int e, item;
    const int per = a.ntiles * a.ny;
    if (!xcd_affine_item(a.E, a.G, a.ips, per, e, item)) return;
    const int y = item / a.ntiles, row0 = (item - y * a.ntiles) * CH_ROWS, B = a.B;
    const int nst = a.count[y];
    {   // stage table of this chain -> LDS (one memory latency instead of one per stage)
        const int* g = reinterpret_cast<const int*>(a.prog + a.first[y]);
        int* l = reinterpret_cast<int*>(stg);
        const int nw = nst * (int)(sizeof(ChainStage) / sizeof(int));
        for (int i = tid; i < nw; i += CH_THREADS) l[i] = g[i];
    }
    // Pull this member's weights into the XCD's L2 now (one 128-byte line per load, spread over the member's
    // workgroups) so that the stages below start from L2, not from HBM.  The loads are fire-and-forget: inline
    // asm keeps them out of the compiler's vmcnt bookkeeping (older loads only make its waits conservative), and
    // `pf` stays live until the closing s_waitcnt so the destination register cannot be reused early.
    float pf = 0.0f;
    for (int i = 0; i < a.npf; ++i) {
        gcptr base = as_global(a.pf_ptr[i]) + (long)e * a.pf_n[i];
        const int nlines = (a.pf_n[i] + 31) >> 5;
        for (int line = (item / a.G) * CH_THREADS + tid; line < nlines; line += a.ips * CH_THREADS) {
            gcptr q = base + line * 32;
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf) : "v"(q) : "memory");
        }
    }
    __syncthreads();
    const bool timed = a.tbuf && item == 0 && e == 0 && tid == 0;
    if (timed) a.tbuf[0] = __builtin_readcyclecounter();
    for (int si = 0; si < nst; ++si) {
        const ChainStage& st = stg[si];
        if (st.zpad > 0) {      // nobody reads dst during this stage, its readers wait for the stage-end barrier
            float* zb = bufs + st.dst * a.bufsz + (st.dk0 + (st.kind == ST_LOAD ? st.K : st.N)) * CH_ROWS;
            for (int i = tid; i < st.zpad * CH_ROWS / 4; i += CH_THREADS) reinterpret_cast<floatx4*>(zb)[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
        if (st.kind == ST_LOAD) {
            float* dst = bufs + st.dst * a.bufsz;
            const int K = st.K;
            if (((K | st.ld_in | st.ldo) & 3) == 0) {          // b128 path: every thread's loads are in flight together
                const int K4 = K >> 2;
                for (int idx = tid; idx < CH_ROWS * K4; idx += CH_THREADS) {
                    const int m = idx / K4, k = (idx - m * K4) * 4;
                    const int row = row0 + m;
                    floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
                    if (row < B) {
                        const long o = ((long)e * B + row) * st.ld_in + k;
                        v = *reinterpret_cast<gcptr4>(as_global(st.g0) + o);
                        if (st.g1) v += *reinterpret_cast<gcptr4>(as_global(st.g1) + o);
                        if (st.gsum) *reinterpret_cast<__attribute__((address_space(1))) floatx4*>(as_global(st.gsum) + ((long)e * B + row) * st.ldo + k) = v;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[(st.dk0 + k + i) * CH_ROWS + m] = v[i];
                }
            } else {
                for (int idx = tid; idx < CH_ROWS * K; idx += CH_THREADS) {
                    const int m = idx / K, k = idx - m * K;
                    const int row = row0 + m;
                    float v = 0.0f;
                    if (row < B) {
                        const long o = ((long)e * B + row) * st.ld_in + k;
                        v = as_global(st.g0)[o];
                        if (st.g1) v += as_global(st.g1)[o];
                        if (st.gsum) as_global(st.gsum)[((long)e * B + row) * st.ldo + k] = v;
                    }
                    dst[(st.dk0 + k) * CH_ROWS + m] = v;
                }
            }
        } else {
            const int N = st.N;
            for (int nb = wave * CH_GW; nb < N; nb += CH_GW * (CH_THREADS / 64)) {
                const int nt = (N - nb + 15) >> 4;
                const ChainPart& p0 = st.part[0];
                unsigned long long* dbg = timed ? a.tbuf + 64 + si * 4 : nullptr;
                if (st.nparts == 1 && !p0.wt && (N & 1) == 0 && (p0.ldw & 1) == 0) chain_group<2, true>(st, bufs, a.bufsz, e, B, row0, nb, lane, dbg);
                else if (nt >= 2) chain_group<2, false>(st, bufs, a.bufsz, e, B, row0, nb, lane, dbg);
                else chain_group<1, false>(st, bufs, a.bufsz, e, B, row0, nb, lane, dbg);
            }
        }
        if (timed) a.tbuf[64 + si * 4 + 3] = __builtin_readcyclecounter();
        __syncthreads();
        if (timed) a.tbuf[si + 1] = __builtin_readcyclecounter();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(pf) : "memory");
}

// ---------------------------------------------------------------------------------------------
// grouped weight-gradient GEMM with the Adam update fused into its epilogue
// ---------------------------------------------------------------------------------------------
struct DwJob {                       // W[e] (M x N) <- Adam(W, X[e]^T dZ[e] + wdc W);  b[e] <- Adam(b, colsum dZ[e])
    const float *X, *dZ;             // X [E][B][ldx] (first M columns), dZ [E][B][N]
    float *W, *Mw, *Vw, *bW, *bM, *bV;
    int ldx, M, N, tile0;            // tile0: first workgroup (blockIdx.x) of this job
    float wdc;
    int tn;                          // column tiles
};
#define DW_MAXJOBS 20
struct DwArgs {
    DwJob job[DW_MAXJOBS];
    int njobs, B, tiles, E;          // tiles: work items per member
    float lr_t, b1, b2, eps;
};

#define TN 64
#define TM 48
#define TK 32
#define LDA (TM + 4)
#define LDB (TN + 4)
#define DW_NSLAB 1                   // slabs per K panel in flight (registers): one keeps the kernel at 120 VGPRs = 4 workgroups per CU

// One 48 x 64 tile of one job per workgroup, reduction over the batch: the whole step's ~925 tiles then fit the chip's
// 1024 workgroup slots (4 per CU) in ONE round (32 x 64 tiles needed 1330 = two rounds).  K is walked in 32-deep slabs:
// stash the slab's loads into LDS as they land, barrier, MFMA sweep; 4 waves side by side, each 48 x 16.  Occupancy beats
// panel depth here: 2-slab panels (156 VGPRs, 3 per CU) and register double-buffering (178 VGPRs) both measured slower.
__global__ __launch_bounds__(256) void dw_adam_kernel(const DwArgs a) {
    __shared__ __attribute__((aligned(16))) float dw_smem[DW_NSLAB * TK * (LDA + LDB)];
    float* const As = dw_smem;
    float* const Bs = dw_smem + DW_NSLAB * TK * LDA;
    constexpr int LDC = TN + 4;                          // the finished tile, staged for the vectorised Adam epilogue
    static_assert(TM * LDC <= DW_NSLAB * TK * (LDA + LDB), "the C tile must fit the slab buffers");
    // Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), each with its own L2.  Consecutive work items
    // (member-major, then job, then tile) re-read the same X / dZ panels, so XCD x gets the x-th CONTIGUOUS eighth of them:
    // a panel is then fetched into one L2 instead of up to eight (the kernel is fabric-bound: W, m, v alone are 44 MB).
    const int per_xcd = (a.tiles * a.E + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= a.tiles * a.E) return;
    const int e = item / a.tiles, tile = item - e * a.tiles;
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < a.njobs && tile >= a.job[ji + 1].tile0) ++ji;
    const DwJob& jb = a.job[ji];
    const int t = tile - jb.tile0;
    const int mb = (t / jb.tn) * TM, nb = (t % jb.tn) * TN;
    const int M = jb.M, N = jb.N, K = a.B;
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const float* A = jb.X + (long)e * K * jb.ldx;       // A(m = k_in, k = b) = X[b][k_in]
    const float* Bm = jb.dZ + (long)e * K * N;          // B(k = b, n)        = dZ[b][n]
    constexpr int MI = TM / 16;
    floatx4 acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float colsum = 0.0f;                                // bias gradient (threads < 64 of the m-tile-0 blocks)
    constexpr int NLA = TM * TK / 256, NLB = TN * TK / 256;
    float ra[DW_NSLAB][NLA], rb[DW_NSLAB][NLB];
    // Loads are UNCONDITIONAL (addresses clamped into the matrix, out-of-range elements zeroed afterwards): a
    // `cond ? *p : 0` select makes hipcc branch around every load and wait for each one in turn.
    const float* pa[NLA];
    const float* pb[NLB];
    int ka[NLA], kb[NLB], la[NLA], lb[NLB];
    bool va[NLA], vb[NLB];
#pragma unroll
    for (int it = 0; it < NLA; ++it) {
        const int idx = tid + it * 256;
        const int ak = idx / TM, am = idx - ak * TM;
        ka[it] = ak; la[it] = ak * LDA + am;
        va[it] = mb + am < M;
        pa[it] = A + (va[it] ? mb + am : 0);
    }
#pragma unroll
    for (int it = 0; it < NLB; ++it) {
        const int idx = tid + it * 256;
        const int bn = idx & (TN - 1), bk = idx / TN;
        kb[it] = bk; lb[it] = bk * LDB + bn;
        vb[it] = nb + bn < N;
        pb[it] = Bm + (vb[it] ? nb + bn : 0);
    }
    const int kmax = K - 1;
    const bool do_colsum = jb.bW && mb == 0 && tid < TN;

    // The loads of slab s+1 are issued right after slab s has been stashed into LDS -- into the SAME registers, which are
    // dead by then -- so their latency runs under slab s's MFMAs at no register cost.
    static_assert(DW_NSLAB == 1, "the slab pipeline below keeps one slab of loads in flight");
    const int KP = jb.X ? K : 0;                                       // X == null: L2-only job, gradient = wdc * W
    auto issue = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NLA; ++it) {
            const int k = k0 + ka[it];
            ra[0][it] = pa[it][(long)(k < kmax ? k : kmax) * jb.ldx];
        }
#pragma unroll
        for (int it = 0; it < NLB; ++it) {
            const int k = k0 + kb[it];
            rb[0][it] = pb[it][(long)(k < kmax ? k : kmax) * N];
        }
    };
    if (KP > 0) issue(0);
    for (int k0 = 0; k0 < KP; k0 += TK) {
        if (k0 > 0) __syncthreads();               // previous slab fully consumed before its LDS is overwritten
#pragma unroll
        for (int it = 0; it < NLA; ++it) As[la[it]] = (va[it] && k0 + ka[it] <= kmax) ? ra[0][it] : 0.0f;
#pragma unroll
        for (int it = 0; it < NLB; ++it) Bs[lb[it]] = (vb[it] && k0 + kb[it] <= kmax) ? rb[0][it] : 0.0f;
        __syncthreads();
        if (k0 + TK < KP) issue(k0 + TK);
        if (do_colsum) {
#pragma unroll
            for (int kk = 0; kk < TK; ++kk) colsum += Bs[kk * LDB + tid];
        }
#pragma unroll
        for (int ks = 0; ks < TK / 4; ++ks) {
            const int kr = ks * 4 + (lane >> 4);
            const float b = Bs[kr * LDB + wn * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[kr * LDA + i * 16 + (lane & 15)], b, acc[i], 0, 0, 0);
        }
    }

    // ---- epilogue: D layout col = lane & 15 -> n, row = (lane >> 4) * 4 + r -> m.  Adam touches W, m and v once each
    // (read + write): that traffic, not the GEMM, is most of this kernel, so the tile goes through LDS and every thread
    // updates 4 consecutive columns with 16-byte accesses (a D-layout thread would touch 12 scattered dwords x 6) ----
    if ((N & 3) == 0) {
        __syncthreads();                                 // every wave is done reading the slab buffers
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) dw_smem[(i * 16 + (lane >> 4) * 4 + r) * LDC + wn * 16 + (lane & 15)] = acc[i][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < TM * TN / 4 / 256; ++it) {
            const int idx = tid + it * 256;
            const int ml = idx / (TN / 4), n4 = (idx % (TN / 4)) * 4;
            const int m = mb + ml, n = nb + n4;
            if (m >= M || n >= N) continue;
            const long o = ((long)e * M + m) * N + n;
            const floatx4 g = *reinterpret_cast<const floatx4*>(dw_smem + ml * LDC + n4);
            floatx4 w = *reinterpret_cast<const floatx4*>(jb.W + o), mo = *reinterpret_cast<const floatx4*>(jb.Mw + o),
                    vo = *reinterpret_cast<const floatx4*>(jb.Vw + o);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float wc = w[c], mc = mo[c], vc = vo[c];
                adam_update(wc, mc, vc, g[c] + jb.wdc * wc, a.lr_t, a.b1, a.b2, a.eps);
                w[c] = wc; mo[c] = mc; vo[c] = vc;
            }
            *reinterpret_cast<floatx4*>(jb.W + o) = w;
            *reinterpret_cast<floatx4*>(jb.Mw + o) = mo;
            *reinterpret_cast<floatx4*>(jb.Vw + o) = vo;
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + i * 16 + (lane >> 4) * 4 + r;
                const int n = nb + wn * 16 + (lane & 15);
                if (m >= M || n >= N) continue;
                const long o = ((long)e * M + m) * N + n;
                float w = jb.W[o], mo = jb.Mw[o], vo = jb.Vw[o];
                adam_update(w, mo, vo, acc[i][r] + jb.wdc * w, a.lr_t, a.b1, a.b2, a.eps);
                jb.W[o] = w; jb.Mw[o] = mo; jb.Vw[o] = vo;
            }
    }
    if (do_colsum && nb + tid < N) {
        const long o = (long)e * N + nb + tid;
        float w = jb.bW[o], mo = jb.bM[o], vo = jb.bV[o];
        adam_update(w, mo, vo, colsum, a.lr_t, a.b1, a.b2, a.eps);
        jb.bW[o] = w; jb.bM[o] = mo; jb.bV[o] = vo;
    }
}

// ---------------------------------------------------------------------------------------------
// input assembly (core/utils.py:372-379 and :619-621 of the reference)
// ---------------------------------------------------------------------------------------------
// Where batch row (e, b) lives in the caller's tensors.  Direct: row r of [E*B, .] tensors.  Indexed (`fit`'s windowed
// dataset, cadm_train_step_rows): rid = idx[r], window w = row_w[rid], future offset f = row_f[rid]; per-step tensors are
// [N, F, .] (source row w*F + f), history tensors [N, .] (source row w).
struct RowMap {
    const long long *idx, *row_w, *row_f;
    int F, B;
    long long idx_ld;                 // idx[e * idx_ld + b]: a batch is a column slice of the [E, n_train] bootstrap matrix
};
__device__ __forceinline__ void map_row(const RowMap& m, long r, long& srow, long& swin) {
    if (!m.idx) { srow = r; swin = r; return; }
    const long long rid = m.idx[(r / m.B) * m.idx_ld + r % m.B];
    swin = m.row_w[rid];
    srow = swin * m.F + m.row_f[rid];
}

struct AsmP {
    RowMap map;
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
    long srow, swin;
    map_row(p.map, row, srow, swin);
    for (int f = threadIdx.x; f < p.P + p.A; f += blockDim.x) {
        if (f < p.P) {
            const float inv = p.obs_std[f] + 1e-10f;
            p.Xff[(long)row * p.K0 + f] = (preproc_at(p.env, p.obs + srow * p.D, f) - p.obs_mean[f]) / inv;
            if (p.has_back) p.Xbk[(long)row * p.K0 + f] = (preproc_at(p.env, p.obs_next + srow * p.D, f) - p.obs_mean[f]) / inv;
        } else {
            const int a = f - p.P;
            const float v = (p.act[srow * p.A + a] - p.act_mean[a]) / (p.act_std[a] + 1e-10f);
            p.Xff[(long)row * p.K0 + f] = v;
            if (p.has_back) p.Xbk[(long)row * p.K0 + f] = v;
        }
    }
    if (p.has_cp) {
        const int n = p.ncpo + p.ncpa;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            float v;
            if (i < p.ncpo) v = (p.cp_obs[swin * p.ncpo + i] - p.cp_obs_mean[i]) / (p.cp_obs_std[i] + 1e-10f);
            else v = (p.cp_act[swin * p.ncpa + (i - p.ncpo)] - p.cp_act_mean[i - p.ncpo]) / (p.cp_act_std[i - p.ncpo] + 1e-10f);
            p.Xcp[(long)row * n + i] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// losses (dynamics.py:269-314) and output-layer gradients
// ---------------------------------------------------------------------------------------------
struct LossP {
    RowMap map;
    const float *mu, *lv, *bmu;              // head outputs [E*B, D]
    const float *delta, *back_delta;         // raw targets [E*B, D] (or through map)
    const float *dmean, *dstd, *bdmean, *bdstd, *maxlv, *minlv;
    float *dMu, *dLv, *dBmu;                 // d loss / d head pre-activation
    long n;                                  // E*B*D
    int D, B, det, has_back;
    float back_coeff;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Deterministic reductions of the loss terms.  out: [4 + 2D] = {mse, mu_loss, var_loss, back_mse, d/d max_logvar [D],
// d/d min_logvar [D]}.  The workgroup that finishes last turns the sums into losses_out = [mse, back_mse, recon]
// (dynamics.py:505-507: recon = loss - reg - coeff * l2) and, when training a probabilistic model, applies Adam to
// max/min_logvar (data term + the 0.01 regulariser of dynamics.py:308) -- nothing else reads them until the next step.
struct ReduceP {
    float* part;                                   // [workgroups][4 + 2D] per-workgroup partial sums
    int D; float* out; unsigned* counter;
    int det, has_back; float back_coeff; float* losses_out;
    int adam_mm;                                   // 1: update max/min_logvar
    float *maxlv, *minlv, *mx_m, *mx_v, *mn_m, *mn_v;
    float lr_t, b1, b2, eps;
};

__device__ __forceinline__ float wave_sum_fixed(float v) {            // xor butterfly: the same order on every run
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// One launch for the losses, the head gradients and every reduction over them (was: a loss kernel that wrote 6 term
// arrays + a reduction kernel that read them back).  A workgroup takes LR_THREADS consecutive elements of [E*B, D],
// reduces its six terms in LDS in a fixed order (4 scalar sums; per-dim sums of the two logvar-bound gradients) and
// writes 4 + 2D partials; the last workgroup to arrive sums the partials (again in a fixed order: no float atomics,
// the result does not depend on which workgroup is last) and finalises.
constexpr int LR_THREADS = 1024;

__global__ __launch_bounds__(LR_THREADS) void loss_reduce_kernel(const LossP p, const ReduceP r) {
    __shared__ float sh[6][LR_THREADS];
    __shared__ float wsum[LR_THREADS / 64][64];
    __shared__ bool is_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned i0 = blockIdx.x * (unsigned)LR_THREADS;          // (host: n < 2^31 -- 32-bit div / mod)
    const unsigned i = i0 + tid;
    float tm[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (i < (unsigned)p.n) {
        const int d = (int)(i % (unsigned)p.D);
        long srow, swin;
        map_row(p.map, (long)(i / (unsigned)p.D), srow, swin);
        const long si = srow * p.D + d;                            // this element in the caller's target tensors
        const float s = 1.0f / ((float)p.B * (float)p.D);         // reduce_mean over b then d; reduce_sum over e
        const float t = (p.delta[si] - p.dmean[d]) / (p.dstd[d] + 1e-10f);
        const float mu = p.mu[i];
        const float diff = mu - t;
        tm[0] = diff * diff * s;                                                  // mse            (:273-274)
        if (p.det) {
            p.dMu[i] = 2.0f * s * diff;
            p.dLv[i] = 0.0f;
        } else {
            const float mx = p.maxlv[d], mn = p.minlv[d], lv0 = p.lv[i];
            const float u = mx - tf_softplus(mx - lv0);                           // core/utils.py:356
            const float lvc = mn + tf_softplus(u - mn);                           // core/utils.py:357
            const float invvar = expf(-lvc);                                      // :303
            tm[1] = diff * diff * invvar * s;                                     // mu_loss        (:304-305)
            tm[2] = lvc * s;                                                      // var_loss       (:306-307)
            const float g_lvc = s * (1.0f - diff * diff * invvar);
            const float s1 = sigmoidf_(u - mn), s2 = sigmoidf_(mx - lv0);         // softplus' = sigmoid
            p.dMu[i] = 2.0f * s * diff * invvar;
            p.dLv[i] = g_lvc * s1 * s2;
            tm[4] = g_lvc * s1 * (1.0f - s2);                                     // d / d max_logvar (without the 0.01 reg)
            tm[5] = g_lvc * (1.0f - s1);                                          // d / d min_logvar
        }
        if (p.has_back) {
            const float tb = (p.back_delta[si] - p.bdmean[d]) / (p.bdstd[d] + 1e-10f);
            const float db = p.bmu[i] - tb;
            tm[3] = db * db * s;                                                  // back_mse       (:280-281)
            p.dBmu[i] = p.back_coeff * 2.0f * s * db;
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) sh[q][tid] = tm[q];
    __syncthreads();
    const int NQ = 4 + 2 * r.D;
    float* part = r.part + (size_t)blockIdx.x * NQ;
    if (wave < 4) {                                                // scalar terms: wave q sums sh[q][*]
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < LR_THREADS / 64; ++k) v += sh[wave][lane + 64 * k];
        v = wave_sum_fixed(v);
        if (lane == 0) part[wave] = v;
    } else {                                                       // per-dim terms: 16 lanes per (bound, dim)
        const int sub = tid & 15, j00 = (int)(i0 % (unsigned)r.D);
        for (int o = (tid - 256) >> 4; o < 2 * r.D; o += (LR_THREADS - 256) >> 4) {     // (uniform per 16-lane group)
            const int which = o / r.D, d = o % r.D;
            float v = 0.0f;
            for (int j = (d - j00 + r.D) % r.D + sub * r.D; j < LR_THREADS; j += 16 * r.D) v += sh[4 + which][j];
#pragma unroll
            for (int x = 8; x > 0; x >>= 1) v += __shfl_xor(v, x, 64);
            if (sub == 0) part[4 + o] = v;
        }
    }
    __syncthreads();                                              // (every wave's partials have left for L2)
    if (tid == 0) {
        __threadfence();                                          // one agent-scope release per workgroup
        is_last = atomicInc(r.counter, gridDim.x - 1) == gridDim.x - 1;     // wraps back to 0 for the next step
        if (is_last) __threadfence();
    }
    __syncthreads();
    if (!is_last) return;
    // sum of the partials: lane = output (64 at a time), the waves take interleaved workgroups, wave partials meet in LDS
    const volatile float* all = r.part;
    volatile float* red = r.out;
    const int W = (int)gridDim.x;
    for (int q0 = 0; q0 < NQ; q0 += 64) {
        const int q = q0 + lane;
        float v = 0.0f;
        if (q < NQ)
            for (int w = wave; w < W; w += LR_THREADS / 64) v += all[(size_t)w * NQ + q];
        wsum[wave][lane] = v;
        __syncthreads();
        if (wave == 0 && q < NQ) {
            float tot = 0.0f;
#pragma unroll
            for (int k = 0; k < LR_THREADS / 64; ++k) tot += wsum[k][lane];
            red[q] = tot;
        }
        __syncthreads();
    }
    __threadfence_block();
    if (tid == 0) {
        const float mse = red[0], mu_loss = red[1], var_loss = red[2], back = red[3];
        float recon = r.det ? mse : mu_loss + var_loss;
        if (r.has_back) recon += r.back_coeff * back;
        r.losses_out[0] = mse;
        r.losses_out[1] = r.has_back ? back : 0.0f;
        r.losses_out[2] = recon;
    }
    if (r.adam_mm && tid < 2 * r.D) {
        const bool mx = tid < r.D;
        const int d = mx ? tid : tid - r.D;
        float* w = (mx ? r.maxlv : r.minlv) + d;
        float* m = (mx ? r.mx_m : r.mn_m) + d;
        float* v = (mx ? r.mx_v : r.mn_v) + d;
        float ww = *w, mm = *m, vv = *v;
        adam_update(ww, mm, vv, red[4 + tid] + (mx ? 0.01f : -0.01f), r.lr_t, r.b1, r.b2, r.eps);
        *w = ww; *m = mm; *v = vv;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct NetBufs {
    std::vector<float*> z, h, dz;     // per hidden layer [E,B,width]; dz: gradient w.r.t. the pre-activation
    float *mu = nullptr, *lv = nullptr;
    float* dctx = nullptr;            // this net's gradient w.r.t. its context input columns [E,B,C]
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
    float *Xff = nullptr, *Xbk = nullptr, *Xcp = nullptr, *dCtx = nullptr;
    NetBufs ff, bk, cp;
    float *dMu = nullptr, *dLv = nullptr, *dBmu = nullptr, *terms = nullptr, *red = nullptr;
    // Adam moments, same order as the registered layers: W then b
    std::vector<AdamSlot> a_ff, a_bk, a_cp;   // 2 per layer
    AdamSlot a_mx, a_mn;
    float* adam_buf = nullptr;
    // chain programs: [fwd ff | fwd back | bwd ff | bwd back | bwd context]
    std::vector<ChainStage> prog_host;
    ChainStage* prog_dev = nullptr;
    int prog_first[5] = {0, 0, 0, 0, 0}, prog_count[5] = {0, 0, 0, 0, 0};
    int chain_bufsz = 0;
};

void cadm_train_free(cadm_ctx* ctx) {
    if (!ctx->train) return;
    if (ctx->train->ws) (void)hipFree(ctx->train->ws);
    if (ctx->train->adam_buf) (void)hipFree(ctx->train->adam_buf);
    if (ctx->train->prog_dev) (void)hipFree(ctx->train->prog_dev);
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
    const size_t Cw = ctx->C > 0 ? ctx->C : 1;
    size_t total = 0;
    auto need = [&](size_t n) { size_t o = total; total += (n + 63) & ~(size_t)63; return o; };
    const size_t oXff = need(R * K0), oXbk = need(R * K0), oXcp = need(R * (cpin > 0 ? cpin : 1));
    const size_t odCtx = need(R * Cw), odCff = need(R * Cw), odCbk = need(R * Cw);
    std::vector<size_t> oz_ff(NH), oh_ff(NH), od_ff(NH), oz_bk(NH), oh_bk(NH), od_bk(NH), oz_cp(ncp), oh_cp(ncp), od_cp(ncp);
    for (int l = 0; l < NH; ++l) {
        oz_ff[l] = need(R * HID); oh_ff[l] = need(R * HID); od_ff[l] = need(R * HID);
        oz_bk[l] = need(R * HID); oh_bk[l] = need(R * HID); od_bk[l] = need(R * HID);
    }
    for (int l = 0; l < ncp; ++l) {
        const size_t w = ctx->cfg.cp_hidden[l];
        oz_cp[l] = need(R * w); oh_cp[l] = need(R * w); od_cp[l] = need(R * w);
    }
    const size_t omu = need(R * D), olv = need(R * D), obmu = need(R * D), oblv = need(R * D);
    const size_t odMu = need(R * D), odLv = need(R * D), odBmu = need(R * D);
    const size_t oterms = need(((R * D + LR_THREADS - 1) / LR_THREADS) * (4 + 2 * (size_t)D)), ored = need(4 + 2 * (size_t)D + 8);
    CADM_CHECK_HIP(hipMalloc(&t->ws, total * sizeof(float)));
    t->ws_floats = total;
    float* w = t->ws;
    t->Xff = w + oXff; t->Xbk = w + oXbk; t->Xcp = w + oXcp; t->dCtx = w + odCtx; t->ff.dctx = w + odCff; t->bk.dctx = w + odCbk;
    for (NetBufs* nb : {&t->ff, &t->bk}) { nb->z.resize(NH); nb->h.resize(NH); nb->dz.resize(NH); }
    t->cp.z.resize(ncp); t->cp.h.resize(ncp); t->cp.dz.resize(ncp);
    for (int l = 0; l < NH; ++l) {
        t->ff.z[l] = w + oz_ff[l]; t->ff.h[l] = w + oh_ff[l]; t->ff.dz[l] = w + od_ff[l];
        t->bk.z[l] = w + oz_bk[l]; t->bk.h[l] = w + oh_bk[l]; t->bk.dz[l] = w + od_bk[l];
    }
    for (int l = 0; l < ncp; ++l) { t->cp.z[l] = w + oz_cp[l]; t->cp.h[l] = w + oh_cp[l]; t->cp.dz[l] = w + od_cp[l]; }
    t->ff.mu = w + omu; t->ff.lv = w + olv; t->bk.mu = w + obmu; t->bk.lv = w + oblv;
    t->dMu = w + odMu; t->dLv = w + odLv; t->dBmu = w + odBmu;
    t->terms = w + oterms; t->red = w + ored;
    CADM_CHECK_HIP(hipMemset(t->red, 0, (4 + 2 * (size_t)D + 8) * sizeof(float)));   // incl. the reduction's arrival counter
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
    CADM_ON_DEVICE(ctx);
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
    CADM_ON_DEVICE(ctx);
    TrainState* t = ctx->train;
    size_t total = 0;
    for (auto* v : {&t->a_ff, &t->a_bk, &t->a_cp}) for (auto& s : *v) total += 2 * s.n;
    total += 4 * (size_t)ctx->D;
    CADM_CHECK_HIP(hipMemsetAsync(t->adam_buf, 0, total * sizeof(float), (hipStream_t)stream));
    t->step = 0;
    return CADM_OK;
}

namespace {

enum { PROG_FWD_FF = 0, PROG_FWD_BK = 1, PROG_BWD_FF = 2, PROG_BWD_BK = 3, PROG_BWD_CP = 4 };

int pad32(int w) { return ((w + 31) & ~31) - w; }

// `complete`: this stage finishes the buffer (width dk0 + K) -> zero-pad behind it
ChainStage load_stage(const float* g0, const float* g1, float* gsum, int ld_in, int ldo, int K, int dst, int dk0, bool complete = true) {
    ChainStage s{};
    s.kind = ST_LOAD; s.g0 = g0; s.g1 = g1; s.gsum = gsum; s.ld_in = ld_in; s.ldo = ldo; s.K = K; s.dst = dst; s.dk0 = dk0;
    s.zpad = complete ? pad32(dk0 + K) : 0;
    return s;
}

ChainPart part_of(const DenseRef& L, int wt, int row0, int K, int src) {
    ChainPart p{};
    p.W = L.W; p.sWe = (long)L.din * L.dout; p.ldw = L.dout; p.wt = wt; p.row0 = row0; p.K = K; p.src = src;
    return p;
}

// Builds the five stage lists for the current pointers and uploads them if anything changed.
int sync_programs(cadm_ctx* ctx, hipStream_t s) {
    TrainState* t = ctx->train;
    const bool has_back = ctx->cfg.back_model != 0, has_cp = ctx->C > 0, det = ctx->cfg.deterministic != 0;
    const int NH = ctx->NH, HID = ctx->HID, D = ctx->D, K0 = ctx->K0, C = ctx->C, PA = ctx->P + ctx->A;
    const int ncp = has_cp ? ctx->cfg.n_cp_hidden : 0;
    const int cpin = (ctx->D + ctx->A) * ctx->cfg.history_length;
    std::vector<ChainStage> prog;
    auto push_gemm = [&](ChainStage g) {
        g.zpad = g.dst >= 0 ? pad32(g.dk0 + g.N) : 0;
        prog.push_back(g);
    };
    int first[5], count[5];
    int maxk = K0 > HID ? K0 : HID;
    maxk = D > maxk ? D : maxk;
    if (has_cp) { maxk = cpin > maxk ? cpin : maxk; for (int l = 0; l < ncp; ++l) maxk = ctx->cfg.cp_hidden[l] > maxk ? ctx->cfg.cp_hidden[l] : maxk; }

    auto fwd_prog = [&](const std::vector<DenseRef>& net, float* X, NetBufs& nb, bool store_cp, bool want_lv) {
        int cur = 0;
        if (has_cp) {
            prog.push_back(load_stage(t->Xcp, nullptr, nullptr, cpin, 0, cpin, 0, 0));
            for (int l = 0; l <= ncp; ++l) {
                const DenseRef& L = ctx->cp[l];
                ChainStage g{};
                g.kind = ST_GEMM; g.N = L.dout; g.nparts = 1; g.part[0] = part_of(L, 0, 0, L.din, cur);
                g.bias = L.b; g.dst = cur ^ 1;
                if (l < ncp) {
                    g.act_o = ACT_RELU; g.ldo = L.dout;
                    if (store_cp) { g.out0 = t->cp.z[l]; g.out1 = t->cp.h[l]; }
                } else {   // context vector -> the ctx columns of this net's input (LDS and global)
                    g.act_o = ACT_NONE; g.out1 = X + PA; g.ldo = K0; g.dk0 = PA;
                }
                push_gemm(g);
                cur ^= 1;
            }
            prog.push_back(load_stage(X, nullptr, nullptr, K0, 0, PA, cur, 0, false));   // [PA, K0) holds the context vector
        } else {
            prog.push_back(load_stage(X, nullptr, nullptr, K0, 0, K0, 0, 0));
        }
        for (int l = 0; l < NH; ++l) {
            ChainStage g{};
            g.kind = ST_GEMM; g.N = HID; g.nparts = 1; g.part[0] = part_of(net[l], 0, 0, net[l].din, cur);
            g.bias = net[l].b; g.act_o = dyn_act(ctx); g.out0 = nb.z[l]; g.out1 = nb.h[l]; g.ldo = HID; g.dst = cur ^ 1;
            push_gemm(g);
            cur ^= 1;
        }
        for (int hd = 0; hd < (want_lv ? 2 : 1); ++hd) {
            ChainStage g{};
            g.kind = ST_GEMM; g.N = D; g.nparts = 1; g.part[0] = part_of(net[NH + hd], 0, 0, HID, cur);
            g.bias = net[NH + hd].b; g.act_o = ACT_NONE; g.out1 = hd ? nb.lv : nb.mu; g.ldo = D; g.dst = -1;
            push_gemm(g);
        }
    };
    auto bwd_prog = [&](const std::vector<DenseRef>& net, NetBufs& nb, const float* dMu, const float* dLv) {
        prog.push_back(load_stage(dMu, nullptr, nullptr, D, 0, D, 0, 0));
        if (dLv) prog.push_back(load_stage(dLv, nullptr, nullptr, D, 0, D, 1, 0));
        {   // d z_{NH-1} = (dMu W_mu^T (+ dLv W_lv^T)) * swish'(z_{NH-1})
            ChainStage g{};
            g.kind = ST_GEMM; g.N = HID; g.nparts = dLv ? 2 : 1;
            g.part[0] = part_of(net[NH], 1, 0, D, 0);
            if (dLv) g.part[1] = part_of(net[NH + 1], 1, 0, D, 1);
            g.zprev = nb.z[NH - 1]; g.ldz = HID; g.act_d = dyn_act(ctx); g.out1 = nb.dz[NH - 1]; g.ldo = HID; g.dst = 2;
            push_gemm(g);
        }
        int cur = 2;
        for (int l = NH - 1; l >= 1; --l) {
            ChainStage g{};
            g.kind = ST_GEMM; g.N = net[l].din; g.nparts = 1; g.part[0] = part_of(net[l], 1, 0, net[l].dout, cur);
            g.zprev = nb.z[l - 1]; g.ldz = HID; g.act_d = dyn_act(ctx); g.out1 = nb.dz[l - 1]; g.ldo = HID; g.dst = (cur + 1) % 3;
            push_gemm(g);
            cur = (cur + 1) % 3;
        }
        if (has_cp) {   // only the context columns of the input carry a gradient
            ChainStage g{};
            g.kind = ST_GEMM; g.N = C; g.nparts = 1; g.part[0] = part_of(net[0], 1, PA, net[0].dout, cur);
            g.out1 = nb.dctx; g.ldo = C; g.dst = -1;
            push_gemm(g);
        }
    };

    first[PROG_FWD_FF] = (int)prog.size(); fwd_prog(ctx->ff, t->Xff, t->ff, true, !det); count[PROG_FWD_FF] = (int)prog.size() - first[PROG_FWD_FF];
    first[PROG_FWD_BK] = (int)prog.size(); if (has_back) fwd_prog(ctx->back, t->Xbk, t->bk, false, false); count[PROG_FWD_BK] = (int)prog.size() - first[PROG_FWD_BK];
    first[PROG_BWD_FF] = (int)prog.size(); bwd_prog(ctx->ff, t->ff, t->dMu, det ? nullptr : t->dLv); count[PROG_BWD_FF] = (int)prog.size() - first[PROG_BWD_FF];
    first[PROG_BWD_BK] = (int)prog.size(); if (has_back) bwd_prog(ctx->back, t->bk, t->dBmu, nullptr); count[PROG_BWD_BK] = (int)prog.size() - first[PROG_BWD_BK];
    first[PROG_BWD_CP] = (int)prog.size();
    if (has_cp) {
        prog.push_back(load_stage(t->ff.dctx, has_back ? t->bk.dctx : nullptr, t->dCtx, C, C, C, 0, 0));
        int cur = 0;
        for (int l = ncp; l >= 1; --l) {
            const DenseRef& L = ctx->cp[l];
            ChainStage g{};
            g.kind = ST_GEMM; g.N = L.din; g.nparts = 1; g.part[0] = part_of(L, 1, 0, L.dout, cur);
            g.zprev = t->cp.z[l - 1]; g.ldz = L.din; g.act_d = ACT_RELU; g.out1 = t->cp.dz[l - 1]; g.ldo = L.din; g.dst = cur ^ 1;
            push_gemm(g);
            cur ^= 1;
        }
    }
    count[PROG_BWD_CP] = (int)prog.size() - first[PROG_BWD_CP];
    for (int i = 0; i < 5; ++i) CADM_REQUIRE(count[i] <= CH_MAXSTAGE, "training chain too long (more than 20 stages): too many layers");

    const bool same = t->prog_dev && prog.size() == t->prog_host.size() &&
                      memcmp(prog.data(), t->prog_host.data(), prog.size() * sizeof(ChainStage)) == 0;
    if (!same) {
        if (t->prog_dev && prog.size() > t->prog_host.size()) { (void)hipFree(t->prog_dev); t->prog_dev = nullptr; }
        if (!t->prog_dev) CADM_CHECK_HIP(hipMalloc(&t->prog_dev, prog.size() * sizeof(ChainStage)));
        CADM_CHECK_HIP(hipStreamSynchronize(s));   // nothing in flight may still read the old table
        CADM_CHECK_HIP(hipMemcpy(t->prog_dev, prog.data(), prog.size() * sizeof(ChainStage), hipMemcpyHostToDevice));
        t->prog_host = prog;
    }
    for (int i = 0; i < 5; ++i) { t->prog_first[i] = first[i]; t->prog_count[i] = count[i]; }
    t->chain_bufsz = CH_ROWS * ((maxk + 31) & ~31);
    return CADM_OK;
}

int launch_chain(cadm_ctx* ctx, int B, int p0, int p1, hipStream_t s) {
    TrainState* t = ctx->train;
    ChainArgs a{};
    a.prog = t->prog_dev;
    a.first[0] = t->prog_first[p0]; a.count[0] = t->prog_count[p0];
    a.ny = 1;
    if (p1 >= 0 && t->prog_count[p1] > 0) { a.first[1] = t->prog_first[p1]; a.count[1] = t->prog_count[p1]; a.ny = 2; }
    a.B = B; a.bufsz = t->chain_bufsz;
    a.tbuf = ctx->tbuf ? ctx->tbuf + 256 * (p0 / 2) : nullptr;   // [fwd | bwd | bwd context] x 256 stamps (tools/chain_timing.py)
    a.E = ctx->E; a.ntiles = (B + CH_ROWS - 1) / CH_ROWS;
    a.G = ctx->E <= 8 ? 8 / ctx->E : 1;
    const int per = a.ntiles * a.ny;
    a.ips = (per + a.G - 1) / a.G;
    const int rounds = (ctx->E + 7) / 8;
    auto pf_net = [&](const std::vector<DenseRef>& net) {
        for (auto& d : net)
            if (a.npf < CH_MAXPF) { a.pf_ptr[a.npf] = d.W; a.pf_n[a.npf] = d.din * d.dout; ++a.npf; }
    };
    if (p0 == PROG_BWD_CP || (p0 == PROG_FWD_FF && ctx->C > 0)) pf_net(ctx->cp);
    if (p0 != PROG_BWD_CP) { pf_net(ctx->ff); if (a.ny == 2) pf_net(ctx->back); }
    const size_t lds = CH_MAXSTAGE * sizeof(ChainStage) + 3 * (size_t)t->chain_bufsz * sizeof(float);
    CADM_REQUIRE(lds <= 160 * 1024, "training chain: layer too wide for the LDS-resident activation tile");
    CADM_REQUIRE((long long)B * (t->chain_bufsz / CH_ROWS) * 4 < (1LL << 32),
                 "training chain: batch of %d rows too large for 32-bit per-member offsets", B);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        CADM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(chain_kernel, dim3(8 * a.ips * rounds), dim3(CH_THREADS), lds, s, a);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

// forward of the context / forward (/ backward) nets on one [E,B,.] batch into the workspace
int forward_nets(cadm_ctx* ctx, const RowMap& map, const float* obs, const float* act, const float* obs_next, const float* cp_obs,
                 const float* cp_act, int B, bool has_back, hipStream_t s) {
    TrainState* t = ctx->train;
    const bool has_cp = ctx->C > 0;
    const int E = ctx->E, D = ctx->D, K0 = ctx->K0;
    const long R = (long)E * B;
    int rc;
    if ((rc = sync_programs(ctx, s))) return rc;
    AsmP ap{};
    ap.map = map;
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
    return launch_chain(ctx, B, PROG_FWD_FF, has_back ? PROG_FWD_BK : -1, s);
}

__global__ void clamp_logvar_kernel(const float* lv, const float* maxlv, const float* minlv, float* out, long n, int D) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % D);
    const float u = maxlv[d] - tf_softplus(maxlv[d] - lv[i]);      // core/utils.py:356
    out[i] = minlv[d] + tf_softplus(u - minlv[d]);                 // core/utils.py:357
}
}  // namespace

static int train_step_impl(cadm_ctx* ctx, const RowMap& map, const float* obs, const float* act, const float* delta,
                           const float* obs_next, const float* back_delta, const float* cp_obs, const float* cp_act, int B,
                           int train, float* losses_out, void* stream) {
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
    const int E = ctx->E, NH = ctx->NH, HID = ctx->HID, D = ctx->D, K0 = ctx->K0;
    const int ncp = has_cp ? ctx->cfg.n_cp_hidden : 0;
    const int cpin = (ctx->D + ctx->A) * ctx->cfg.history_length;
    const long R = (long)E * B;

    if ((rc = forward_nets(ctx, map, obs, act, obs_next, cp_obs, cp_act, B, has_back, s))) return rc;

    // ---- losses + head gradients ----
    LossP lp{};
    lp.map = map;
    lp.mu = t->ff.mu; lp.lv = t->ff.lv; lp.bmu = t->bk.mu; lp.delta = delta; lp.back_delta = back_delta;
    lp.dmean = ctx->st.delta_mean; lp.dstd = ctx->st.delta_std; lp.bdmean = ctx->st.back_delta_mean; lp.bdstd = ctx->st.back_delta_std;
    lp.maxlv = ctx->ff_maxlv; lp.minlv = ctx->ff_minlv;
    lp.dMu = t->dMu; lp.dLv = t->dLv; lp.dBmu = t->dBmu;
    lp.n = R * D; lp.D = D; lp.B = B; lp.det = det; lp.has_back = has_back; lp.back_coeff = hp.back_coeff;
    // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)  (TF1 Adam)
    if (train) t->step += 1;
    const float lr_t = (float)(hp.learning_rate * sqrt(1.0 - pow((double)hp.beta2, (double)t->step)) /
                               (1.0 - pow((double)hp.beta1, (double)t->step)));
    ReduceP rp{};
    rp.part = t->terms; rp.D = D; rp.out = t->red; rp.counter = reinterpret_cast<unsigned*>(t->red + 4 + 2 * D);
    rp.det = det; rp.has_back = has_back; rp.back_coeff = hp.back_coeff; rp.losses_out = losses_out;
    rp.adam_mm = train && !det;
    rp.maxlv = ctx->ff_maxlv; rp.minlv = ctx->ff_minlv;
    rp.mx_m = t->a_mx.m; rp.mx_v = t->a_mx.v; rp.mn_m = t->a_mn.m; rp.mn_v = t->a_mn.v;
    rp.lr_t = lr_t; rp.b1 = hp.beta1; rp.b2 = hp.beta2; rp.eps = hp.epsilon;
    CADM_REQUIRE(lp.n < (1L << 31), "cadm_train_step: E*B*D too large");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3((unsigned)((lp.n + LR_THREADS - 1) / LR_THREADS)), dim3(LR_THREADS), 0, s, lp, rp);
    CADM_CHECK_HIP(hipGetLastError());
    if (!train) return CADM_OK;

    // ---- backward + Adam ----
    const float coeff = hp.weight_decay_coeff;
    auto wd_dyn = [&](int l) { return coeff * (l < NH ? hp.weight_decays[l] : hp.weight_decays[NH]); };

    // backward chains (read W) ...
    if ((rc = launch_chain(ctx, B, PROG_BWD_FF, has_back ? PROG_BWD_BK : -1, s))) return rc;
    if (has_cp && (rc = launch_chain(ctx, B, PROG_BWD_CP, -1, s))) return rc;

    // ... then every layer's weight gradient + Adam as one grouped launch (overwrites W)
    DwArgs da{};
    da.B = B; da.lr_t = lr_t; da.b1 = hp.beta1; da.b2 = hp.beta2; da.eps = hp.epsilon;
    int tiles = 0;
    auto add_job = [&](const float* X, int ldx, const float* dZ, const DenseRef& L, float wdc, AdamSlot& aw, AdamSlot& ab) -> int {
        CADM_REQUIRE(da.njobs < DW_MAXJOBS, "cadm_train_step: too many layers for the grouped weight-gradient launch");
        DwJob& j = da.job[da.njobs++];
        j.X = X; j.dZ = dZ; j.W = L.W; j.Mw = aw.m; j.Vw = aw.v; j.bW = L.b; j.bM = ab.m; j.bV = ab.v;
        j.ldx = ldx; j.M = L.din; j.N = L.dout; j.tile0 = tiles; j.wdc = wdc; j.tn = (L.dout + TN - 1) / TN;
        tiles += j.tn * ((L.din + TM - 1) / TM);
        return CADM_OK;
    };
    auto net_jobs = [&](std::vector<DenseRef>& net, const float* X, NetBufs& nb, std::vector<AdamSlot>& ad, const float* dMu,
                        const float* dLv) -> int {
        int r;
        for (int l = 0; l < NH; ++l)
            if ((r = add_job(l == 0 ? X : nb.h[l - 1], l == 0 ? K0 : HID, nb.dz[l], net[l], wd_dyn(l), ad[2 * l], ad[2 * l + 1]))) return r;
        if ((r = add_job(nb.h[NH - 1], HID, dMu, net[NH], wd_dyn(NH), ad[2 * NH], ad[2 * NH + 1]))) return r;
        if (dLv && (r = add_job(nb.h[NH - 1], HID, dLv, net[NH + 1], wd_dyn(NH + 1), ad[2 * (NH + 1)], ad[2 * (NH + 1) + 1]))) return r;
        return CADM_OK;
    };
    if ((rc = net_jobs(ctx->ff, t->Xff, t->ff, t->a_ff, t->dMu, det ? nullptr : t->dLv))) return rc;
    if (has_back && (rc = net_jobs(ctx->back, t->Xbk, t->bk, t->a_bk, t->dBmu, nullptr))) return rc;
    if (has_cp) {
        auto wd_cp = [&](int l) { return coeff * (l < ncp ? hp.context_weight_decays[l] : hp.context_weight_decays[ncp]); };
        for (int l = 0; l <= ncp; ++l)
            if ((rc = add_job(l == 0 ? t->Xcp : t->cp.h[l - 1], l == 0 ? cpin : ctx->cp[l - 1].dout, l == ncp ? t->dCtx : t->cp.dz[l],
                              ctx->cp[l], wd_cp(l), t->a_cp[2 * l], t->a_cp[2 * l + 1]))) return rc;
    }
    // output_logvar outside the data path (deterministic forward net / backward net): its weight only sees the L2 term
    // (a job without data: X = null), its bias has no gradient at all and is skipped like TF does
    auto l2_only_job = [&](const DenseRef& L, float wdc, AdamSlot& aw) -> int {
        CADM_REQUIRE(da.njobs < DW_MAXJOBS, "cadm_train_step: too many layers for the grouped weight-gradient launch");
        DwJob& j = da.job[da.njobs++];
        j.X = nullptr; j.dZ = nullptr; j.W = L.W; j.Mw = aw.m; j.Vw = aw.v; j.bW = nullptr; j.bM = nullptr; j.bV = nullptr;
        j.ldx = 0; j.M = L.din; j.N = L.dout; j.tile0 = tiles; j.wdc = wdc; j.tn = (L.dout + TN - 1) / TN;
        tiles += j.tn * ((L.din + TM - 1) / TM);
        return CADM_OK;
    };
    if (det && (rc = l2_only_job(ctx->ff[NH + 1], wd_dyn(NH + 1), t->a_ff[2 * (NH + 1)]))) return rc;
    if (has_back && (rc = l2_only_job(ctx->back[NH + 1], wd_dyn(NH + 1), t->a_bk[2 * (NH + 1)]))) return rc;
    da.tiles = tiles; da.E = E;
    hipLaunchKernelGGL(dw_adam_kernel, dim3(8 * ((tiles * E + 7) / 8)), dim3(256), 0, s, da);
    CADM_CHECK_HIP(hipGetLastError());
    ctx->packed = false;   // planner streams are stale until cadm_repack
    return CADM_OK;
}

extern "C" int cadm_train_step(cadm_ctx* ctx, const float* obs, const float* act, const float* delta,
                               const float* obs_next, const float* back_delta, const float* cp_obs,
                               const float* cp_act, int B, int train, float* losses_out, void* stream) {
    return train_step_impl(ctx, RowMap{}, obs, act, delta, obs_next, back_delta, cp_obs, cp_act, B, train, losses_out, stream);
}

// The same step on rows of a WINDOWED dataset resident on the device (fit(), dynamics.py:382-569 + :676-696): per-step
// tensors [N, F, .], history tensors [N, .]; training row rid = (row_w[rid], row_f[rid]); the batch is idx[e * idx_ld + b].
extern "C" int cadm_train_step_rows(cadm_ctx* ctx, const float* ds_obs, const float* ds_act, const float* ds_delta,
                                    const float* ds_obs_next, const float* ds_back_delta, const float* ds_cp_obs,
                                    const float* ds_cp_act, int F, const long long* row_w, const long long* row_f,
                                    const long long* idx, long long idx_ld, int B, int train, float* losses_out,
                                    void* stream) {
    CADM_REQUIRE(F >= 1 && row_w && row_f && idx && idx_ld >= B, "cadm_train_step_rows: bad row index arguments");
    RowMap map{idx, row_w, row_f, F, B, idx_ld};
    return train_step_impl(ctx, map, ds_obs, ds_act, ds_delta, ds_obs_next, ds_back_delta, ds_cp_obs, ds_cp_act, B, train, losses_out,
                           stream);
}

// One-step prediction heads of every member on an [E,B,.] batch (the vanilla reference's `_get_pred`,
// mlp_ensemble_cem_dynamics.py:185-189: [mlp.mu, mlp.logvar]): normalised mean and clamped log-variance.
extern "C" int cadm_predict(cadm_ctx* ctx, const float* obs, const float* act, const float* cp_obs, const float* cp_act,
                            int B, float* mu_out, float* logvar_out, void* stream) {
    CADM_REQUIRE(ctx && obs && act && mu_out && B > 0, "cadm_predict: bad arguments");
    CADM_ON_DEVICE(ctx);
    CADM_REQUIRE(ctx->st.set, "cadm_predict: normalisation stats not set");
    CADM_REQUIRE(ctx->C == 0 || (cp_obs && cp_act), "cadm_predict: cp_obs / cp_act required (context model)");
    for (auto& d : ctx->ff) CADM_REQUIRE(d.W && d.b, "cadm_predict: ff_model weights not registered");
    if (ctx->C > 0) for (auto& d : ctx->cp) CADM_REQUIRE(d.W && d.b, "cadm_predict: context_model weights not registered");
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_state(ctx);
    if (rc) return rc;
    if ((rc = ensure_workspace(ctx, B))) return rc;
    if ((rc = forward_nets(ctx, RowMap{}, obs, act, nullptr, cp_obs, cp_act, B, false, s))) return rc;
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
