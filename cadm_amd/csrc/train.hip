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
// packed operand streams of the chain kernel (layout: ChainSeg)
// ---------------------------------------------------------------------------------------------
struct PackDst {          // where element (m, n) of a layer W [M][N] lives in one stream (tr: 0 forward, 1 transposed)
    float* P; long sP;    // stream base, member stride (floats); P == null: no stream
    int KB, kb0;          // k-blocks of the whole stream; this layer's first one (streams concatenated along k)
    int row0, ncols;      // forward: Bop(k, n') = W[k][n'];  transposed: Bop(k, n') = W[row0 + n'][k];  n' < ncols
    int nt, pad;          // tiles of the stream (even)
};
__device__ __forceinline__ long pack_index(const PackDst& d, int k, int np) {    // float index inside a member's stream
    return (((long)(d.kb0 + (k >> 4)) * d.nt + (np >> 4)) * 64 + ((k >> 2) & 3) * 16 + (np & 15)) * 4 + (k & 3);
}

struct PackJob {
    const float* W; int M, N;        // [E][M][N]
    PackDst d; int tr, nk, ntile;    // nk: valid k; ntile: tiles of the stream (even)
};
// Full (re)build of one layer's part of a stream, zero padding included: one float4 per thread.
__global__ void train_pack_kernel(const PackJob j, int E) {
    const int KBl = (j.nk + 15) >> 4;
    const long per = (long)j.ntile * KBl * 64, idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= per * E) return;
    const int e = (int)(idx / per);
    const long r = idx - e * per;
    const int lane = (int)(r & 63), t = (int)((r >> 6) % KBl), tile = (int)((r >> 6) / KBl);
    const int c = lane & 15, kq = lane >> 4, np = 16 * tile + c;
    const float* W = j.W + (long)e * j.M * j.N;
    floatx4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 16 * t + 4 * kq + i;
        v[i] = (k < j.nk && np < j.d.ncols) ? (j.tr ? W[(long)(j.d.row0 + np) * j.N + k] : W[(long)k * j.N + np]) : 0.0f;
    }
    *reinterpret_cast<floatx4*>(j.d.P + (long)e * j.d.sP + (((long)(j.d.kb0 + t) * j.d.nt + tile) * 64 + lane) * 4) = v;
}

// ---------------------------------------------------------------------------------------------
// input assembly (core/utils.py:372-379 and :619-621 of the reference): done by the forward chain's input tiles
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

struct ChainAsm {         // raw batch -> normalised network inputs
    RowMap map;
    const float *act, *cp_obs, *cp_act;
    const float *obs_mean, *obs_std, *act_mean, *act_std, *cp_obs_mean, *cp_obs_std, *cp_act_mean, *cp_act_std;
    int D, A, P, ncpo, ncpa, env;
};

// ---------------------------------------------------------------------------------------------
// loss terms and their reductions (dynamics.py:269-314): parameters of the forward chain's closing phase (chain_loss_phase)
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
    int Dp;                                  // row stride of dMu / dLv / dBmu (D rounded up to 4: zero columns behind D)
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

// ---------------------------------------------------------------------------------------------
// chain kernel: a list of GEMM stages over a 16-row batch tile held in LDS
// ---------------------------------------------------------------------------------------------
// Pointers read out of a stage table are generic to the compiler (flat_load: slower, and it ties the vector-memory
// counter to the LDS one); they all point to device memory, so say so.
typedef __attribute__((address_space(1))) const float* gcptr;
typedef __attribute__((address_space(1))) float* gptr;
__device__ __forceinline__ gcptr as_global(const float* p) { return (gcptr)p; }
__device__ __forceinline__ gptr as_global(float* p) { return (gptr)p; }
#define CH_ROWS 16
#ifndef CADM_Q0_8
#define CADM_Q0_8 8
#endif
#ifndef CADM_Q0_4
#define CADM_Q0_4 12
#endif
#ifndef CADM_Q1_4
#define CADM_Q1_4 4
#endif
#ifndef CADM_Q2_4
#define CADM_Q2_4 2
#endif
// Two flavours of the chain kernel (template parameter NW = waves per workgroup), chosen per launch by the number of work items:
//   NW = 8  ONE workgroup per CU, two waves per SIMD: one wave's LDS / load / scalar work overlaps the other's MFMAs.  The latency
//           flavour: a step at the reference's batch size (256 rows x 5 members x 2 nets = 160 work items) is one partial round of the chip.
//   NW = 4  THREE workgroups per CU (168 registers per wave, 53.6 KB of LDS each): three independent 16-row chains per CU, so one
//           chain's epilogue / barrier / stage start (36 % of a stage, chain_timing) runs beside the others' MFMAs.  The throughput
//           flavour, for launches of more work items than CUs: B = 4096 1.633 -> 1.200 ms per step (0.177 -> 0.241 of the fp32
//           matrix peak), B = 1024 0.410 -> 0.340; at B = 256 it would be 0.144 instead of 0.120 ms (profiles/r5_train_scaling.md).
//           With the work items spread over all eight XCDs (xcd_spread_item: the member-affine mapping left three idle) 0.88 ms = 0.328.
#define CH_WAVES_MAX 8
#define CH_THREADS_MAX (64 * CH_WAVES_MAX)
#define CH_RING 8             // operand blocks (16 k x 2 tiles) of a wave's ring; CH_RING - 1 are in flight
#define CH_MAXSTAGE 20
#define CH_BLK_FLOATS 256     // one operand block of one tile: 64 lanes x float4

// A GEMM stage computes acc[16 rows x N] = src[16 x K] * Bop[K x N] for up to two column SEGMENTS (e.g. the mu and
// logvar heads side by side), each with its own operand stream, bias and outputs.  The operand stream is a packed copy
// of the layer (train_pack_kernel; kept current by dw_adam_kernel's epilogue): per member [k-block][tile][lane] float4,
// tile = 16 output columns, k-block = 16 k, lane (c, kq) holds Bop(16 t + 4 kq + i, 16 tile + c), i = 0..3 -- exactly
// the A operand of four v_mfma_f32_16x16x4_f32 k-steps, zero-padded in both directions.  A wave reads the 2 KB of its
// tile pair per k-block; the workgroup's waves -- and the member's other workgroups, which walk the same stages at the
// same time -- together read one contiguous window of the stream per k-block (all of the L2's channels, not the few
// that per-tile streams a fixed stride apart would hit).  The same load shape in every stage of every chain, forward
// or transposed.
struct ChainSeg {
    const float* P;        // packed operand [E][KB][tiles (even)][64][4]
    long sP;               // member stride (floats)
    const float *bias, *zprev;                           // bias [E][N]; zprev [E][B][ldz]: pre-activation whose act' scales the result
    float *out0, *out1;                                  // [E][B][ldo]: value before act_o / after
    int N, ldo, ldz, vec;                                // vec: N, ldo, ldz, dk0 all multiples of 4 -> 16-byte accesses
    int nt;                                              // tiles of the stream
    int pkB, pkC, pad;                                   // (host, finish of sync_programs) vec | nt << 8;  N | ldz << 16: the lookup of a wave's next
                                                         //  group reads these instead of the four fields (registers: see chain_group)
};
struct ChainStage {
    int KB, src, dst, dk0, act_d, act_o, ntp, tp1;       // KB: k-blocks; ntp: tile pairs (all segments); tp1: first pair of segment 1
    int zfill;                                           // zfill: columns N .. of the last tile are written as zeros
    unsigned char nxt[8];                                // wave slot w's next stage behind this one (index inside the chain; 31: none) | 0x80 if its
                                                         //  pair there (tile pair w) belongs to segment 1 -- filled by finish_chain_table (host)
    int pkA;                                             // (host) KB | src << 8 | tp1 << 16 | ntp << 24
    int pad[4];
    ChainSeg seg[2];
};
struct ChainLoad {        // input tile -> LDS: K columns of g0 (+ g1) [E][B][ld_in] become rows dk0 .. dk0 + K - 1 of buffer dst,
    const float *g0, *g1; // zeros up to row zero_to (the consumer's k loop runs whole 16-row blocks without masking)
    float* gsum;          // echo of the sum [E][B][ldg]
    int ld_in, ldg, K, dst, dk0, zero_to;
    int mode, pad;        // 0: as stored;  1..4: assembled from the raw batch (ChainAsm; formerly assemble_kernel): 1 = obs_preproc of
                          //    the g0 rows (obs or next obs), 2 = action, 3 / 4 = the context encoder's (obs, act) history
};
struct ChainArgs {
    const ChainStage* prog;
    int first[2], count[2];                              // stage range per chain (y)
    ChainLoad pre[2][4]; int npre[2];                    // the chain's inputs (kernel arguments: they are requested before the table is)
    ChainAsm asmp;
    int loss_on, loss_buf, loss_lv0, loss_slots, loss_final;   // forward launch of a training step: losses + head gradients behind the
                                                         //  heads; loss_final: this launch also sums the partials (evaluation)
    LossP lossp; ReduceP lossr;                          //  (head outputs in LDS buffer loss_buf: mu at rows 0.., logvar at rows loss_lv0..)
    int B, bufsz;                                        // rows per member, floats per LDS activation buffer
    int E, ny, ntiles, G, ips;                           // work decomposition, see chain_kernel
    int spread, per_xcd;                                 // spread: XCD x takes the x-th contiguous eighth of the (member-major) work items
    int y_base, slot_ny;                                 // loss phase: this launch's chain y counts as y + y_base of slot_ny (a forward pass split
                                                         //  into one launch per net, launch_forward: same terms, same partial-sum slots as the joint launch)
    unsigned long long* tfine;                           // (same item) wave 0's epilogue, per GEMM stage: [6 si ..] activation math done, LDS tile
                                                         // stored, global stores issued, next group looked up, its operands requested
    unsigned long long* tbuf;                            // cadm_dev_set_timing_buffer: clocks of member 0's first work item:
                                                         // [0..63] stage boundaries, [64 + 4 si ..] wave 0: group start, k loop end,
                                                         // epilogue end, barrier reached
};

typedef __attribute__((address_space(1))) const char* gcbytes;
typedef __attribute__((address_space(1))) const floatx4* gcptr4;
typedef float floatx2 __attribute__((ext_vector_type(2)));

// The operand ring lives in a[0:63], named literally: slot s holds block i (i % 8 == s) of the wave's two tiles in
// a[8 s : 8 s + 3] and a[8 s + 4 : 8 s + 7].  Loads and MFMAs on it are inline asm, for two reasons:
//  * hipcc cannot pipeline loads across a loop back edge (its s_waitcnt placement waits for every outstanding load at the
//    first use behind it), let alone across a stage boundary; asm loads are invisible to its counters and are ordered
//    with explicit `s_waitcnt vmcnt(n)`: vector-memory operations of a wave complete in issue order, so "at most n
//    younger operations outstanding" is exact when the n youngest are ring loads and conservative when compiler-issued
//    accesses (epilogue operands, z / h stores) sit between them;
//  * a ring held in compiler-allocated registers gets MOVED at control-flow merges (the stage loop, the conditional
//    refills): a copy of a register with a load in flight reads stale data, the hardware does not interlock that.
//    Registers the compiler never sees cannot be moved.  (It has no reason to touch AGPRs in this kernel -- the ISA
//    hygiene test checks that it does not.)
// Wait states the hazard recognizer cannot place inside asm (cdna_hip_programming.md 5.7): `s_nop 4` between a
// readfirstlane'd base and the load that reads it, `s_nop 1` between a VALU-written operand and the MFMA (ring_begin), 12 states between
// the last MFMA and the first reader of its accumulator (ring_done).
#define CH_RING_REGS                                                                                                                   \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19",   \
        "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37",  \
        "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55",  \
        "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}
// Values read out of the LDS stage table are wave-uniform, but the compiler cannot know: make them scalar.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ gcbytes uni(gcbytes p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (gcbytes)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ gcbytes uni_ptr(const float* p) { return uni((gcbytes)as_global(p)); }

// LDS activation tile: element (k, row m) at ((k >> 2) * 16 + m) * 4 + (k & 3): the B operand of a k-block is one
// lane-linear ds_read_b128 (lane (m, kq) <- k = 16 t + 4 kq + 0..3), and a D fragment of the transposed product
// (lane (m, q) holds columns 4 q + 0..3 of its tile) is written back as one lane-linear ds_write_b128.
__device__ __forceinline__ int lds_at(int k, int m) { return ((k >> 2) * CH_ROWS + m) * 4 + (k & 3); }

struct ChainGroup {       // one wave's work in one stage: a tile pair
    int si, tp;           // stage, pair index (si < 0: none)
    int KB;
    gcbytes w0;           // block 0 of the pair (tile 1 right behind tile 0)
    int bstep;            // bytes from one k-block to the next
    int sg, nb, src;      // its segment, first column inside the segment, LDS buffer of the stage's input: looked up with the
                          // group (one stage ahead), so that nothing the k loop needs is read out of the table at stage start
    int ntp; unsigned nx; // the stage's tile pairs; ChainStage::nxt of this wave slot: what the NEXT lookup starts from (scalars, no LDS read)
};

// block i of the group -> ring slot S (both tiles); (uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset)
template <int S>
__device__ __forceinline__ void ring_issue(const ChainGroup& g, int i, unsigned loff) {
    gcbytes p0 = g.w0 + (long)i * g.bstep;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 a[%2*8:%2*8+3], %0, %1\n\tglobal_load_dwordx4 a[%2*8+4:%2*8+7], %0, %1 offset:1024"
                 :: "v"(loff), "s"(p0), "n"(S) : "memory", CH_RING_REGS);
}
// the first CH_RING - 1 blocks of a group: issued one stage ahead (before the previous group's stores and the barrier)
template <int S>
__device__ __forceinline__ void ring_prologue_from(const ChainGroup& g, unsigned loff) {
    if (S < g.KB) {
        ring_issue<S>(g, S, loff);
        if constexpr (S + 1 < CH_RING - 1) ring_prologue_from<S + 1>(g, loff);
    }
}
__device__ __forceinline__ void ring_prologue(const ChainGroup& g, unsigned loff) {
    if (g.si >= 0) ring_prologue_from<0>(g, loff);
}
// tail of a group (nothing left to issue): at most `rem` younger blocks may still be outstanding.  Three levels instead of
// seven: a taken scalar branch costs more than the MFMA it delays, and the blocks a coarser wait adds were issued at least
// four block times ago.
__device__ __forceinline__ void wait_blocks(int rem) {
    if (rem >= 4) wait_vmcnt<8>();
    else if (rem >= 2) wait_vmcnt<4>();
    else wait_vmcnt<0>();
}
static_assert(CH_RING == 8, "wait_blocks, the ring's register names and the slot arithmetic assume a ring of 8");
// acc += (weights of ring slot S, tile J, k-step U) x (activation column x): weights are the A operand, so a lane (m, q)
// of D holds columns 4 q + 0..3 of the tile for batch row m
template <int S, int J, int U>
__device__ __forceinline__ void ring_mfma(floatx4& acc, float x) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, a[%2], %1, %0" : "+v"(acc) : "v"(x), "n"(S * 8 + J * 4 + U));
}
// consume block i (slot S): refill the slot freed by block i - 1, wait for block i, 8 MFMAs
template <int S>
__device__ __forceinline__ void ring_step(const ChainGroup& g, int i, unsigned loff, const float* abase, floatx4 (&xa)[4],
                                          floatx4 (&acc)[2]) {
    const int rem = g.KB - 1 - i;
    if (rem >= CH_RING - 1) {
        ring_issue<(S + CH_RING - 1) % CH_RING>(g, i + CH_RING - 1, loff);
        wait_vmcnt<2 * (CH_RING - 1)>();
    } else {
        wait_blocks(rem);
    }
    const int ia = i + 2 < g.KB ? i + 2 : g.KB - 1;
    xa[(S + 2) & 3] = *reinterpret_cast<const floatx4*>(abase + CH_BLK_FLOATS * ia);
    __builtin_amdgcn_sched_barrier(0);     // keep the LDS read two blocks ahead of its use
    const floatx4 x = xa[S & 3];
    ring_mfma<S, 0, 0>(acc[0], x[0]); ring_mfma<S, 1, 0>(acc[1], x[0]);
    ring_mfma<S, 0, 1>(acc[0], x[1]); ring_mfma<S, 1, 1>(acc[1], x[1]);
    ring_mfma<S, 0, 2>(acc[0], x[2]); ring_mfma<S, 1, 2>(acc[1], x[2]);
    ring_mfma<S, 0, 3>(acc[0], x[3]); ring_mfma<S, 1, 3>(acc[1], x[3]);
}
template <int S>
__device__ __forceinline__ void ring_steps(const ChainGroup& g, int i0, unsigned loff, const float* abase, floatx4 (&xa)[4],
                                           floatx4 (&acc)[2]) {
    if (S == 0 || i0 + S < g.KB) {
        ring_step<S>(g, i0 + S, loff, abase, xa, acc);
        if constexpr (S + 1 < CH_RING) ring_steps<S + 1>(g, i0, loff, abase, xa, acc);
    }
}
// VALU-written accumulators (the zeroing moves) -> first MFMA.  The MFMAs' other operands never come out of a VALU
// instruction: weights are written by the ring's loads, activations by ds_read_b128 (tests/test_isa_hygiene.py checks the
// instruction in front of every MFMA of this kernel).
__device__ __forceinline__ void ring_begin(floatx4 (&acc)[2]) { asm volatile("s_nop 1" : "+v"(acc[0]), "+v"(acc[1])); }
__device__ __forceinline__ void ring_done(floatx4 (&acc)[2]) {   // last MFMA -> first VALU read of its accumulator (8-pass op)
    asm volatile("s_nop 11" : "+v"(acc[0]), "+v"(acc[1]));
}

__device__ __forceinline__ ChainGroup group_of(const ChainStage* stg, int si, int tp, int e, int wave) {
    const ChainStage& st = stg[si];
    const int tp1 = uni(st.tp1);
    const int sg = tp >= tp1 ? 1 : 0;
    const ChainSeg& seg = st.seg[sg];
    ChainGroup g;
    g.si = si; g.tp = tp; g.KB = uni(st.KB);
    g.sg = sg; g.nb = 32 * (tp - (sg ? tp1 : 0)); g.src = uni(st.src);
    g.w0 = uni((gcbytes)(as_global(seg.P) + (long)e * seg.sP + (long)(2 * (tp - (sg ? tp1 : 0))) * CH_BLK_FLOATS));
    g.bstep = uni(seg.nt) * (CH_BLK_FLOATS * 4);
    g.ntp = uni(st.ntp); g.nx = (unsigned)uni((int)st.nxt[wave]);
    return g;
}
// the wave's next tile pair behind (si, tp): the next pass of the same stage, else its pair in the next GEMM stage
template <int NW>
__device__ __forceinline__ ChainGroup next_group(const ChainStage* stg, int nst, int si, int tp, int wave, int e) {
    if (si >= 0 && tp + NW < uni(stg[si].ntp)) return group_of(stg, si, tp + NW, e, wave);
    for (int sj = si + 1; sj < nst; ++sj)
        if (wave < uni(stg[sj].ntp)) return group_of(stg, sj, wave, e, wave);
    ChainGroup g;
    g.si = -1; g.tp = 0; g.KB = 0; g.w0 = nullptr; g.bstep = 0; g.sg = 0; g.nb = 0; g.src = 0; g.ntp = 0; g.nx = 31;
    return g;
}

// Epilogue operands of a group (bias, act'(z) source): lane (m, q) needs columns 4 q + 0..3 of both tiles for row m.
// Requested one stage ahead, right before the group's first ring blocks -- so that, in issue order, nothing but ring
// loads follows a ring load and the k loop's vmcnt counts are exact (clamped, never predicated).
struct ChainOps { floatx4 bv[2], zp[2]; };
__device__ __forceinline__ void load_ops(const ChainStage* stg, const ChainGroup& g, int e, int B, int row0, int lane, ChainOps& o) {
    if (g.si < 0) return;
    const ChainStage& st = stg[g.si];
    const int m = lane & 15, q = lane >> 4;
    const int tp1 = uni(st.tp1);
    const int sg = g.tp >= tp1 ? 1 : 0;
    const ChainSeg& seg = st.seg[sg];
    const int nb = 32 * (g.tp - (sg ? tp1 : 0));
    const int N = uni(seg.N), ldz = uni(seg.ldz);
    const bool VEC = uni(seg.vec) != 0;
    gcbytes p_bias = uni_ptr(seg.bias), p_z = uni_ptr(seg.zprev);
    const bool has_z = p_z != nullptr, has_b = p_bias != nullptr;
    const long mrow = (long)e * B;
    const int row = row0 + m, rowc = row < B ? row : B - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n0 = nb + 16 * j + 4 * q;
        if (VEC) {
            const int nc = n0 < N ? n0 : 0;
            gcbytes bb = has_b ? p_bias + ((long)e * N + nc) * 4 : g.w0;
            gcbytes zb = has_z ? p_z + ((mrow + rowc) * ldz + nc) * 4 : g.w0;
            o.bv[j] = *reinterpret_cast<gcptr4>(bb);
            o.zp[j] = *reinterpret_cast<gcptr4>(zb);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + r, nc = n < N ? n : N - 1;
                gcbytes bb = has_b ? p_bias + ((long)e * N + nc) * 4 : g.w0;
                gcbytes zb = has_z ? p_z + ((mrow + rowc) * ldz + nc) * 4 : g.w0;
                o.bv[j][r] = *reinterpret_cast<gcptr>(bb);
                o.zp[j][r] = *reinterpret_cast<gcptr>(zb);
            }
        }
    }
}

// One tile pair of a GEMM stage: k loop over the ring, epilogue.  At the end of the epilogue -- behind this group's
// stores -- the NEXT group's operands and first ring blocks are requested: by the time the stage-end barrier has been
// passed they have landed, so a stage starts with MFMAs instead of an L2 round trip.
template <int NW>
__device__ __forceinline__ void chain_group(const ChainStage* stg, int nst, int wave, const ChainGroup& g, ChainGroup& nxt, ChainOps& ops,
                                            float* bufs, int bufsz, int e, int B, int row0, int lane, unsigned long long* dbg,
                                            unsigned long long* fine) {
    if (dbg) dbg[0] = __builtin_readcyclecounter();
    const ChainStage& st = stg[g.si];
    const int m = lane & 15, q = lane >> 4;
    const unsigned loff = 16u * (unsigned)lane;
    const int sg = g.sg, rtp1 = st.tp1;
    const ChainSeg& seg = st.seg[sg];
    const int nb = g.nb;                                  // first column of the pair inside its segment
    // The epilogue's stage constants are REQUESTED here (plain LDS reads into VGPRs, all independent) and made scalar behind the
    // k loop: read and used in front of it, their two or three dependent LDS round trips delayed every stage's first MFMA.
    const int rN = seg.N, rldo = seg.ldo, rdk0 = st.dk0, rdst = st.dst, ract_d = st.act_d, ract_o = st.act_o, rzf = st.zfill, rvec = seg.vec;
    const float *rbias = seg.bias, *rz = seg.zprev;
    float *ro0 = seg.out0, *ro1 = seg.out1;
    const long mrow = (long)e * B;                        // first row of this member in the [E][B][.] tensors
    const int row = row0 + m;
    const floatx4 bv[2] = {ops.bv[0], ops.bv[1]}, zp[2] = {ops.zp[0], ops.zp[1]};
    const int n0[2] = {nb + 4 * q, nb + 16 + 4 * q};
    floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
    {   // ---- k loop: block i of the pair sits in ring slot i % 8; its loads were issued 7 blocks earlier ----
        const float* abase = bufs + g.src * bufsz + 4 * lane;             // activation block i: + 256 i floats
        floatx4 xa[4];
        xa[0] = *reinterpret_cast<const floatx4*>(abase);
        xa[1] = *reinterpret_cast<const floatx4*>(abase + CH_BLK_FLOATS * (g.KB > 1 ? 1 : 0));
        ring_begin(acc);
#pragma unroll 1
        for (int i0 = 0; i0 < g.KB; i0 += CH_RING) ring_steps<0>(g, i0, loff, abase, xa, acc);
        ring_done(acc);
    }
    if (dbg) dbg[1] = __builtin_readcyclecounter();
    // stage constants -> SGPRs: scalar address bases, uniform branches on the activation kinds
    const int N = uni(rN), ldo = uni(rldo), dk0 = uni(rdk0), dsti = uni(rdst), tp1 = uni(rtp1);
    const int act_d = uni(ract_d), act_o = uni(ract_o), zfill = uni(rzf);
    const bool VEC = uni(rvec) != 0;
    gcbytes p_o0 = uni_ptr(ro0), p_o1 = uni_ptr(ro1);
    const bool has_z = uni_ptr(rz) != nullptr, has_b = uni_ptr(rbias) != nullptr, s0 = p_o0 != nullptr, s1 = p_o1 != nullptr;
    floatx4 v0[2];                                // [tile][r]: before the output activation
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v0[j][r] = acc[j][r] + (has_b ? bv[j][r] : 0.0f);
    if (has_z) {
        if (act_d == ACT_SWISH) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = zp[j][r], sg_ = sigmoid_fast(z);
                    v0[j][r] *= sg_ * (1.0f + z * (1.0f - sg_));
                }
        } else if (act_d == ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v0[j][r] = zp[j][r] > 0.0f ? v0[j][r] : 0.0f;
        } else if (act_d == ACT_TANH) {          // 1 - tanh(z)^2 = 4 s (1 - s), s = sigmoid(2z)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float sg_ = sigmoid_fast(2.0f * zp[j][r]); v0[j][r] *= 4.0f * sg_ * (1.0f - sg_); }
        } else if (act_d == ACT_SIGMOID) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float sg_ = sigmoid_fast(zp[j][r]); v0[j][r] *= sg_ * (1.0f - sg_); }
        }
    }
    // global stores: (uniform base of this member) + 32-bit byte offset (host checks B * ldo * 4 < 2^32).  The value before the output
    // activation is stored as soon as it exists -- not next to the activated one: eight registers fewer are live through the activation math.
    typedef __attribute__((address_space(1))) float* gfp;
    typedef __attribute__((address_space(1))) floatx4* gf4p;
    auto store_rows = [&](gcbytes base, const floatx4 (&v)[2]) {
        if (row >= B) return;
        if (VEC) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (n0[j] >= N) continue;
                *(gf4p)(base + 4u * (unsigned)(row * ldo + n0[j])) = v[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0[j] + r;
                    if (n >= N) continue;
                    *(gfp)(base + 4u * (unsigned)(row * ldo + n)) = v[j][r];
                }
        }
    };
    if (s0) store_rows(p_o0 + mrow * ldo * 4, v0);
    floatx4 v1[2];                                // ... and after
    if (act_o == ACT_SWISH) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v1[j][r] = v0[j][r] * sigmoid_fast(v0[j][r]);
    } else if (act_o == ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v1[j][r] = fmaxf(v0[j][r], 0.0f);
    } else if (act_o == ACT_TANH) {              // as the planner: 2 sigmoid(2z) - 1, odd series near 0 where that cancels
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = v0[j][r], x2 = x * x;
                const float ser = x * fmaf(x2, fmaf(x2, fmaf(x2, -0.05396825396825397f, 0.13333333333333333f), -0.3333333333333333f), 1.0f);
                v1[j][r] = fabsf(x) < 0.1f ? ser : fmaf(2.0f, sigmoid_fast(2.0f * x), -1.0f);
            }
    } else if (act_o == ACT_SIGMOID) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v1[j][r] = sigmoid_fast(v0[j][r]);
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) v1[j] = v0[j];
    }
    // The wave's next group -- the next pass of this stage (tile pair tp + NW), else its pair in the stage the table names (ChainStage::nxt,
    // precomputed on the host) -- is known from scalars that came with THIS group's descriptor, so its whole descriptor is ONE batch of LDS reads,
    // requested here -- behind the activation math, whose registers it would compete for: hipcc parks values in the ring's AGPRs otherwise -- and
    // consumed behind this group's stores, which run under its latency.  (Until round 5 the
    // lookup walked the table behind the stores -- stage's pair count, next stage's, the group's fields, the operands' fields: three to four
    // dependent LDS round trips, 2-3 k of an epilogue's 5-6 k cycles under load, tools/chain_timing.py.)
    const unsigned nx = g.nx;
    const bool same_stage = g.tp + NW < g.ntp;
    const int nsi = same_stage ? g.si : ((nx & 31u) == 31u ? -1 : (int)(nx & 31u));
    const int ntpp = same_stage ? g.tp + NW : wave;
    const int nsg = same_stage ? (ntpp >= tp1 ? 1 : 0) : (int)(nx >> 7);
    struct { int A, B, C, nx; const float *P, *bias, *z; long sP; } rq;
    auto request_next = [&]() {
        const ChainStage& ns = stg[nsi < 0 ? 0 : nsi];
        const ChainSeg& nseg = ns.seg[nsg];
        rq.A = ns.pkA; rq.B = nseg.pkB; rq.C = nseg.pkC; rq.nx = ns.nxt[wave];
        rq.P = nseg.P; rq.bias = nseg.bias; rq.z = nseg.zprev; rq.sP = nseg.sP;
    };
    if constexpr (NW == 8) request_next();
    if (fine) fine[0] = __builtin_readcyclecounter();
    if (dsti >= 0) {
        float* dst = bufs + dsti * bufsz;
        const int sg0 = sg ? 32 * tp1 : 0;                // a second segment's columns follow the first's tile pairs
        if (VEC) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<floatx4*>(dst + lds_at(dk0 + sg0 + n0[j], m)) = n0[j] < N ? v1[j] : floatx4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0[j] + r;
                    if (n < N || zfill) dst[lds_at(dk0 + sg0 + n, m)] = n < N ? v1[j][r] : 0.0f;
                }
        }
    }
    if constexpr (NW != 8) request_next();      // (the 4-wave flavour has 84 VGPRs: requested in front of the activation tile's stores, hipcc parks values in the ring's AGPRs)
    if (fine) fine[1] = __builtin_readcyclecounter();
    if (s1) store_rows(p_o1 + mrow * ldo * 4, v1);
    if (fine) fine[2] = __builtin_readcyclecounter();
    // ---- the wave's next group: descriptor (requested behind the k loop, see above) -> scalars, its epilogue operands and first ring blocks ----
    nxt.si = nsi; nxt.tp = ntpp; nxt.sg = nsg;
    if (nsi >= 0) {
        const int pA = uni(rq.A), pB = uni(rq.B), pC = uni(rq.C);
        const int qtp1 = (pA >> 16) & 255, qN = pC & 0xffff, qldz = (int)((unsigned)pC >> 16);
        const bool qVEC = (pB & 1) != 0;
        const int tp_in = ntpp - (nsg ? qtp1 : 0);                         // pair index inside its segment
        nxt.KB = pA & 255; nxt.src = (pA >> 8) & 255; nxt.nb = 32 * tp_in;
        nxt.w0 = uni((gcbytes)(as_global(rq.P) + (long)e * rq.sP + (long)(2 * tp_in) * CH_BLK_FLOATS));
        nxt.bstep = (pB >> 8) * (CH_BLK_FLOATS * 4);
        nxt.ntp = (int)((unsigned)pA >> 24); nxt.nx = (unsigned)uni(rq.nx);
        gcbytes p_bias = uni_ptr(rq.bias), p_z = uni_ptr(rq.z);
        const bool hz = p_z != nullptr, hb = p_bias != nullptr;
        const int rowc = row < B ? row : B - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c0 = nxt.nb + 16 * j + 4 * q;
            if (qVEC) {
                const int nc = c0 < qN ? c0 : 0;
                gcbytes bb = hb ? p_bias + ((long)e * qN + nc) * 4 : nxt.w0;
                gcbytes zb = hz ? p_z + ((mrow + rowc) * qldz + nc) * 4 : nxt.w0;
                ops.bv[j] = *reinterpret_cast<gcptr4>(bb);
                ops.zp[j] = *reinterpret_cast<gcptr4>(zb);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = c0 + r, nc = n < qN ? n : qN - 1;
                    gcbytes bb = hb ? p_bias + ((long)e * qN + nc) * 4 : nxt.w0;
                    gcbytes zb = hz ? p_z + ((mrow + rowc) * qldz + nc) * 4 : nxt.w0;
                    ops.bv[j][r] = *reinterpret_cast<gcptr>(bb);
                    ops.zp[j][r] = *reinterpret_cast<gcptr>(zb);
                }
            }
        }
    } else {
        nxt.KB = 0; nxt.w0 = nullptr; nxt.bstep = 0; nxt.nb = 0; nxt.src = 0; nxt.ntp = 0; nxt.nx = 31;
    }
    if (fine) fine[3] = fine[4] = __builtin_readcyclecounter();      // (one stamp for both: a per-thread branch inside the uniform one above makes hipcc treat the group's scalars as per-lane values)
    ring_prologue(nxt, loff);
    if (dbg) dbg[2] = __builtin_readcyclecounter();
}

// The chain's input tiles.  Element idx of a tile = (16-column block, row, column in the block): a wave covers 4 rows x 16
// columns -- 4 segments of 64 bytes per load (the LDS-linear order, 16 rows x 4 columns, costs 16 segments per load and
// made this prologue slower than the separate assembly kernel it replaces), a 4-way bank conflict on the LDS side.
// Two phases, so that the loads of ALL of a chain's tiles (and the stage table's) are in flight together: fetch requests the
// raw operands of a tile's elements -- the value, and for the assembled tiles its mean and std --, commit turns them into
// inputs.  One global round trip for the whole kernel prologue instead of one per tile and operand.  A tile has ONE source
// per operand (the assembled inputs are two tiles each: observation columns, action columns), so an element's addresses are
// base + column: nothing for hipcc to branch on between the loads.
template <int NU>
struct ChainIn { float x[NU], a[NU]; };      // (the third operand -- the std of an assembled column -- is fetched at commit time: an L2 hit by then,
                                             //  and a third fewer registers per element in flight across the one HBM round trip)
struct ChainInSrc {
    gcptr x0, a0, b0;
    int hc;                   // half-cheetah obs_preproc (columns 0..2 <- o[1], sin o[2], cos o[2])
    int shift;                // ant obs_preproc: column f <- o[f + 1]
    bool rok, two;
    long grow;
};
__device__ __forceinline__ ChainInSrc chain_input_src(const ChainLoad& d, const ChainAsm& ap, int e, int B, int row0, int tid) {
    ChainInSrc r;
    const int row = row0 + ((tid >> 4) & 15);                 // (a thread's elements are 256 apart: it keeps its row)
    r.rok = row < B;
    r.grow = (long)e * B + (r.rok ? row : 0);
    long srow = r.grow, swin = r.grow;
    if (d.mode) map_row(ap.map, r.grow, srow, swin);
    r.hc = 0; r.shift = 0; r.two = false;
    if (d.mode == 0) {
        r.x0 = as_global(d.g0) + r.grow * d.ld_in;
        r.two = d.g1 != nullptr;
        r.a0 = r.two ? as_global(d.g1) + r.grow * d.ld_in : r.x0;
        r.b0 = r.x0;
    } else if (d.mode == 1) {          // preprocessed observation columns of (next) obs rows
        r.x0 = as_global(d.g0) + srow * ap.D; r.a0 = as_global(ap.obs_mean); r.b0 = as_global(ap.obs_std);
        r.hc = ap.env == CADM_ENV_HALFCHEETAH;
        r.shift = ap.env == CADM_ENV_ANT;
    } else if (d.mode == 2) {          // action columns
        r.x0 = as_global(ap.act) + srow * ap.A; r.a0 = as_global(ap.act_mean); r.b0 = as_global(ap.act_std);
    } else if (d.mode == 3) {          // context encoder: observation history
        r.x0 = as_global(ap.cp_obs) + swin * ap.ncpo; r.a0 = as_global(ap.cp_obs_mean); r.b0 = as_global(ap.cp_obs_std);
    } else {                           // context encoder: action history
        r.x0 = as_global(ap.cp_act) + swin * ap.ncpa; r.a0 = as_global(ap.cp_act_mean); r.b0 = as_global(ap.cp_act_std);
    }
    return r;
}
template <int NT, int NU>
__device__ __forceinline__ void chain_input_fetch(const ChainLoad& d, const ChainInSrc& r, int base, int tid, ChainIn<NU>& q) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int idx = base + u * NT + tid;
        const int k = (idx >> 8) * 16 + (idx & 15);
        const int j = k < d.K ? k : 0;
        const int jx = r.hc ? (j == 0 ? 1 : j <= 2 ? 2 : j) : j + r.shift;     // preproc_at's source column
        q.x[u] = r.x0[jx];
        q.a[u] = r.a0[j];
    }
}
template <int NT, int NU>
__device__ __forceinline__ void chain_input_commit(const ChainLoad& d, const ChainInSrc& r, int base, int tid, const ChainIn<NU>& q,
                                                   float* bufs, int bufsz) {
    float* dst = bufs + d.dst * bufsz;
    const int m = (tid >> 4) & 15;
    float sd[NU];
    if (d.mode != 0) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = base + u * NT + tid;
            const int k = (idx >> 8) * 16 + (idx & 15);
            sd[u] = r.b0[k < d.K ? k : 0];
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int idx = base + u * NT + tid;
        const int k = (idx >> 8) * 16 + (idx & 15);
        const bool ok = k < d.K && r.rok;
        float x;
        if (d.mode == 0) {
            x = r.two ? q.x[u] + q.a[u] : q.x[u];
        } else {
            float t = q.x[u];
            if (u == 0 && r.hc && base == 0) {              // (columns 1 and 2 only exist in a thread's first element; hc: mode 1)
                if (k == 1) t = sinf(t);
                else if (k == 2) t = cosf(t);
            }
            x = (t - q.a[u]) / (sd[u] + 1e-10f);
        }
        x = ok ? x : 0.0f;
        if (d.dk0 + k < d.zero_to) dst[lds_at(d.dk0 + k, m)] = x;
        if (ok && d.gsum) as_global(d.gsum)[r.grow * d.ldg + k] = x;
    }
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
// More work items than the affine mapping's XCDs can hold in one round (large batches): that mapping leaves 8 - G E XCDs idle -- three
// of eight for the 5-member ensemble, found in round 5 with the per-item clocks: an item took 160 k cycles, the launch 6 rounds of them.
// Then XCD x takes the x-th CONTIGUOUS eighth of the member-major item list instead (as dw_adam_kernel does): every XCD is busy, and
// its L2 still holds the weights of at most two members (E <= 8).
__device__ __forceinline__ bool xcd_spread_item(int E, int per_xcd, int per, int& e, int& item) {
    const int g = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (g >= E * per) return false;
    e = g / per;
    item = g - e * per;
    return true;
}

// Sums of the workgroups' partials in a fixed order (G groups of threads take contiguous chunks of slots -- loads eight at a
// time: one after the other they are 160 dependent round trips, + 48 us measured --, then the chunks are added in order), the
// three reported losses, and Adam on max / min_logvar (data term + the 0.01 regulariser of dynamics.py:308).
template <int NT, bool COHERENT>
__device__ __forceinline__ void loss_finalize(const ReduceP& r, int slots, float* scr, int tid) {
    const int D = r.D, NQ = 4 + 2 * D;
    float* red = r.out;
    const int W = NQ < NT ? NQ : NT, G = NT / W, CS = (slots + G - 1) / G;
    for (int q0 = 0; q0 < NQ; q0 += NT) {
        const int q = q0 + tid % W, g = tid / W;
        if (g < G && q < NQ) {
            float v = 0.0f;
            const int w1 = (g + 1) * CS < slots ? (g + 1) * CS : slots;
            for (int w0 = g * CS; w0 < w1; w0 += 8) {
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* src = r.part + (size_t)(w0 + u < w1 ? w0 + u : w1 - 1) * NQ + q;
                    x[u] = COHERENT ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) v += w0 + u < w1 ? x[u] : 0.0f;
            }
            scr[g * W + (q - q0)] = v;
        }
        __syncthreads();
        if (tid < W && q0 + tid < NQ) {
            float tot = 0.0f;
            for (int gg = 0; gg < G; ++gg) tot += scr[gg * W + tid];
            red[q0 + tid] = tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float mse = red[0], mu_loss = red[1], var_loss = red[2], back = red[3];
        float recon = r.det ? mse : mu_loss + var_loss;
        if (r.has_back) recon += r.back_coeff * back;
        r.losses_out[0] = mse;
        r.losses_out[1] = r.has_back ? back : 0.0f;
        r.losses_out[2] = recon;
    }
    if (r.adam_mm && tid < 2 * D) {
        const bool mx = tid < D;
        const int d = mx ? tid : tid - D;
        float* w = (mx ? r.maxlv : r.minlv) + d;
        float* m = (mx ? r.mx_m : r.mn_m) + d;
        float* v = (mx ? r.mx_v : r.mn_v) + d;
        float ww = *w, mm = *m, vv = *v;
        adam_update(ww, mm, vv, red[4 + tid] + (mx ? 0.01f : -0.01f), r.lr_t, r.b1, r.b2, r.eps);
        *w = ww; *m = mm; *v = vv;
    }
}

// Closing phase of the forward launch of a training step: the workgroup's 16 rows x D head outputs are still in LDS, so the
// loss terms, the head gradients and the workgroup's share of every reduction are taken here instead of in a launch of their
// own (9 us of pure latency).  Forward-net workgroups own terms {mse, mu_loss, var_loss, d/d max_logvar,
// d/d min_logvar}, backward-model workgroups back_mse.  Reductions in a fixed order throughout: a workgroup's partials
// (rows ascending), then -- by the workgroup that arrives last -- all partials in slot order: no float atomics, the result
// does not depend on which workgroup is last.  That one also finalises (losses_out, Adam on max / min_logvar), exactly as
// the separate loss / reduction launch of earlier rounds did.
// The normalised target of a thread's FIRST element (el = tid; the only one when 16 D <= 512): requested in the kernel prologue,
// a whole forward pass before it is needed -- its memory latency used to sit at the end of the launch.
__device__ __forceinline__ float chain_loss_target(const LossP& p, int e, int y, int row0, int el) {
    const int D = p.D, m = el / D, d = el - m * D, row = row0 + m;
    if (el >= CH_ROWS * D || row >= p.B) return 0.0f;
    long srow, swin;
    map_row(p.map, (long)e * p.B + row, srow, swin);
    const long si = srow * D + d;                                      // this element in the caller's target tensors
    return y == 0 ? (p.delta[si] - p.dmean[d]) / (p.dstd[d] + 1e-10f) : (p.back_delta[si] - p.bdmean[d]) / (p.bdstd[d] + 1e-10f);
}

template <int NW>
__device__ __forceinline__ void chain_loss_phase(const ChainArgs& a, float* bufs, float* scr, int e, int y, int row0, int tid, float tgt0) {
    constexpr int CH_THREADS = 64 * NW;
    const LossP& p = a.lossp;
    const ReduceP& r = a.lossr;
    const int D = p.D, B = p.B, nel = CH_ROWS * D, lane = tid & 63, wave = tid >> 6;
    const float* hb = bufs + a.loss_buf * a.bufsz;
    for (int el = tid; el < nel; el += CH_THREADS) {
        const int m = el / D, d = el - m * D, row = row0 + m;
        float tm[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (row < B) {
            const long grow = (long)e * B + row, i = grow * p.Dp + d;
            const float s = 1.0f / ((float)B * (float)D);             // reduce_mean over b then d; reduce_sum over e
            const float mu = hb[lds_at(d, m)];
            const float tgt = el == tid ? tgt0 : chain_loss_target(p, e, y, row0, el);     // (normalised target)
            if (y == 0) {
                const float t = tgt;
                const float diff = mu - t;
                tm[0] = diff * diff * s;                                                  // mse            (:273-274)
                if (p.det) {
                    p.dMu[i] = 2.0f * s * diff;
                    p.dLv[i] = 0.0f;
                } else {
                    const float mx = p.maxlv[d], mn = p.minlv[d], lv0 = hb[lds_at(a.loss_lv0 + d, m)];
                    const float u = mx - tf_softplus(mx - lv0);                           // core/utils.py:356
                    const float lvc = mn + tf_softplus(u - mn);                           // core/utils.py:357
                    const float invvar = expf(-lvc);                                      // :303
                    tm[1] = diff * diff * invvar * s;                                     // mu_loss        (:304-305)
                    tm[2] = lvc * s;                                                      // var_loss       (:306-307)
                    const float g_lvc = s * (1.0f - diff * diff * invvar);
                    const float s1 = sigmoidf_(u - mn), s2 = sigmoidf_(mx - lv0);         // softplus' = sigmoid
                    p.dMu[i] = 2.0f * s * diff * invvar;
                    p.dLv[i] = g_lvc * s1 * s2;
                    // 1 - sigmoid(x) = sigmoid(-x), evaluated as such: with min_logvar = -10 the factor is ~5e-5 and `1 - s1`
                    // would keep 3 of its digits (the autodiff graph's g - g s1 does cancel like that; this is the exact value)
                    tm[4] = g_lvc * s1 * sigmoidf_(lv0 - mx);                             // d / d max_logvar (without the 0.01 reg)
                    tm[5] = g_lvc * sigmoidf_(mn - u);                                    // d / d min_logvar
                }
            } else {
                const float tb = tgt;
                const float db = mu - tb;
                tm[3] = db * db * s;                                                      // back_mse       (:280-281)
                p.dBmu[i] = p.back_coeff * 2.0f * s * db;
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) scr[q * nel + el] = tm[q];
    }
    __syncthreads();
    const int NQ = 4 + 2 * D;
    const int slot = (e * a.slot_ny + y) * a.ntiles + row0 / CH_ROWS;      // (y: the caller passes y + y_base)
    float* part = r.part + (size_t)slot * NQ;
    if (wave < 4) {                                                // scalar terms: wave q sums scr[q][*]
        float v = 0.0f;
        for (int j = lane; j < nel; j += 64) v += scr[wave * nel + j];
        v = wave_sum_fixed(v);
        if (lane == 0) {
            if (a.loss_final) __hip_atomic_store(part + wave, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else part[wave] = v;
        }
    }
    {                                                              // per-dim terms: one thread per (bound, dim), rows ascending
        for (int o = (NW > 4 ? tid - 256 : tid); o >= 0 && o < 2 * D; o += 256) {
            const int which = o / D, d = o - which * D;
            float v = 0.0f;
            for (int m = 0; m < CH_ROWS; ++m) v += scr[(4 + which) * nel + m * D + d];
            if (a.loss_final) __hip_atomic_store(part + 4 + o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else part[4 + o] = v;
        }
    }
    // Training step: the partials are ordinary stores; the sums are taken by a spare workgroup of the weight-gradient launch
    // (loss_finalize in dw_adam_kernel: the kernel boundary orders the two, nothing waits for anybody, and the reduction is off
    // the step's critical path).  Evaluation (no further launch): hand-off to the last workgroup WITHOUT an agent-scope fence --
    // a release fence writes back the XCD's whole L2, which at this point holds the megabytes of z / h the chain has just
    // stored (measured: + 48 us on the launch).  There the partials are device-coherent stores (sc1: written through, past
    // the non-coherent L2s) that have completed (vmcnt) before the arrival counter is bumped, and the last workgroup reads them
    // with device-coherent loads.  This is the "sc1 payload -> asm vmcnt(0) -> agent atomic flag / sc1 loads on the consumer" form
    // MI355X_MICROARCH.md lists as valid for gfx950 (handoff-flag, "drained sc1"); it is a statement about THIS target, which is
    // the only one the library is built for (Makefile: ARCH = gfx950), not about the HIP memory model in general.  Compiler side:
    // the asm wait carries a "memory" clobber and both __syncthreads() are workgroup fences, so no access moves across them.
    // tests/test_gpu_train.py::test_eval_losses_equal_the_training_steps_reduction pins the result (bit-equal to the two-launch
    // reduction, under load, many repetitions).
    if (!a.loss_final) return;
    int* const flag = reinterpret_cast<int*>(scr + (6 * nel > CH_THREADS ? 6 * nel : CH_THREADS));     // (launch_chain sizes scr)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *flag = atomicInc(r.counter, a.loss_slots - 1) == (unsigned)(a.loss_slots - 1);     // wraps back to 0 for the next step
    __syncthreads();
    if (!*flag) return;
    loss_finalize<256, true>(r, a.loss_slots, scr, tid);      // (256 threads' chunking in BOTH flavours -- and in dw_adam_kernel's copy: the same sums, bit for bit)
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 3) void chain_kernel(const ChainArgs a) {
    constexpr int CH_THREADS = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float chain_smem[];
    ChainStage* const stg = reinterpret_cast<ChainStage*>(chain_smem);
    float* const bufs = chain_smem + (CH_MAXSTAGE * sizeof(ChainStage)) / sizeof(float);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    // The argument block is ~1 KB (input tiles, assembly pointers): read where it is used, its 64-byte lines arrive one
    // dependent scalar-memory round trip after the other (4.7 k cycles of prologue were measured that way even for the
    // smallest chain).  Touch every line now, in one batch.
    {
        typedef __attribute__((address_space(4))) const int* kargp;
        kargp ka = (kargp)__builtin_amdgcn_kernarg_segment_ptr();
        int sink = 0;
#pragma unroll
        for (unsigned o = 0; o < sizeof(ChainArgs); o += 64) sink += ka[o / 4];
        asm volatile("" ::"s"(sink));
    }
    int e, item;
    const int per = a.ntiles * a.ny;
    if (a.spread ? !xcd_spread_item(a.E, a.per_xcd, per, e, item) : !xcd_affine_item(a.E, a.G, a.ips, per, e, item)) return;
    const int y = item / a.ntiles, row0 = (item - y * a.ntiles) * CH_ROWS, B = a.B;
    const int nst = y ? a.count[1] : a.count[0];
    const bool timed0 = a.tbuf && item == 0 && e == 0 && tid == 0;
    if (timed0) a.tbuf[200] = __builtin_readcyclecounter();
    // stage table of this chain -> LDS (one memory latency instead of one per stage); requested first, stored behind the
    // input tiles, which are described by kernel arguments and so are on their way before the table has arrived
    constexpr int TW = (CH_MAXSTAGE * (int)(sizeof(ChainStage) / sizeof(int)) + CH_THREADS - 1) / CH_THREADS;
    int tv[TW];
    const int nw = nst * (int)(sizeof(ChainStage) / sizeof(int));
    {
        const int* g = reinterpret_cast<const int*>(a.prog + (y ? a.first[1] : a.first[0]));
#pragma unroll
        for (int u = 0; u < TW; ++u) {
            const int i = tid + u * CH_THREADS;
            tv[u] = g[i < nw ? i : 0];
        }
    }
    {
        // (descriptors by value at static kernel-argument offsets: indexing them with the runtime y makes every field access
        //  a scalar memory load of its own -- 13 k cycles of prologue were measured that way)
        const int np = y ? a.npre[1] : a.npre[0];
        const ChainLoad d0 = y ? a.pre[1][0] : a.pre[0][0], d1 = y ? a.pre[1][1] : a.pre[0][1], d2 = y ? a.pre[1][2] : a.pre[0][2],
                        d3 = y ? a.pre[1][3] : a.pre[0][3];
        // elements per thread requested in one go -- two registers each: value and mean (or second summand); the std follows at commit
        // time --: 8 / 4 / 2 / 2 (8 waves: 128 / 64 / 32 / 32 columns) and 12 / 4 / 2 / 2 (4 waves: 192 / 64 / 32 / 32): the reference's
        // input tiles (180 + 60 history columns, 20 + 6) in ONE round trip to HBM; wider tiles loop.  The 4-wave flavour's 84 VGPRs do not
        // hold that: hipcc parks values in AGPRs here -- harmless in front of the first ring load, and only there
        // (tests/test_isa_hygiene.py checks from the first ring load on).
        constexpr int Q0 = NW == 8 ? CADM_Q0_8 : CADM_Q0_4, Q1 = NW == 8 ? 4 : CADM_Q1_4, Q2 = NW == 8 ? 2 : CADM_Q2_4;
        ChainIn<Q0> q0;
        ChainIn<Q1> q1;
        ChainIn<Q2> q2, q3;
        const ChainInSrc r0 = chain_input_src(d0, a.asmp, e, B, row0, tid), r1 = chain_input_src(np > 1 ? d1 : d0, a.asmp, e, B, row0, tid),
                         r2 = chain_input_src(np > 2 ? d2 : d0, a.asmp, e, B, row0, tid), r3 = chain_input_src(np > 3 ? d3 : d0, a.asmp, e, B, row0, tid);
        chain_input_fetch<CH_THREADS>(d0, r0, 0, tid, q0);
        if (np > 1) chain_input_fetch<CH_THREADS>(d1, r1, 0, tid, q1);
        if (np > 2) chain_input_fetch<CH_THREADS>(d2, r2, 0, tid, q2);
        if (np > 3) chain_input_fetch<CH_THREADS>(d3, r3, 0, tid, q3);
        if (timed0) a.tbuf[201] = __builtin_readcyclecounter();
        chain_input_commit<CH_THREADS>(d0, r0, 0, tid, q0, bufs, a.bufsz);
        if (np > 1) chain_input_commit<CH_THREADS>(d1, r1, 0, tid, q1, bufs, a.bufsz);
        if (np > 2) chain_input_commit<CH_THREADS>(d2, r2, 0, tid, q2, bufs, a.bufsz);
        if (np > 3) chain_input_commit<CH_THREADS>(d3, r3, 0, tid, q3, bufs, a.bufsz);
        for (int i = 0; i < np; ++i) {                        // the rest of wide tiles, one round trip per 32 columns
            const ChainLoad& d = i == 0 ? d0 : i == 1 ? d1 : i == 2 ? d2 : d3;
            const ChainInSrc& r = i == 0 ? r0 : i == 1 ? r1 : i == 2 ? r2 : r3;
            const int done = (i == 0 ? Q0 : i == 1 ? Q1 : Q2) * CH_THREADS;
            // (four elements per thread and round trip: one at a time, the 4-wave flavour took 13 dependent round trips for the context
            //  encoder's 240-column tile -- 30 k cycles of prologue under load, tools/chain_timing.py)
            constexpr int QR = 4;
            ChainIn<QR> qr;
            for (int base = done; base < ((d.zero_to - d.dk0 + 15) & ~15) * CH_ROWS; base += QR * CH_THREADS) {
                chain_input_fetch<CH_THREADS>(d, r, base, tid, qr);
                chain_input_commit<CH_THREADS>(d, r, base, tid, qr, bufs, a.bufsz);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < TW; ++u)
        if (tid + u * CH_THREADS < nw) reinterpret_cast<int*>(stg)[tid + u * CH_THREADS] = tv[u];
    if (timed0) a.tbuf[202] = __builtin_readcyclecounter();
    __syncthreads();
    const bool timed = a.tbuf && item == 0 && e == 0 && tid == 0;
    if (timed) a.tbuf[0] = __builtin_readcyclecounter();
    const float tgt0 = a.loss_on ? chain_loss_target(a.lossp, e, y + a.y_base, row0, tid) : 0.0f;
    ChainOps ops;
    ChainGroup cur = next_group<NW>(stg, nst, -1, 0, wave, e);
    load_ops(stg, cur, e, B, row0, lane, ops);
    ring_prologue(cur, 16u * (unsigned)lane);
    for (int si = 0; si < nst; ++si) {
        while (cur.si == si) {
            ChainGroup nxt;
            unsigned long long* dbg = timed ? a.tbuf + 64 + si * 4 : nullptr;
            chain_group<NW>(stg, nst, wave, cur, nxt, ops, bufs, a.bufsz, e, B, row0, lane, dbg, timed ? a.tfine + si * 6 : nullptr);
            cur = nxt;
        }
        if (timed) a.tbuf[64 + si * 4 + 3] = __builtin_readcyclecounter();
        __syncthreads();
        if (timed) a.tbuf[si + 1] = __builtin_readcyclecounter();
    }
    // (the loss terms' scratch: an activation buffer the chain is done with -- the head outputs sit in loss_buf, the other two are dead;
    //  a region of its own behind the buffers cost the third workgroup per CU its LDS)
    if (a.loss_on) chain_loss_phase<NW>(a, bufs, bufs + ((a.loss_buf + 1) % 3) * a.bufsz, e, y + a.y_base, row0, tid, tgt0);
}

// ---------------------------------------------------------------------------------------------
// grouped weight-gradient GEMM with the Adam update fused into its epilogue
// ---------------------------------------------------------------------------------------------
struct DwJob {                       // W[e] (M x N) <- Adam(W, X[e]^T dZ[e] + wdc W);  b[e] <- Adam(b, colsum dZ[e])
    const float *X, *dZ;             // X [E][B][ldx] (first M columns), dZ [E][B][N]
    const float* dZ2;                // optional second gradient, added on load (the context encoder's, from the two dynamics nets)
    float *W, *Mw, *Vw, *bW, *bM, *bV;
    int ldx, M, N, tile0;            // tile0: first workgroup (blockIdx.x) of this job
    int ldz, pad;                    // row stride of dZ (>= N: the head / context gradients are stored with rows padded to 16 bytes)
    float wdc;
    int tn;                          // column tiles
    PackDst pf, pb;                  // the chain kernel's packed copies of W (forward / transposed operand), kept current here
};
#define DW_MAXJOBS 20
struct DwArgs {
    int tile0s[DW_MAXJOBS];          // the jobs' first tiles again, compact: the job search reads two 64-byte lines of the argument block
                                     // instead of one line per job it steps over (each a dependent scalar-memory round trip)
    DwJob job[DW_MAXJOBS];
    int njobs, B, tiles, E;          // tiles: work items per member
    float lr_t, b1, b2, eps;
    ReduceP lossr; int loss_slots;   // the step's loss partials (chain_loss_phase), summed by a spare workgroup of this launch
    unsigned long long* tbuf;        // cadm_dev_set_timing_buffer (tools/chain_timing.py): per workgroup [1024 + 2 b] start / end,
                                     // [4096 + b] job and flavour, [5200 + b] end of the slab loop (s_memrealtime, 100 MHz)
};
static_assert(sizeof(DwArgs) <= 4096, "kernel argument block");

#define TN 64
#define TM 48
#define TK 32
#define LDA (TM + 4)
#define LDB (TN + 4)
#ifndef CADM_DW_EXPERIMENT
#define CADM_DW_EXPERIMENT 0
#endif
#define DW_NSLAB 1                   // slabs per K panel in flight (registers): one keeps the kernel at 120 VGPRs = 4 workgroups per CU

// One 48 x 64 tile of one job per workgroup, reduction over the batch: the whole step's ~925 tiles then fit the chip's
// 1024 workgroup slots (4 per CU) in ONE round (32 x 64 tiles needed 1330 = two rounds).  K is walked in 32-deep slabs:
// stash the slab's loads into LDS as they land, barrier, MFMA sweep; 4 waves side by side, each 48 x 16.  Occupancy beats
// panel depth here: 2-slab panels (156 VGPRs, 3 per CU) and register double-buffering (178 VGPRs) both measured slower.
__global__ __launch_bounds__(256) void dw_adam_kernel(const DwArgs a) {
    constexpr int LDK = TK + 4;                          // fast path: slabs stored [feature][k]
    constexpr int DW_SLAB = (TM + TN) * LDK;             // ... in TWO buffers, so that a slab costs one barrier (see the slab loop)
    constexpr int DW_SMEM = DW_NSLAB * TK * (LDA + LDB) > 2 * DW_SLAB ? DW_NSLAB * TK * (LDA + LDB) : 2 * DW_SLAB;
    __shared__ __attribute__((aligned(16))) float dw_smem[DW_SMEM];
    float* const As = dw_smem;
    float* const Bs = dw_smem + DW_NSLAB * TK * LDA;
    constexpr int LDC = TN + 4;                          // the finished tile, staged for the vectorised Adam epilogue
    static_assert(TM * LDC <= DW_NSLAB * TK * (LDA + LDB), "the C tile must fit the slab buffers");
    // Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), each with its own L2.  Consecutive work items
    // (member-major, then job, then tile) re-read the same X / dZ panels, so XCD x gets the x-th CONTIGUOUS eighth of them:
    // a panel is then fetched into one L2 instead of up to eight (W, m, v alone are 44 MB of traffic per launch; measured later: the placement of the panels makes no difference).
    if (blockIdx.x >= gridDim.x - 8) {                   // (eight spare workgroups keep the XCD arithmetic below; one works)
        if (blockIdx.x == gridDim.x - 8 && a.loss_slots > 0) loss_finalize<256, false>(a.lossr, a.loss_slots, dw_smem, threadIdx.x);
        return;
    }
    const int per_xcd = (a.tiles * a.E + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= a.tiles * a.E) return;
    const int e = item / a.tiles, tile = item - e * a.tiles;
#ifdef CADM_DW_TIMING       // (developer build only, tools/chain_timing.py: the stamps cost registers -- 132 VGPRs = 3 workgroups per CU)
    const bool tstamp = a.tbuf && threadIdx.x == 0 && blockIdx.x < 1000;
    if (tstamp) a.tbuf[1024 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < a.njobs && tile >= a.tile0s[ji + 1]) ++ji;
    const DwJob& jb = a.job[ji];
    const int t = tile - jb.tile0;
    const int mb = (t / jb.tn) * TM, nb = (t % jb.tn) * TN;
    const int M = jb.M, N = jb.N, K = a.B;
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const float* A = jb.X + (long)e * K * jb.ldx;       // A(m = k_in, k = b) = X[b][k_in]
    const float* Bm = jb.dZ + (long)e * K * jb.ldz;     // B(k = b, n)        = dZ[b][n]
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
    const bool two = jb.dZ2 != nullptr;
    const long d2 = two ? jb.dZ2 - jb.dZ : 0;            // (same shape and member stride as dZ)
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
            const long o = (long)(k < kmax ? k : kmax) * jb.ldz;
            const float v1 = pb[it][o], v2 = pb[it][o + d2];        // (both loads unconditional: d2 = 0 without a second gradient)
            rb[0][it] = two ? v1 + v2 : v1;
        }
    };
    // Fast path (whole slabs, 16-byte rows): a slab is fetched with 16-byte loads -- a lane takes 4 consecutive features of one
    // batch row, 8 lanes 128 contiguous bytes of it (lanes along the batch instead -- conflict-free stores without a swizzle --
    // fetch a 64-byte line per 16 bytes used: 0.225 ms per step) -- and stashed TRANSPOSED ([feature][k], k contiguous), so that
    // an MFMA operand for 4 k-steps is one ds_read_b128: per slab and wave 4 global loads, 14 LDS writes and
    // 8 LDS reads next to the 24 MFMAs, where the generic path below spends 14 + 14 + 32 and a clamp / select per element.
    // (On this part the matrix pipe does not overlap with another wave's VALU work: every instruction saved is MFMA time.)
    // k-steps are taken in the order k = 16 g + 4 q + u (lane group q, u = 0..3) -- any order, as long as A and B agree.
    // (rows are read in 16-byte pieces up to the next multiple of 4 columns: the workspace pads the odd-width tensors -- the
    //  normalised inputs, the head and context gradients -- with zero columns, so that the few jobs on them do not fall back to
    //  the scalar loop: they were the launch's tail, 25-28 us of slab loop against 13-17)
    const int Mq = (M + 3) & ~3, Nq = (N + 3) & ~3;
    const bool vec = KP > 0 && (K % TK) == 0 && ((jb.ldx | jb.ldz) & 3) == 0 && Mq <= jb.ldx && Nq <= jb.ldz &&
                     ((reinterpret_cast<size_t>(jb.X) | reinterpret_cast<size_t>(jb.dZ) | reinterpret_cast<size_t>(jb.dZ2)) & 15) == 0;
#ifndef CADM_DW_NO_DMA
    // Slabs by LDS-DMA (one gradient source; the jobs that add a second one on load keep the register path below): a slab goes
    // global -> LDS in 16 buffer_load_dwordx4 .. lds of the workgroup (4 per wave: 4 batch rows x 12 / 16 quads each), no registers, no
    // ds_write, in the tensors' own [row][feature] order; operands are then single dwords (lane (c, q): feature c of row q of a 4-row
    // group).  The 4 rows of one MFMA come from 4 DIFFERENT groups -- each group starts 16 floats further round the banks -- so the
    // four lane groups of a ds_read hit four different quarter-sets of banks: k-steps are taken in the order (t, r) -> rows
    // {4 (4 t + q) + r : q = 0..3}, any order as long as A and B agree.
    const bool dma = vec && !two && (size_t)K * jb.ldx * 4 < (1ull << 32) && (size_t)K * jb.ldz * 4 < (1ull << 32);
    if (dma) {
        constexpr int AG = 4 * TM + 16, BG = 4 * TN + 16, DBUF = 8 * (AG + BG);      // floats per 4-row group of A / B, per slab buffer
        static_assert(TK == 32 && 2 * DBUF <= DW_SMEM && TM * LDC <= DBUF, "slab buffers of the LDS-DMA path");
        typedef __attribute__((address_space(3))) void* ldsp;
        const int c = lane & 15, kq = lane >> 4, wu = __builtin_amdgcn_readfirstlane(wn);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)K * jb.ldx * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bm, 0, (unsigned)((size_t)K * jb.ldz * 4), 0x00020000);
        const int ar = lane / 12, aq = lane - 12 * ar;                               // (lanes 0..47: 4 rows x 12 quads of A)
        const int ma = mb + 4 * aq < Mq ? mb + 4 * aq : Mq - 4, nq = nb + 4 * c < Nq ? nb + 4 * c : Nq - 4;
        const unsigned va = (unsigned)((ar * jb.ldx + ma) * 4), vb = (unsigned)((kq * jb.ldz + nq) * 4);
        const bool n_ok = nb + 16 * wn < N;
        auto request = [&](int k0, int par) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * wu + u;
                float* const ga = dw_smem + par * DBUF + j * AG;
                float* const gb = dw_smem + par * DBUF + 8 * AG + j * BG;
                const unsigned sa = (unsigned)(k0 + 4 * j) * (unsigned)jb.ldx * 4u, sb = (unsigned)(k0 + 4 * j) * (unsigned)jb.ldz * 4u;
                if (lane < 48) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (ldsp)ga, 16, va, sa, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (ldsp)gb, 16, vb, sb, 0, 0);
            }
        };
        auto compute = [&](int par) {
            const float* Ab = dw_smem + par * DBUF + kq * AG + c;
            const float* Bb = dw_smem + par * DBUF + 8 * AG + kq * BG + 16 * wn + c;
            if (do_colsum) {
                const float* Bc = dw_smem + par * DBUF + 8 * AG + tid;
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) colsum += Bc[j * BG + r * TN];
            }
            if (n_ok) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float b = Bb[4 * t * BG + r * TN];
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            if (mb + 16 * i >= M) continue;
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ab[4 * t * AG + r * TM + 16 * i], b, acc[i], 0, 0, 0);
                        }
                    }
            }
        };
        request(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int par = 0;
        for (int k0 = 0; k0 < KP; k0 += TK, par ^= 1) {
            if (k0 + TK < KP) request(k0 + TK, par ^ 1);       // (the buffer computed from one iteration ago: every wave is past that iteration's barrier)
            compute(par);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (vec && !dma) {
#else
    if (vec) {
#endif
        float* const At = dw_smem;
        float* const Bt = dw_smem + TM * LDK;
        const int c = lane & 15, q = lane >> 4;
        // quads of a slab: A 32 rows x 12 (8 per row for every thread, the other 4 for threads 0..127), B 32 rows x 16 (8 + 8):
        // a wave reads 8 rows x 128 bytes (or 16 x 64) per load.  LDS position of (feature f, k): f * LDK + 4 * ((k >> 2) ^
        // ((f >> 2) & 7)) + (k & 3) -- the XOR spreads a wave's transposed stores over all banks (2 lanes per bank)
        const int kl = tid >> 3, ql = tid & 7, kl2 = (tid >> 2) & 31, ql2 = 8 + (tid & 3);
        const bool a1 = tid < 128;
        const int ma0 = mb + 4 * ql < Mq ? mb + 4 * ql : Mq - 4, ma1 = mb + 4 * ql2 < Mq ? mb + 4 * ql2 : Mq - 4;
        const int nb0 = nb + 4 * ql < Nq ? nb + 4 * ql : Nq - 4, nb1 = nb + 32 + 4 * ql < Nq ? nb + 32 + 4 * ql : Nq - 4;
        const floatx4* pA0 = reinterpret_cast<const floatx4*>(A + (long)kl * jb.ldx + ma0);
        const floatx4* pA1 = reinterpret_cast<const floatx4*>(A + (long)kl2 * jb.ldx + ma1);
        const floatx4* pB0 = reinterpret_cast<const floatx4*>(Bm + (long)kl * jb.ldz + nb0);
        const floatx4* pB1 = reinterpret_cast<const floatx4*>(Bm + (long)kl * jb.ldz + nb1);
        const long sA = (long)TK * jb.ldx / 4, sB = (long)TK * jb.ldz / 4;    // slab strides in float4
        float* const wA0 = At + (4 * ql) * LDK + 4 * ((kl >> 2) ^ (ql & 7)) + (kl & 3);
        float* const wA1 = At + (4 * ql2) * LDK + 4 * ((kl2 >> 2) ^ (ql2 & 7)) + (kl2 & 3);
        float* const wB0 = Bt + (4 * ql) * LDK + 4 * ((kl >> 2) ^ (ql & 7)) + (kl & 3);
        float* const wB1 = Bt + (32 + 4 * ql) * LDK + 4 * ((kl >> 2) ^ ((8 + ql) & 7)) + (kl & 3);
        const bool n_ok = nb + 16 * wn < N;                              // units past the matrix edge are skipped
        // two slabs of loads in flight (registers): slab s + 2 is requested when slab s has been stashed
        struct Slab { floatx4 a0, a1, b0, b1, c0, c1; } r[2];      // (c: the second gradient, added when the slab is stashed)
        auto fetch = [&](Slab& d) {
            d.a0 = *pA0; d.a1 = a1 ? *pA1 : floatx4{0.f, 0.f, 0.f, 0.f}; d.b0 = *pB0; d.b1 = *pB1;
            if (two) { d.c0 = pB0[d2 / 4]; d.c1 = pB1[d2 / 4]; }     // (used at the stash only: the branch costs no wait)
            pA0 += sA; pA1 += sA; pB0 += sB; pB1 += sB;
        };
        // One barrier per slab: slab s is stashed into buffer s & 1 while the slower waves may still be reading slab s - 1 out of the other
        // one; the stores of slab s + 1 (same buffer as s - 1) come behind the barrier of slab s, which every wave passes only after its
        // reads of slab s - 1.  (Single-buffered until round 5: two barriers per 24 MFMAs.)
        auto slab = [&](Slab& d, int k0, int par) {
            const int bo = par * DW_SLAB;
#if CADM_DW_EXPERIMENT == 3      // (timing experiments, tools/build_variant.sh: what a part of the slab loop costs -- results are wrong)
            if (k0 < 2 * TK)
#endif
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wA0[bo + j * LDK] = d.a0[j];
                if (a1) wA1[bo + j * LDK] = d.a1[j];
                wB0[bo + j * LDK] = two ? d.b0[j] + d.c0[j] : d.b0[j];
                wB1[bo + j * LDK] = two ? d.b1[j] + d.c1[j] : d.b1[j];
            }
            __syncthreads();
#if CADM_DW_EXPERIMENT != 2
            if (k0 + 2 * TK < KP) fetch(d);
#endif
            if (do_colsum) {
#pragma unroll
                for (int x = 0; x < TK / 4; ++x) {
                    const floatx4 v = *reinterpret_cast<const floatx4*>(Bt + bo + tid * LDK + 4 * (x ^ ((tid >> 2) & 7)));
                    colsum += v[0]; colsum += v[1]; colsum += v[2]; colsum += v[3];
                }
            }
            if (n_ok) {
                const int fb = 16 * wn + c;
#pragma unroll
                for (int g = 0; g < TK / 16; ++g) {
                    const floatx4 b4 = *reinterpret_cast<const floatx4*>(Bt + bo + fb * LDK + 4 * ((4 * g + q) ^ ((fb >> 2) & 7)));
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        if (mb + 16 * i >= M) continue;
                        const int fa = 16 * i + c;
                        const floatx4 a4 = *reinterpret_cast<const floatx4*>(At + bo + fa * LDK + 4 * ((4 * g + q) ^ ((fa >> 2) & 7)));
#if CADM_DW_EXPERIMENT == 1
                        acc[i] += a4 * b4;
#else
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u], b4[u], acc[i], 0, 0, 0);
#endif
                    }
                }
            }
        };
        fetch(r[0]);
        if (KP > TK) fetch(r[1]);
        for (int k0 = 0; k0 < KP; k0 += 2 * TK) {
            slab(r[0], k0, 0);
            if (k0 + TK < KP) slab(r[1], k0 + TK, 1);
        }
    }
    if (!vec && KP > 0) issue(0);
    for (int k0 = 0; !vec && k0 < KP; k0 += TK) {
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
#ifdef CADM_DW_TIMING
    if (tstamp) a.tbuf[5200 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    if ((N & 3) == 0) {
        __syncthreads();                                 // every wave is done reading the slab buffers
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) dw_smem[(i * 16 + (lane >> 4) * 4 + r) * LDC + wn * 16 + (lane & 15)] = acc[i][r];
        __syncthreads();
        // W, m, v of the thread's three column quads are requested TOGETHER (addresses clamped into the layer, never predicated:
        // with a branch around each quad hipcc waits for one quad's loads before it issues the next -- three dependent round trips)
        constexpr int NQD = TM * TN / 4 / 256;
        floatx4 w3[NQD], m3[NQD], v3[NQD];
#pragma unroll
        for (int it = 0; it < NQD; ++it) {
            const int idx = tid + it * 256;
            const int ml = idx / (TN / 4), n4 = (idx % (TN / 4)) * 4;
            const int mc = mb + ml < M ? mb + ml : M - 1, nc = nb + n4 < N ? nb + n4 : N - 4;
            const long o = ((long)e * M + mc) * N + nc;
            w3[it] = *reinterpret_cast<const floatx4*>(jb.W + o); m3[it] = *reinterpret_cast<const floatx4*>(jb.Mw + o);
            v3[it] = *reinterpret_cast<const floatx4*>(jb.Vw + o);
        }
#pragma unroll
        for (int it = 0; it < NQD; ++it) {
            const int idx = tid + it * 256;
            const int ml = idx / (TN / 4), n4 = (idx % (TN / 4)) * 4;
            const int m = mb + ml, n = nb + n4;
            if (m >= M || n >= N) continue;
            const long o = ((long)e * M + m) * N + n;
            const floatx4 g = *reinterpret_cast<const floatx4*>(dw_smem + ml * LDC + n4);
            floatx4 w = w3[it], mo = m3[it], vo = v3[it];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float wc = w[c], mc = mo[c], vc = vo[c];
                adam_update(wc, mc, vc, g[c] + jb.wdc * wc, a.lr_t, a.b1, a.b2, a.eps);
                w[c] = wc; mo[c] = mc; vo[c] = vc;
            }
            *reinterpret_cast<floatx4*>(jb.W + o) = w;
            *reinterpret_cast<floatx4*>(jb.Mw + o) = mo;
            *reinterpret_cast<floatx4*>(jb.Vw + o) = vo;
            // packed copies: forward operand (k = m, column n): the 4 columns are 4 lanes of one block; transposed operand
            // (k = n, column m - row0): the 4 columns are one lane's 4 k
            if (jb.pf.P) {
                float* q = jb.pf.P + (long)e * jb.pf.sP + pack_index(jb.pf, m, n);
#pragma unroll
                for (int c = 0; c < 4; ++c) q[4 * c] = w[c];
            }
            if (jb.pb.P) {
                const int np = m - jb.pb.row0;
                if (np >= 0 && np < jb.pb.ncols) *reinterpret_cast<floatx4*>(jb.pb.P + (long)e * jb.pb.sP + pack_index(jb.pb, n, np)) = w;
            }
        }
    } else {
        // (odd N: the heads, the context vector) -- the loads of all 12 elements first, clamped, for the same reason
        float ws_[MI][4], ms_[MI][4], vs_[MI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + i * 16 + (lane >> 4) * 4 + r, n = nb + wn * 16 + (lane & 15);
                const long o = ((long)e * M + (m < M ? m : M - 1)) * N + (n < N ? n : N - 1);
                ws_[i][r] = jb.W[o]; ms_[i][r] = jb.Mw[o]; vs_[i][r] = jb.Vw[o];
            }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + i * 16 + (lane >> 4) * 4 + r;
                const int n = nb + wn * 16 + (lane & 15);
                if (m >= M || n >= N) continue;
                const long o = ((long)e * M + m) * N + n;
                float w = ws_[i][r], mo = ms_[i][r], vo = vs_[i][r];
                adam_update(w, mo, vo, acc[i][r] + jb.wdc * w, a.lr_t, a.b1, a.b2, a.eps);
                jb.W[o] = w; jb.Mw[o] = mo; jb.Vw[o] = vo;
                if (jb.pf.P) jb.pf.P[(long)e * jb.pf.sP + pack_index(jb.pf, m, n)] = w;
                const int np = m - jb.pb.row0;
                if (jb.pb.P && np >= 0 && np < jb.pb.ncols) jb.pb.P[(long)e * jb.pb.sP + pack_index(jb.pb, n, np)] = w;
            }
    }
#ifdef CADM_DW_TIMING
    if (tstamp) { a.tbuf[1024 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); a.tbuf[4096 + blockIdx.x] = ji * 2 + (vec ? 1 : 0); }
#endif
    if (do_colsum && nb + tid < N) {
        const long o = (long)e * N + nb + tid;
        float w = jb.bW[o], mo = jb.bM[o], vo = jb.bV[o];
        adam_update(w, mo, vo, colsum, a.lr_t, a.b1, a.b2, a.eps);
        jb.bW[o] = w; jb.bM[o] = mo; jb.bV[o] = vo;
    }
}

// ---------------------------------------------------------------------------------------------
// losses (dynamics.py:269-314) and output-layer gradients
// ---------------------------------------------------------------------------------------------
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
    std::vector<float*> cp_dz_bk;     // the context encoder's dz as propagated from the backward model's chain (cp.dz: from the forward net's)
    float *dMu = nullptr, *dLv = nullptr, *dBmu = nullptr, *terms = nullptr, *red = nullptr;
    // Adam moments, same order as the registered layers: W then b
    std::vector<AdamSlot> a_ff, a_bk, a_cp;   // 2 per layer
    AdamSlot a_mx, a_mn;
    float* adam_buf = nullptr;
    // packed operand streams of the chain kernel, per registered layer (forward / transposed); one allocation
    std::vector<PackDst> pf_ff, pb_ff, pf_bk, pb_bk, pf_cp, pb_cp;
    std::vector<int> pt_ff, pt_bk, pt_cp, ptb_ff, ptb_bk, ptb_cp;     // tiles (even) of the forward / transposed stream
    float* pack_buf = nullptr;
    // chain programs: [fwd ff | fwd back | bwd ff | bwd back | bwd context]
    std::vector<ChainStage> prog_host;
    ChainStage* prog_dev = nullptr;
    int prog_first[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prog_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ChainLoad pre[8][4]; int npre[8] = {0, 0, 0, 0, 0, 0, 0, 0};         // the programs' input tiles (kernel arguments of the launch)
    ChainAsm asmp{};                                            // raw batch of the current call (forward launch)
    int loss_buf = 0, loss_lv0 = 0;                             // where the forward chains leave the head outputs in LDS
    int K0p = 0, cpinp = 0, Dp = 0, Cp = 0;                     // padded row strides of Xff / Xbk, Xcp, dMu / dLv / dBmu, the dctx buffers
    int chain_bufsz = 0;
};

void cadm_train_free(cadm_ctx* ctx) {
    if (!ctx->train) return;
    if (ctx->train->ws) (void)hipFree(ctx->train->ws);
    if (ctx->train->adam_buf) (void)hipFree(ctx->train->adam_buf);
    if (ctx->train->prog_dev) (void)hipFree(ctx->train->prog_dev);
    if (ctx->train->pack_buf) (void)hipFree(ctx->train->pack_buf);
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

int cadm_train_adam_slot(cadm_ctx* ctx, int net, int layer, int is_bias, float** m, float** v, size_t* n) {
    *m = *v = nullptr; *n = 0;
    TrainState* t = ctx->train;
    if (!t || !t->adam_buf) return CADM_ESTATE;
    const AdamSlot* s = nullptr;
    if (layer < 0) {
        if (net != CADM_NET_FF || layer < -2) return CADM_EINVAL;
        s = layer == -1 ? &t->a_mx : &t->a_mn;
    } else {
        const std::vector<AdamSlot>& v_ = net == CADM_NET_FF ? t->a_ff : net == CADM_NET_BACK ? t->a_bk : t->a_cp;
        const size_t i = 2 * (size_t)layer + (is_bias ? 1 : 0);
        if (i >= v_.size()) return CADM_EINVAL;
        s = &v_[i];
    }
    *m = s->m; *v = s->v; *n = s->n;
    return CADM_OK;
}

static inline int kblocks(int k) { return (k + 15) >> 4; }
static inline int even_tiles(int n) { return 2 * ((((n + 15) >> 4) + 1) / 2); }

// Streams of the chain kernel's stages (geometry only; filled by pack_streams and kept current by dw_adam_kernel):
//   forward    every layer: Bop(k, n) = W[k][n]
//   transposed hidden layers l >= 1 and the context net's layers l >= 1: Bop(k, n') = W[n'][k];
//              layer 0 of a dynamics net: only its context rows (row0 = P + A, C of them) carry a gradient;
//              the heads mu (| logvar) of a net share ONE stream, concatenated along k: dz = [dMu | dLv] [Wmu | Wlv]^T
static int alloc_packs(cadm_ctx* ctx) {
    TrainState* t = ctx->train;
    const int E = ctx->E, NH = ctx->NH, D = ctx->D, C = ctx->C, PA = ctx->P + ctx->A;
    const bool has_back = ctx->cfg.back_model != 0, has_cp = C > 0, det = ctx->cfg.deterministic != 0;
    size_t total = 0;
    auto region = [&](PackDst& d, int ntile) { d.nt = ntile; d.sP = (long)ntile * d.KB * CH_BLK_FLOATS; d.P = reinterpret_cast<float*>(total + 1); total += (size_t)E * d.sP; };
    auto net = [&](const std::vector<DenseRef>& L, bool dyn, bool with_lv, std::vector<PackDst>& pf, std::vector<PackDst>& pb,
                   std::vector<int>& pt, std::vector<int>& ptb) {
        const int n = (int)L.size();
        pf.assign(n, PackDst{}); pb.assign(n, PackDst{}); pt.assign(n, 0); ptb.assign(n, 0);
        for (int l = 0; l < n; ++l) {
            if (dyn && l == NH + 1 && !with_lv) continue;          // logvar head outside the data path
            PackDst& f = pf[l];
            f.KB = kblocks(L[l].din); f.kb0 = 0; f.row0 = 0; f.ncols = L[l].dout;
            pt[l] = even_tiles(L[l].dout);
            region(f, pt[l]);
            PackDst& b = pb[l];
            if (dyn && l == 0) {
                if (!has_cp) continue;
                b.KB = kblocks(L[0].dout); b.kb0 = 0; b.row0 = PA; b.ncols = C;
                ptb[0] = even_tiles(C);
                region(b, ptb[0]);
            } else if (dyn && l >= NH) {
                const int KBd = kblocks(D);
                b.KB = KBd * (with_lv ? 2 : 1); b.kb0 = (l - NH) * KBd; b.row0 = 0; b.ncols = L[l].din;
                ptb[l] = even_tiles(L[l].din);
                if (l == NH) region(b, ptb[l]);
                else { b.P = pb[NH].P; b.sP = pb[NH].sP; b.nt = pb[NH].nt; }
            } else if (l >= 1) {
                b.KB = kblocks(L[l].dout); b.kb0 = 0; b.row0 = 0; b.ncols = L[l].din;
                ptb[l] = even_tiles(L[l].din);
                region(b, ptb[l]);
            }
        }
    };
    net(ctx->ff, true, !det, t->pf_ff, t->pb_ff, t->pt_ff, t->ptb_ff);
    if (has_back) net(ctx->back, true, false, t->pf_bk, t->pb_bk, t->pt_bk, t->ptb_bk);
    if (has_cp) net(ctx->cp, false, false, t->pf_cp, t->pb_cp, t->pt_cp, t->ptb_cp);
    CADM_CHECK_HIP(hipMalloc(&t->pack_buf, total * sizeof(float)));
    for (auto* v : {&t->pf_ff, &t->pb_ff, &t->pf_bk, &t->pb_bk, &t->pf_cp, &t->pb_cp})
        for (auto& d : *v)
            if (d.P) d.P = t->pack_buf + (reinterpret_cast<size_t>(d.P) - 1);
    ctx->train_packs_stale = true;
    return CADM_OK;
}

// (Re)build every stream from the registered master weights: after cadm_set_weights / cadm_repack, i.e. whenever the
// caller may have written the weights; the training step itself updates them in dw_adam_kernel's epilogue.
static int pack_streams(cadm_ctx* ctx, hipStream_t s) {
    TrainState* t = ctx->train;
    auto one = [&](const DenseRef& L, const PackDst& d, int tr, int nk, int ntile) -> int {
        if (!d.P) return CADM_OK;
        PackJob j{};
        j.W = L.W; j.M = L.din; j.N = L.dout; j.d = d; j.tr = tr; j.nk = nk; j.ntile = ntile;
        const long n = (long)ctx->E * ntile * kblocks(nk) * 64;
        hipLaunchKernelGGL(train_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, j, ctx->E);
        CADM_CHECK_HIP(hipGetLastError());
        return CADM_OK;
    };
    auto net = [&](const std::vector<DenseRef>& L, std::vector<PackDst>& pf, std::vector<PackDst>& pb, std::vector<int>& pt,
                   std::vector<int>& ptb) -> int {
        int rc;
        for (size_t l = 0; l < pf.size(); ++l) {
            if ((rc = one(L[l], pf[l], 0, L[l].din, pt[l]))) return rc;
            if ((rc = one(L[l], pb[l], 1, L[l].dout, ptb[l]))) return rc;
        }
        return CADM_OK;
    };
    int rc;
    if ((rc = net(ctx->ff, t->pf_ff, t->pb_ff, t->pt_ff, t->ptb_ff))) return rc;
    if (ctx->cfg.back_model && (rc = net(ctx->back, t->pf_bk, t->pb_bk, t->pt_bk, t->ptb_bk))) return rc;
    if (ctx->C > 0 && (rc = net(ctx->cp, t->pf_cp, t->pb_cp, t->pt_cp, t->ptb_cp))) return rc;
    ctx->train_packs_stale = false;
    return CADM_OK;
}

static inline int r4(int n) { return (n + 3) & ~3; }

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
    // odd-width tensors that the weight-gradient launch reads get rows padded to a multiple of 4 floats (zero columns: the
    // buffer is cleared once, nothing writes them): dw_adam_kernel then takes its 16-byte loop on them too
    const int K0p = r4(K0), cpinp = r4(cpin > 0 ? cpin : 1), Dp = r4(D), Cp = r4((int)Cw);
    t->K0p = K0p; t->cpinp = cpinp; t->Dp = Dp; t->Cp = Cp;
    const size_t oXff = need(R * K0p), oXbk = need(R * K0p), oXcp = need(R * cpinp);
    const size_t odCtx = need(R * Cp), odCff = need(R * Cp), odCbk = need(R * Cp);
    std::vector<size_t> oz_ff(NH), oh_ff(NH), od_ff(NH), oz_bk(NH), oh_bk(NH), od_bk(NH), oz_cp(ncp), oh_cp(ncp), od_cp(ncp), od_cpb(ncp);
    for (int l = 0; l < NH; ++l) {
        oz_ff[l] = need(R * HID); oh_ff[l] = need(R * HID); od_ff[l] = need(R * HID);
        oz_bk[l] = need(R * HID); oh_bk[l] = need(R * HID); od_bk[l] = need(R * HID);
    }
    for (int l = 0; l < ncp; ++l) {
        const size_t w = ctx->cfg.cp_hidden[l];
        oz_cp[l] = need(R * w); oh_cp[l] = need(R * w); od_cp[l] = need(R * w); od_cpb[l] = need(R * w);
    }
    const size_t omu = need(R * D), olv = need(R * D), obmu = need(R * D), oblv = need(R * D);
    const size_t odMu = need(R * Dp), odLv = need(R * Dp), odBmu = need(R * Dp);
    const size_t oterms = need((size_t)ctx->E * 2 * ((B + CH_ROWS - 1) / CH_ROWS) * (4 + 2 * (size_t)D)), ored = need(4 + 2 * (size_t)D + 8);   // a slot per forward workgroup
    CADM_CHECK_HIP(hipMalloc(&t->ws, total * sizeof(float)));
    CADM_CHECK_HIP(hipMemset(t->ws, 0, total * sizeof(float)));
    t->ws_floats = total;
    float* w = t->ws;
    t->Xff = w + oXff; t->Xbk = w + oXbk; t->Xcp = w + oXcp; t->dCtx = w + odCtx; t->ff.dctx = w + odCff; t->bk.dctx = w + odCbk;
    for (NetBufs* nb : {&t->ff, &t->bk}) { nb->z.resize(NH); nb->h.resize(NH); nb->dz.resize(NH); }
    t->cp.z.resize(ncp); t->cp.h.resize(ncp); t->cp.dz.resize(ncp);
    for (int l = 0; l < NH; ++l) {
        t->ff.z[l] = w + oz_ff[l]; t->ff.h[l] = w + oh_ff[l]; t->ff.dz[l] = w + od_ff[l];
        t->bk.z[l] = w + oz_bk[l]; t->bk.h[l] = w + oh_bk[l]; t->bk.dz[l] = w + od_bk[l];
    }
    t->cp_dz_bk.resize(ncp);
    for (int l = 0; l < ncp; ++l) { t->cp.z[l] = w + oz_cp[l]; t->cp.h[l] = w + oh_cp[l]; t->cp.dz[l] = w + od_cp[l]; t->cp_dz_bk[l] = w + od_cpb[l]; }
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

enum { PROG_FWD_FF = 0, PROG_FWD_BK = 1, PROG_BWD_FF = 2, PROG_BWD_BK = 3,
       PROG_FWD_BK_NOCP = 4,                                  // large batches: the backward model's forward chain with the context columns read back (fwd_prog)
       PROG_BWD_FF_NOCP = 5, PROG_BWD_BK_NOCP = 6, PROG_BWD_CP = 7,   // ... and the backward chains cut in front of the context encoder + ONE pass down it (bwd_prog)
       NPROG = 8 };

ChainLoad input_tile(const float* g0, const float* g1, float* gsum, int ld_in, int ldg, int K, int dst, int dk0, int zero_to, int mode = 0) {
    ChainLoad d{};
    d.mode = mode;
    d.g0 = g0; d.g1 = g1; d.gsum = gsum; d.ld_in = ld_in; d.ldg = ldg; d.K = K; d.dst = dst; d.dk0 = dk0; d.zero_to = zero_to;
    return d;
}

ChainStage gemm_stage(int src, int dst, int dk0, int act_d, int act_o) {
    ChainStage g{};
    g.src = src; g.dst = dst; g.dk0 = dk0; g.act_d = act_d; g.act_o = act_o; g.zfill = dst >= 0 && dk0 == 0;
    return g;
}

bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// appends a column segment (its packed operand `d` with `ntile` tiles) to a GEMM stage
void add_seg(ChainStage& g, const PackDst& d, int ntile, const float* bias, const float* zprev, float* out0, float* out1, int N,
             int ldo, int ldz) {
    const int si = g.ntp == 0 ? 0 : 1;
    ChainSeg& sg = g.seg[si];
    sg.P = d.P; sg.sP = d.sP; sg.nt = d.nt; sg.bias = bias; sg.zprev = zprev; sg.out0 = out0; sg.out1 = out1; sg.N = N; sg.ldo = ldo; sg.ldz = ldz;
    sg.vec = (N & 3) == 0 && ((!out0 && !out1) || (ldo & 3) == 0) && (!zprev || (ldz & 3) == 0) && (g.dst < 0 || (g.dk0 & 3) == 0) &&
             aligned16(bias) && aligned16(zprev) && aligned16(out0) && aligned16(out1);
    g.KB = d.KB;
    g.ntp += ntile / 2;
    if (si == 0) g.tp1 = g.ntp;          // (a second segment starts here)
}

// Builds the five stage lists for the current pointers and uploads them if anything changed.
int sync_programs(cadm_ctx* ctx, hipStream_t s) {
    TrainState* t = ctx->train;
    const bool has_back = ctx->cfg.back_model != 0, has_cp = ctx->C > 0, det = ctx->cfg.deterministic != 0;
    const int NH = ctx->NH, HID = ctx->HID, D = ctx->D, K0 = ctx->K0, C = ctx->C, PA = ctx->P + ctx->A;
    const int ncp = has_cp ? ctx->cfg.n_cp_hidden : 0;
    const int cpin = (ctx->D + ctx->A) * ctx->cfg.history_length;
    std::vector<ChainStage> prog;
    int first[NPROG], count[NPROG];
    int maxk = 16;                       // k extent of the widest LDS activation buffer
    int cur_prog = 0;
    for (int i = 0; i < NPROG; ++i) t->npre[i] = 0;
    auto push = [&](const ChainStage& g) {
        const int ext = g.dst >= 0 ? g.dk0 + 32 * g.ntp : 0;       // whole tile pairs are written (zeros behind N)
        maxk = ext > maxk ? ext : maxk;
        prog.push_back(g);
    };
    auto input = [&](const ChainLoad& d) {
        maxk = d.zero_to > maxk ? d.zero_to : maxk;
        t->pre[cur_prog][t->npre[cur_prog]++] = d;
    };

    // ctx_from: the context vector is not computed by this chain but read back from the forward net's input echo (its context columns,
    // written by the forward net's launch in front of this one): large batches, launch_forward
    auto fwd_prog = [&](const std::vector<DenseRef>& net, const std::vector<PackDst>& pf, const std::vector<int>& pt, float* X,
                        NetBufs& nb, bool store_cp, bool want_lv, const float* ctx_from = nullptr) {
        int cur;
        if (has_cp && ctx_from) {      // (buffer 2, as the chain with the encoder in front: the same walk through the buffers, the same loss_buf)
            input(input_tile(nullptr, nullptr, X, 0, t->K0p, ctx->P, 2, 0, ctx->P, 1));
            input(input_tile(nullptr, nullptr, X + ctx->P, 0, t->K0p, ctx->A, 2, ctx->P, PA, 2));
            input(input_tile(ctx_from, nullptr, X + PA, t->K0p, t->K0p, C, 2, PA, 16 * kblocks(K0)));
            cur = 2;
        } else if (has_cp) {
            // both inputs at once: the context encoder's history in buffer 0, this net's (obs, act) columns in buffer 2,
            // where the encoder's last stage drops the context vector behind them
            // (assembled from the raw batch on the way in; the forward net's workgroups also leave the normalised copies that
            //  the weight-gradient launch reads as layer-0 inputs.  g0 of the raw tiles is set per call: forward_nets)
            const int ncpo = D * ctx->cfg.history_length;
            input(input_tile(nullptr, nullptr, store_cp ? t->Xcp : nullptr, 0, t->cpinp, ncpo, 0, 0, ncpo, 3));
            input(input_tile(nullptr, nullptr, store_cp ? t->Xcp + ncpo : nullptr, 0, t->cpinp, cpin - ncpo, 0, ncpo, 16 * kblocks(cpin), 4));
            input(input_tile(nullptr, nullptr, X, 0, t->K0p, ctx->P, 2, 0, ctx->P, 1));
            input(input_tile(nullptr, nullptr, X + ctx->P, 0, t->K0p, ctx->A, 2, ctx->P, 16 * kblocks(K0), 2));
            cur = 0;
            for (int l = 0; l <= ncp; ++l) {
                const DenseRef& L = ctx->cp[l];
                if (l < ncp) {
                    ChainStage g = gemm_stage(cur, cur ^ 1, 0, ACT_NONE, ACT_RELU);
                    add_seg(g, t->pf_cp[l], t->pt_cp[l], L.b, nullptr, store_cp ? t->cp.z[l] : nullptr, store_cp ? t->cp.h[l] : nullptr,
                            L.dout, L.dout, 0);
                    push(g);
                    cur ^= 1;
                } else {   // context vector -> the ctx columns of this net's input (LDS and global)
                    ChainStage g = gemm_stage(cur, 2, PA, ACT_NONE, ACT_NONE);
                    add_seg(g, t->pf_cp[l], t->pt_cp[l], L.b, nullptr, nullptr, X + PA, L.dout, t->K0p, 0);
                    push(g);
                }
            }
            cur = 2;
        } else {
            input(input_tile(nullptr, nullptr, X, 0, t->K0p, ctx->P, 0, 0, ctx->P, 1));
            input(input_tile(nullptr, nullptr, X + ctx->P, 0, t->K0p, ctx->A, 0, ctx->P, 16 * kblocks(K0), 2));
            cur = 0;
        }
        for (int l = 0; l < NH; ++l) {
            const int dst = cur == 0 ? 1 : 0;
            ChainStage g = gemm_stage(cur, dst, 0, ACT_NONE, dyn_act(ctx));
            add_seg(g, pf[l], pt[l], net[l].b, nullptr, nb.z[l], nb.h[l], HID, HID, 0);
            push(g);
            cur = dst;
        }
        {   // the heads side by side: mu on the first tile pairs, logvar behind them; also left in LDS for the loss phase
            const int hd = cur == 0 ? 1 : 0;
            t->loss_buf = hd; t->loss_lv0 = 16 * pt[NH];
            ChainStage g = gemm_stage(cur, hd, 0, ACT_NONE, ACT_NONE);
            g.zfill = 0;
            add_seg(g, pf[NH], pt[NH], net[NH].b, nullptr, nullptr, nb.mu, D, D, 0);
            if (want_lv) add_seg(g, pf[NH + 1], pt[NH + 1], net[NH + 1].b, nullptr, nullptr, nb.lv, D, D, 0);
            push(g);
        }
    };
    auto bwd_prog = [&](const std::vector<DenseRef>& net, const std::vector<PackDst>& pb, const std::vector<int>& ptb, NetBufs& nb,
                        const float* dMu, const float* dLv, std::vector<float*>& cp_dz, bool with_cp = true) {
        const int KBd = kblocks(D);
        // [dMu | dLv] side by side along k (the heads' transposed operands are concatenated the same way)
        input(input_tile(dMu, nullptr, nullptr, t->Dp, 0, D, 0, 0, 16 * KBd));
        if (dLv) input(input_tile(dLv, nullptr, nullptr, t->Dp, 0, D, 0, 16 * KBd, 32 * KBd));
        {   // d z_{NH-1} = (dMu W_mu^T (+ dLv W_lv^T)) * act'(z_{NH-1})
            ChainStage g = gemm_stage(0, 1, 0, dyn_act(ctx), ACT_NONE);
            add_seg(g, pb[NH], ptb[NH], nullptr, nb.z[NH - 1], nullptr, nb.dz[NH - 1], HID, HID, HID);
            push(g);
        }
        int cur = 1;
        for (int l = NH - 1; l >= 1; --l) {
            const int dst = (cur + 1) % 3;
            ChainStage g = gemm_stage(cur, dst, 0, dyn_act(ctx), ACT_NONE);
            add_seg(g, pb[l], ptb[l], nullptr, nb.z[l - 1], nullptr, nb.dz[l - 1], net[l].din, HID, HID);
            push(g);
            cur = dst;
        }
        if (has_cp) {   // only the context columns of the input carry a gradient ...
            int dst = (cur + 1) % 3;
            ChainStage g = gemm_stage(cur, dst, 0, ACT_NONE, ACT_NONE);
            add_seg(g, pb[0], ptb[0], nullptr, nullptr, nullptr, nb.dctx, C, t->Cp, 0);
            push(g);
            cur = dst;
            // ... and it goes on down the context encoder in the same chain: backpropagation is linear in the incoming
            // gradient, so each dynamics net carries ITS share (cp_dz) and the weight-gradient launch adds the two on load --
            // no third chain launch, no pass of the context gradient through global memory
            // (with_cp = false, large batches: the chain ends here, with its share of the context gradient in nb.dctx; cp_bwd_prog)
            for (int l = ncp; with_cp && l >= 1; --l) {
                const DenseRef& L = ctx->cp[l];
                dst = (cur + 1) % 3;
                ChainStage h = gemm_stage(cur, dst, 0, ACT_RELU, ACT_NONE);
                add_seg(h, t->pb_cp[l], t->ptb_cp[l], nullptr, t->cp.z[l - 1], nullptr, cp_dz[l - 1], L.din, L.din, L.din);
                push(h);
                cur = dst;
            }
        }
    };

    cur_prog = PROG_FWD_FF; first[PROG_FWD_FF] = (int)prog.size(); fwd_prog(ctx->ff, t->pf_ff, t->pt_ff, t->Xff, t->ff, true, !det); count[PROG_FWD_FF] = (int)prog.size() - first[PROG_FWD_FF];
    cur_prog = PROG_FWD_BK; first[PROG_FWD_BK] = (int)prog.size(); if (has_back) fwd_prog(ctx->back, t->pf_bk, t->pt_bk, t->Xbk, t->bk, false, false); count[PROG_FWD_BK] = (int)prog.size() - first[PROG_FWD_BK];
    cur_prog = PROG_BWD_FF; first[PROG_BWD_FF] = (int)prog.size(); bwd_prog(ctx->ff, t->pb_ff, t->ptb_ff, t->ff, t->dMu, det ? nullptr : t->dLv, t->cp.dz); count[PROG_BWD_FF] = (int)prog.size() - first[PROG_BWD_FF];
    cur_prog = PROG_BWD_BK; first[PROG_BWD_BK] = (int)prog.size(); if (has_back) bwd_prog(ctx->back, t->pb_bk, t->ptb_bk, t->bk, t->dBmu, nullptr, t->cp_dz_bk); count[PROG_BWD_BK] = (int)prog.size() - first[PROG_BWD_BK];
    // Large batches: ONE pass down the context encoder on the SUM of the two dynamics nets' context gradients (an input tile adds the two on
    // the way in) instead of one pass per net inside its backward chain: a third launch of short items, half the context-encoder work.
    // Linear in the incoming gradient, so the same gradient -- summed before the pass instead of behind it (in dw_adam_kernel's loads):
    // equal to fp32 roundoff, not bit for bit; the gradient bars against fp64 are asserted for both forms.
    auto cp_bwd_prog = [&]() {
        input(input_tile(t->ff.dctx, t->bk.dctx, nullptr, t->Cp, 0, C, 0, 0, 16 * kblocks(C)));
        int cur = 0;
        for (int l = ncp; l >= 1; --l) {
            const DenseRef& L = ctx->cp[l];
            const int dst = (cur + 1) % 3;
            ChainStage h = gemm_stage(cur, dst, 0, ACT_RELU, ACT_NONE);
            add_seg(h, t->pb_cp[l], t->ptb_cp[l], nullptr, t->cp.z[l - 1], nullptr, t->cp.dz[l - 1], L.din, L.din, L.din);
            push(h);
            cur = dst;
        }
    };
    const bool can_merge = has_back && has_cp && ncp >= 1;
    cur_prog = PROG_BWD_FF_NOCP; first[cur_prog] = (int)prog.size(); if (can_merge) bwd_prog(ctx->ff, t->pb_ff, t->ptb_ff, t->ff, t->dMu, det ? nullptr : t->dLv, t->cp.dz, false); count[cur_prog] = (int)prog.size() - first[cur_prog];
    cur_prog = PROG_BWD_BK_NOCP; first[cur_prog] = (int)prog.size(); if (can_merge) bwd_prog(ctx->back, t->pb_bk, t->ptb_bk, t->bk, t->dBmu, nullptr, t->cp_dz_bk, false); count[cur_prog] = (int)prog.size() - first[cur_prog];
    cur_prog = PROG_BWD_CP; first[cur_prog] = (int)prog.size(); if (can_merge) cp_bwd_prog(); count[cur_prog] = (int)prog.size() - first[cur_prog];
    cur_prog = PROG_FWD_BK_NOCP; first[PROG_FWD_BK_NOCP] = (int)prog.size();
    if (has_back && has_cp) fwd_prog(ctx->back, t->pf_bk, t->pt_bk, t->Xbk, t->bk, false, false, t->Xff + PA);
    count[PROG_FWD_BK_NOCP] = (int)prog.size() - first[PROG_FWD_BK_NOCP];
    for (int i = 0; i < NPROG; ++i) CADM_REQUIRE(count[i] <= CH_MAXSTAGE, "training chain too long (more than 20 stages): too many layers");
    static_assert(CH_MAXSTAGE < 31, "ChainStage::nxt keeps a stage index in 5 bits (31: none)");
    for (int i = 0; i < NPROG; ++i)      // where each wave slot goes behind a stage (chain_group looks it up in ONE LDS read instead of walking the table)
        for (int k = 0; k < count[i]; ++k)
            for (int w = 0; w < CH_WAVES_MAX; ++w) {
                unsigned char v = 31;
                for (int j = k + 1; j < count[i]; ++j)
                    if (w < prog[first[i] + j].ntp) { v = (unsigned char)(j | (w >= prog[first[i] + j].tp1 ? 0x80 : 0)); break; }
                prog[first[i] + k].nxt[w] = v;
            }
    for (ChainStage& g : prog) {          // packed copies of the fields a lookup needs (fewer registers in flight across an epilogue)
        CADM_REQUIRE(g.KB < 256 && g.src < 256 && g.tp1 < 256 && g.ntp < 128, "training chain: stage too wide for the packed descriptor");
        g.pkA = g.KB | g.src << 8 | g.tp1 << 16 | g.ntp << 24;
        for (ChainSeg& sg : g.seg) {
            CADM_REQUIRE(sg.nt < (1 << 23) && sg.N < 65536 && sg.ldz < 32768, "training chain: layer too wide for the packed descriptor");
            sg.pkB = (sg.vec ? 1 : 0) | sg.nt << 8;
            sg.pkC = sg.N | sg.ldz << 16;
        }
    }

    const bool same = t->prog_dev && prog.size() == t->prog_host.size() &&
                      memcmp(prog.data(), t->prog_host.data(), prog.size() * sizeof(ChainStage)) == 0;
    if (!same) {
        if (t->prog_dev && prog.size() > t->prog_host.size()) { (void)hipFree(t->prog_dev); t->prog_dev = nullptr; }
        if (!t->prog_dev) CADM_CHECK_HIP(hipMalloc(&t->prog_dev, prog.size() * sizeof(ChainStage)));
        CADM_CHECK_HIP(hipStreamSynchronize(s));   // nothing in flight may still read the old table
        CADM_CHECK_HIP(hipMemcpy(t->prog_dev, prog.data(), prog.size() * sizeof(ChainStage), hipMemcpyHostToDevice));
        t->prog_host = prog;
    }
    for (int i = 0; i < NPROG; ++i) { t->prog_first[i] = first[i]; t->prog_count[i] = count[i]; }
    t->chain_bufsz = CH_ROWS * ((maxk + 15) & ~15);
    {   // (an activation buffer doubles as the loss phase's scratch: 6 term arrays of 16 x D elements, or 512 chunk sums, + a flag)
        const int terms = 6 * CH_ROWS * ctx->D, need = ((terms > CH_THREADS_MAX ? terms : CH_THREADS_MAX) + 4 + 63) & ~63;
        if (t->chain_bufsz < need) t->chain_bufsz = need;
    }
    return CADM_OK;
}

struct ChainLossCfg { LossP lp; ReduceP rp; int final; };

int launch_chain(cadm_ctx* ctx, int B, int p0, int p1, hipStream_t s, const ChainLossCfg* loss = nullptr, int y_base = 0, int slot_ny = 0) {
    TrainState* t = ctx->train;
    ChainArgs a{};
    a.prog = t->prog_dev;
    a.first[0] = t->prog_first[p0]; a.count[0] = t->prog_count[p0];
    a.ny = 1;
    if (p1 >= 0 && t->prog_count[p1] > 0) { a.first[1] = t->prog_first[p1]; a.count[1] = t->prog_count[p1]; a.ny = 2; }
    a.npre[0] = t->npre[p0];
    for (int i = 0; i < t->npre[p0]; ++i) a.pre[0][i] = t->pre[p0][i];
    if (a.ny == 2) { a.npre[1] = t->npre[p1]; for (int i = 0; i < t->npre[p1]; ++i) a.pre[1][i] = t->pre[p1][i]; }
    a.asmp = t->asmp;
    a.B = B; a.bufsz = t->chain_bufsz;
    a.tbuf = ctx->tbuf ? ctx->tbuf + 256 * (p0 / 2) : nullptr;   // [fwd | bwd] x 256 stamps (tools/chain_timing.py)
    a.tfine = ctx->tbuf ? ctx->tbuf + 512 + 128 * (p0 / 2) : nullptr;
    if (p0 >= PROG_FWD_BK_NOCP) a.tbuf = a.tfine = nullptr;      // (the extra launches of the large-batch path are not clocked)
    a.E = ctx->E; a.ntiles = (B + CH_ROWS - 1) / CH_ROWS;
    a.y_base = y_base; a.slot_ny = slot_ny ? slot_ny : a.ny;
    a.G = ctx->E <= 8 ? 8 / ctx->E : 1;
    const int per = a.ntiles * a.ny;
    a.ips = (per + a.G - 1) / a.G;
    const int rounds = (ctx->E + 7) / 8;
    const size_t lds = CH_MAXSTAGE * sizeof(ChainStage) + 3 * (size_t)t->chain_bufsz * sizeof(float);
    if (loss) {     // closing loss phase: 6 term arrays of the workgroup's 16 x D elements + a flag, in an activation buffer the chain is done with
        a.loss_on = 1; a.loss_final = loss->final; a.loss_buf = t->loss_buf; a.loss_lv0 = t->loss_lv0; a.loss_slots = ctx->E * a.slot_ny * a.ntiles;
        a.lossp = loss->lp; a.lossr = loss->rp;
        const size_t terms = 6 * (size_t)CH_ROWS * ctx->D;          // (the final reduction reuses it for up to 512 chunk sums)
        CADM_REQUIRE(((terms > CH_THREADS_MAX ? terms : CH_THREADS_MAX) + 4) * sizeof(float) <= (size_t)t->chain_bufsz * sizeof(float),
                     "training chain: loss scratch does not fit an activation buffer");
    }
    CADM_REQUIRE(lds <= 160 * 1024, "training chain: layer too wide for the LDS-resident activation tile");
    CADM_REQUIRE((long long)B * (t->chain_bufsz / CH_ROWS) * 4 < (1LL << 32),
                 "training chain: batch of %d rows too large for 32-bit per-member offsets", B);
    // more work items than CUs: the throughput flavour (three 4-wave workgroups per CU), if three of them fit the LDS
    const long items = (long)ctx->E * per;
    const bool wide = ctx->train_force_nw ? ctx->train_force_nw == 4 : (items > ctx->n_cus && 3 * lds <= 160 * 1024);
    const void* fn = wide ? reinterpret_cast<const void*>(&chain_kernel<4>) : reinterpret_cast<const void*>(&chain_kernel<8>);
    size_t& attr = wide ? ctx->chain_attr_lds4 : ctx->chain_attr_lds;
    if (lds > attr) {      // per ctx = per device (a process-wide flag would leave a second GPU's attribute unset)
        CADM_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    // member -> XCD affinity while one round of it holds the launch; beyond that every XCD takes a contiguous eighth of the items
    const int affine_slots = (ctx->n_cus / 8) * (wide ? 3 : 1);        // workgroups an XCD holds at once
    a.spread = ctx->train_force_spread ? ctx->train_force_spread == 1 : (ctx->E * a.G < 8 && a.ips > affine_slots);
    a.per_xcd = (int)((items + 7) / 8);
    const unsigned grid = a.spread ? 8u * (unsigned)a.per_xcd : 8u * (unsigned)(a.ips * rounds);
    if (wide) hipLaunchKernelGGL(chain_kernel<4>, dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(chain_kernel<8>, dim3(grid), dim3(512), lds, s, a);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

// forward of the context / forward (/ backward) nets on one [E,B,.] batch into the workspace
// The large-batch training path (one forward launch per net; ONE pass of the summed context gradient down the encoder) changes the
// arithmetic in the last bits (a sum taken before a linear pass instead of behind it), so WHEN it is taken must be a property of the
// batch, not of the device the step happens to run on: the rule "work items >= 1.5 rounds of three workgroups per CU" is
// evaluated for the part this library is written for -- MI355X, 256 CUs -- whatever ctx->n_cus says: with 5 members + backward model the
// switch is at B = 1856 (1843.2 rounded up to the 16-row tile).  INTEGRATION.md, "numeric envelope"; tests/test_gpu_train.py pins it.
#define CADM_LARGE_BATCH_CUS 256
int forward_nets(cadm_ctx* ctx, const RowMap& map, const float* obs, const float* act, const float* obs_next, const float* cp_obs,
                 const float* cp_act, int B, bool has_back, hipStream_t s, const ChainLossCfg* loss = nullptr) {
    TrainState* t = ctx->train;
    const int D = ctx->D;
    int rc;
    if (!t->pack_buf && (rc = alloc_packs(ctx))) return rc;
    if (ctx->train_packs_stale && (rc = pack_streams(ctx, s))) return rc;
    if ((rc = sync_programs(ctx, s))) return rc;
    ChainAsm& ap = t->asmp;
    ap.map = map;
    ap.act = act; ap.cp_obs = cp_obs; ap.cp_act = cp_act;
    ap.obs_mean = ctx->st.obs_mean; ap.obs_std = ctx->st.obs_std; ap.act_mean = ctx->st.act_mean; ap.act_std = ctx->st.act_std;
    ap.cp_obs_mean = ctx->st.cp_obs_mean; ap.cp_obs_std = ctx->st.cp_obs_std;
    ap.cp_act_mean = ctx->st.cp_act_mean; ap.cp_act_std = ctx->st.cp_act_std;
    ap.D = D; ap.A = ctx->A; ap.P = ctx->P;
    ap.ncpo = D * ctx->cfg.history_length; ap.ncpa = ctx->A * ctx->cfg.history_length;
    ap.env = ctx->cfg.env_kind;
    for (int i = 0; i < t->npre[PROG_FWD_FF]; ++i) if (t->pre[PROG_FWD_FF][i].mode == 1) t->pre[PROG_FWD_FF][i].g0 = obs;
    for (int i = 0; i < t->npre[PROG_FWD_BK]; ++i) if (t->pre[PROG_FWD_BK][i].mode == 1) t->pre[PROG_FWD_BK][i].g0 = obs_next;
    for (int i = 0; i < t->npre[PROG_FWD_BK_NOCP]; ++i) if (t->pre[PROG_FWD_BK_NOCP][i].mode == 1) t->pre[PROG_FWD_BK_NOCP][i].g0 = obs_next;
    // Large batches: one launch per net, the backward model's behind the forward net's -- its chains then READ the context vector the
    // forward net's chains have left in their input echo instead of running the context encoder a second time on the same histories
    // (4 of a chain's 9 stages; in the joint launch -- the reference's batch: one partial round of the chip -- the two chains of a
    // row tile run side by side and the recomputation costs nothing).  Same arithmetic, same loss partials in the same slots.
    const long items = (long)ctx->E * 2 * ((B + CH_ROWS - 1) / CH_ROWS);
    const bool split = has_back && ctx->C > 0 && t->prog_count[PROG_FWD_BK_NOCP] > 0 &&
                       (ctx->train_force_spread ? ctx->train_force_spread == 1 : 2 * items >= 9L * CADM_LARGE_BATCH_CUS);      // (>= 1.5 rounds of three workgroups per CU: B = 2048 0.4485 -> 0.4445 ms, B = 1024 0.249 -> 0.280)
    if (!split) return launch_chain(ctx, B, PROG_FWD_FF, has_back ? PROG_FWD_BK : -1, s, loss);
    if ((rc = launch_chain(ctx, B, PROG_FWD_FF, -1, s, loss, 0, 2))) return rc;
    return launch_chain(ctx, B, PROG_FWD_BK_NOCP, -1, s, loss, 1, 2);
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
    const int E = ctx->E, NH = ctx->NH, HID = ctx->HID, D = ctx->D;
    const int ncp = has_cp ? ctx->cfg.n_cp_hidden : 0;
    const long R = (long)E * B;

    // ---- forward + losses + head gradients ----
    LossP lp{};
    lp.map = map;
    lp.mu = t->ff.mu; lp.lv = t->ff.lv; lp.bmu = t->bk.mu; lp.delta = delta; lp.back_delta = back_delta;
    lp.dmean = ctx->st.delta_mean; lp.dstd = ctx->st.delta_std; lp.bdmean = ctx->st.back_delta_mean; lp.bdstd = ctx->st.back_delta_std;
    lp.maxlv = ctx->ff_maxlv; lp.minlv = ctx->ff_minlv;
    lp.dMu = t->dMu; lp.dLv = t->dLv; lp.dBmu = t->dBmu;
    lp.n = R * D; lp.D = D; lp.Dp = t->Dp; lp.B = B; lp.det = det; lp.has_back = has_back; lp.back_coeff = hp.back_coeff;
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
    // forward chains; their closing phase takes the losses, the head gradients and the reductions (chain_loss_phase)
    ChainLossCfg lc{lp, rp, train ? 0 : 1};
    if ((rc = forward_nets(ctx, map, obs, act, obs_next, cp_obs, cp_act, B, has_back, s, &lc))) return rc;
    if (!train) return CADM_OK;

    // ---- backward + Adam ----
    const float coeff = hp.weight_decay_coeff;
    auto wd_dyn = [&](int l) { return coeff * (l < NH ? hp.weight_decays[l] : hp.weight_decays[NH]); };

    // backward chains (read W) ...
    const long bw_items = (long)E * 2 * ((B + CH_ROWS - 1) / CH_ROWS);
    const bool merge = has_back && has_cp && t->prog_count[PROG_BWD_CP] > 0 &&
                       (ctx->train_force_merge ? ctx->train_force_merge == 1 : 2 * bw_items >= 9L * CADM_LARGE_BATCH_CUS);      // (forward_nets' rule)
    if (!merge) {
        if ((rc = launch_chain(ctx, B, PROG_BWD_FF, has_back ? PROG_BWD_BK : -1, s))) return rc;
    } else {
        if ((rc = launch_chain(ctx, B, PROG_BWD_FF_NOCP, PROG_BWD_BK_NOCP, s))) return rc;
        if ((rc = launch_chain(ctx, B, PROG_BWD_CP, -1, s))) return rc;
    }

    // ... then every layer's weight gradient + Adam as one grouped launch (overwrites W)
    DwArgs da{};
    da.B = B; da.lr_t = lr_t; da.b1 = hp.beta1; da.b2 = hp.beta2; da.eps = hp.epsilon;
    int tiles = 0;
    auto add_job = [&](const float* X, int ldx, const float* dZ, int ldz, const DenseRef& L, float wdc, AdamSlot& aw, AdamSlot& ab,
                       const PackDst& pf, const PackDst& pb) -> int {
        CADM_REQUIRE(da.njobs < DW_MAXJOBS, "cadm_train_step: too many layers for the grouped weight-gradient launch");
        DwJob& j = da.job[da.njobs++];
        j.X = X; j.dZ = dZ; j.dZ2 = nullptr; j.W = L.W; j.Mw = aw.m; j.Vw = aw.v; j.bW = L.b; j.bM = ab.m; j.bV = ab.v;
        j.ldx = ldx; j.ldz = ldz; j.M = L.din; j.N = L.dout; j.tile0 = tiles; j.wdc = wdc; j.tn = (L.dout + TN - 1) / TN;
        j.pf = pf; j.pb = pb;
        tiles += j.tn * ((L.din + TM - 1) / TM);
        return CADM_OK;
    };
    auto net_jobs = [&](std::vector<DenseRef>& net, const float* X, NetBufs& nb, std::vector<AdamSlot>& ad, const float* dMu,
                        const float* dLv, std::vector<PackDst>& pf, std::vector<PackDst>& pb) -> int {
        int r;
        for (int l = 0; l < NH; ++l)
            if ((r = add_job(l == 0 ? X : nb.h[l - 1], l == 0 ? t->K0p : HID, nb.dz[l], HID, net[l], wd_dyn(l), ad[2 * l], ad[2 * l + 1], pf[l], pb[l]))) return r;
        if ((r = add_job(nb.h[NH - 1], HID, dMu, t->Dp, net[NH], wd_dyn(NH), ad[2 * NH], ad[2 * NH + 1], pf[NH], pb[NH]))) return r;
        if (dLv && (r = add_job(nb.h[NH - 1], HID, dLv, t->Dp, net[NH + 1], wd_dyn(NH + 1), ad[2 * (NH + 1)], ad[2 * (NH + 1) + 1], pf[NH + 1], pb[NH + 1]))) return r;
        return CADM_OK;
    };
    if ((rc = net_jobs(ctx->ff, t->Xff, t->ff, t->a_ff, t->dMu, det ? nullptr : t->dLv, t->pf_ff, t->pb_ff))) return rc;
    if (has_back && (rc = net_jobs(ctx->back, t->Xbk, t->bk, t->a_bk, t->dBmu, nullptr, t->pf_bk, t->pb_bk))) return rc;
    if (has_cp) {
        auto wd_cp = [&](int l) { return coeff * (l < ncp ? hp.context_weight_decays[l] : hp.context_weight_decays[ncp]); };
        for (int l = 0; l <= ncp; ++l) {      // gradient = the forward net's share (+ the backward model's), added on load
            if ((rc = add_job(l == 0 ? t->Xcp : t->cp.h[l - 1], l == 0 ? t->cpinp : ctx->cp[l - 1].dout, l == ncp ? t->ff.dctx : t->cp.dz[l],
                              l == ncp ? t->Cp : ctx->cp[l].dout, ctx->cp[l], wd_cp(l), t->a_cp[2 * l], t->a_cp[2 * l + 1], t->pf_cp[l], t->pb_cp[l]))) return rc;
            if (has_back && !(merge && l < ncp)) da.job[da.njobs - 1].dZ2 = l == ncp ? t->bk.dctx : t->cp_dz_bk[l];      // (merged: cp.dz holds the sum)
        }
    }
    // output_logvar outside the data path (deterministic forward net / backward net): its weight only sees the L2 term
    // (a job without data: X = null), its bias has no gradient at all and is skipped like TF does
    auto l2_only_job = [&](const DenseRef& L, float wdc, AdamSlot& aw) -> int {
        CADM_REQUIRE(da.njobs < DW_MAXJOBS, "cadm_train_step: too many layers for the grouped weight-gradient launch");
        DwJob& j = da.job[da.njobs++];
        j.X = nullptr; j.dZ = nullptr; j.dZ2 = nullptr; j.W = L.W; j.Mw = aw.m; j.Vw = aw.v; j.bW = nullptr; j.bM = nullptr; j.bV = nullptr;
        j.pf = PackDst{}; j.pb = PackDst{};
        j.ldx = 0; j.ldz = 0; j.M = L.din; j.N = L.dout; j.tile0 = tiles; j.wdc = wdc; j.tn = (L.dout + TN - 1) / TN;
        tiles += j.tn * ((L.din + TM - 1) / TM);
        return CADM_OK;
    };
    if (det && (rc = l2_only_job(ctx->ff[NH + 1], wd_dyn(NH + 1), t->a_ff[2 * (NH + 1)]))) return rc;
    if (has_back && (rc = l2_only_job(ctx->back[NH + 1], wd_dyn(NH + 1), t->a_bk[2 * (NH + 1)]))) return rc;
    da.tiles = tiles; da.E = E;
    da.tbuf = ctx->tbuf;
    for (int i = 0; i < da.njobs; ++i) da.tile0s[i] = da.job[i].tile0;
    da.lossr = rp; da.loss_slots = E * (has_back ? 2 : 1) * ((B + CH_ROWS - 1) / CH_ROWS);
    hipLaunchKernelGGL(dw_adam_kernel, dim3(8 * ((tiles * E + 7) / 8) + 8), dim3(256), 0, s, da);
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
