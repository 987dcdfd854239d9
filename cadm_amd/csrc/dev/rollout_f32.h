#pragma once
// DEVELOPER COMPARISON KERNEL (round 1's fp32-MFMA rollout): linked only into libcadm_hip_dev.so, selected only through
// cadm_dev_set_rollout (dev/dev_api.hip).  The product library has ONE rollout path, rollout_xdl.h.
//
// Fused trajectory-sampling rollout: ONE launch advances every (candidate, particle) row through
// the whole horizon -- input assembly, the 6-matmul ensemble MLP, Gaussian head, state update and
// reward accumulation (reference core/utils.py:431-472; SURVEY.md groups G3..G6).
//
// Mapping (DESIGN.md "rollout kernel"):
//   * workgroup = 4 waves (one per SIMD) = MT tiles of 16 rows of ONE ensemble member;
//   * every dense layer is evaluated transposed, OUT^T = W^T * IN^T, with v_mfma_f32_16x16x4_f32:
//     weights are the A operand (streamed from L2 in pre-packed fragment order through a 3-deep
//     register ring of buffer loads with scalar offsets), the 16 data rows are the B/D columns.
//     The D-layout of a layer's output IS the B-layout of the next layer's input, so activations
//     cross layers through LDS with lane-linear ds_write_b128 / ds_read_b128 and no transposes;
//   * the 13 output tiles of a 200-wide layer are split over the 4 waves as 3 full tiles each plus
//     one tile that is K-split (wave w takes k-step w of every chunk); partial sums meet in LDS;
//   * the rollout state lives in REGISTERS of the 256 "feature threads" (thread (row, fg) owns input
//     features fg, fg+16, .. of its row and tracks the observation dims they read); after the head
//     pass every thread redoes the tiny Gaussian-head math for its own dims from the head tiles in
//     LDS, so state update + next-step input assembly are one phase with no extra barrier.
#include <type_traits>
#include <utility>

#include "../common.h"
#include "../rollout_args.h"
#include "../rollout_env.h"


namespace {

template <int ENV_, int C_, int HID_, int MT_>
struct RC {
    static constexpr int ENV = ENV_, C = C_, HID = HID_, MT = MT_;
    static constexpr int D = env_D(ENV), A = env_A(ENV), P = env_P(ENV);
    static constexpr int K0 = P + A + C;
    static constexpr int NC0 = (K0 + 15) / 16;                         // chunks of layer 0
    static constexpr int KSL0 = ((K0 - 16 * (NC0 - 1)) + 3) / 4;       // k-steps in its last chunk
    static constexpr int NT = (HID + 15) / 16;                         // hidden tiles == chunks of K=HID layers
    static constexpr int KSLH = ((HID - 16 * (NT - 1)) + 3) / 4;
    static constexpr int NF = NT / 4, NS = NT % 4;                     // full tiles per wave / split tiles
    static constexpr int NTO = (D + 7) / 8;                            // head tiles (8 dims: mu|lv)
    static constexpr int NFO = NTO / 4, NSO = NTO % 4;
#ifndef CADM_PF
#define CADM_PF 3
#endif
    static constexpr int PF = CADM_PF;                                 // weight ring depth (chunks)
    // head layer with only K-split tiles: split K by CHUNK instead of by k-step (see head_pass_csplit);
    // must match make_geo() in capi.hip
    static constexpr bool OCS = NFO == 0 && NSO > 0 && NS == 1;
    static constexpr int NSL = (NT + 3) / 4;                           // chunk slots per wave in that mode
    static constexpr int OX_NCH = OCS ? NSL : NT, OX_NFO = OCS ? NSO : NFO, OX_NSO = OCS ? 0 : NSO;
    static constexpr int NFM = cmax(cmax(NF, OX_NFO), 1), NSM = cmax(cmax(NS, OX_NSO), 1);
    static constexpr int NP = (D + 1) / 2;                             // obs dim pairs
    static constexpr int NPI = (NP + 15) / 16;                         // pair tasks per thread
    static constexpr int NAI = (A + 15) / 16;                          // action-feature tasks per thread
    // LDS carve (floats)
    static constexpr int X_IN = 0;
    static constexpr int ACT_A = X_IN + MT * NC0 * 256;
    static constexpr int ACT_B = ACT_A + MT * (NT - NS) * 256;
    static constexpr int PART_A = ACT_B + MT * (NT - NS) * 256;
    static constexpr int PART_B = PART_A + MT * cmax(NS, 1) * 4 * 256;
    static constexpr int OPART = PART_B + MT * cmax(NS, 1) * 4 * 256;  // K-split head tiles: 4 partials each
    static constexpr int OFULL = OPART + MT * cmax(NSO, 1) * 4 * 256;  // full head tiles (+bias)
    static constexpr int STATS = OFULL + MT * cmax(4 * NFO, 1) * 256;
    static constexpr int ST_OBS_MEAN = STATS, ST_OBS_DEN = ST_OBS_MEAN + P, ST_ACT_MEAN = ST_OBS_DEN + P,
                         ST_ACT_DEN = ST_ACT_MEAN + A;
    static constexpr int CTRL_S = rup(ST_ACT_DEN + A, 4);              // + MT*16*H floats (dynamic)
};


template <class G>
struct Ring {
    floatx4 f[G::PF][G::NFM];
    float s[G::PF][G::NSM];
};

// one chunk of this wave's stream: NFO float4 blocks + NSO dword blocks, lane-linear
template <int SLOT, int NFO, int NSO, class G>
__device__ __forceinline__ void ring_load(Ring<G>& ring, __amdgpu_buffer_rsrc_t rsrc, unsigned soff, int lane) {
#ifdef CADM_ABLATE_SAME_LINES
    soff = 0;      // developer ablation: every ring load re-reads the same few KB (L1 hits): issue cost without memory time
#endif
#pragma unroll
    for (int i = 0; i < NFO; ++i)
        ring.f[SLOT][i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + i * 1024, soff, 0));
#pragma unroll
    for (int s = 0; s < NSO; ++s)
        ring.s[SLOT][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4 + NFO * 1024 + s * 256, soff, 0));
}

// One dense layer's MFMA sweep for this wave.
//   ring holds chunks 0..PF-1 of this layer on entry and chunks 0..PF-1 of the NEXT layer on exit.
//   B operands: chunks [0, NCH-IN_NS) come from lds_in (already activated), the last IN_NS chunks
//   are the producer's K-split tiles, rebuilt into bsplit / bsplit_sel by the `side` hook (they are
//   only needed at the END of the sweep, so the rebuild is interleaved with an early chunk: a co-resident workgroup's
//   MFMAs fill the matrix pipe meanwhile -- a wave's own MFMAs do not, see profiles/r1_mfma_issue_microbench.md).
//   The K-split output tiles of THIS layer use k-step `wave` of every chunk: its B operand is one
//   float per lane, read straight from LDS at a wave-dependent address (no selects, no branches).
//   side(jc) is extra independent work scheduled inside chunk jc's MFMA region.
template <class G, int NCH, int KSL, int NFO, int NSO, int NX_NCH, int NX_NFO, int NX_NSO, int IN_NS, unsigned SIDE_MASK, class Side>
__device__ __forceinline__ void mfma_pass(Ring<G>& ring, __amdgpu_buffer_rsrc_t rsrc, unsigned wcur, unsigned wnext,
                                          const float* lds_in, const floatx4 (&bsplit)[G::MT][cmax(IN_NS, 1)],
                                          const float (&bsplit_sel)[G::MT][cmax(IN_NS, 1)],
                                          floatx4 (&accF)[G::MT][cmax(NFO, 1)],
                                          floatx4 (&accS)[G::MT][cmax(NSO, 1)], int wave, int lane, Side&& side) {
    constexpr int MT = G::MT, PF = G::PF;
    constexpr int NCHPAD = rup(NCH, PF);
    constexpr int NLDS = NCH - IN_NS;                       // chunks resident in lds_in
    constexpr unsigned SLOTB = (NFO * 4 + NSO) * 256;       // bytes per chunk of this wave's stream
    constexpr unsigned NX_SLOTB = (NX_NFO * 4 + NX_NSO) * 256;
    const float* lds_lane = lds_in + lane * 4;
    const float* lds_sel = lds_in + lane * 4 + wave;
    // B operands are double-buffered one chunk ahead; sched_barrier pins the software pipeline
    // (ds_read of chunk j+1 | MFMAs of chunk j + side work | ring loads of chunk j+PF).
    floatx4 bq[2][MT];
    float bs[2][MT];
    auto load_b = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if constexpr (j < NLDS) {
                if constexpr (NFO > 0) bq[j & 1][mt] = *reinterpret_cast<const floatx4*>(lds_lane + (mt * NLDS + j) * 256);
                if constexpr (NSO > 0) bs[j & 1][mt] = lds_sel[(mt * NLDS + j) * 256];
            } else {
                bq[j & 1][mt] = bsplit[mt][j - NLDS];
                bs[j & 1][mt] = bsplit_sel[mt][j - NLDS];
            }
        }
    };
    load_b(std::integral_constant<int, 0>{});
    static_for(std::make_integer_sequence<int, NCHPAD>{}, [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int slot = j % PF;
        if constexpr (j < NCH) {
            constexpr int nk = (j == NCH - 1) ? KSL : 4;
            if constexpr (j + 1 < NCH) load_b(std::integral_constant<int, j + 1>{});
            side(jc);
#pragma unroll
            for (int r = 0; r < nk; ++r)
#pragma unroll
                for (int i = 0; i < NFO; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        accF[mt][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.f[slot][i][r], bq[j & 1][mt][r],
                                                                           accF[mt][i], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NSO; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    accS[mt][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.s[slot][s], bs[j & 1][mt],
                                                                       accS[mt][s], 0, 0, 0);
        }
        constexpr int jj = j + PF;
        if constexpr (jj < NCH) {
            ring_load<slot, NFO, NSO>(ring, rsrc, wcur + jj * SLOTB, lane);
        } else if constexpr (jj >= NCHPAD && (jj - NCHPAD) < NX_NCH) {
            ring_load<slot, NX_NFO, NX_NSO>(ring, rsrc, wnext + (jj - NCHPAD) * NX_SLOTB, lane);
        }
        if constexpr (j < NCH) {
            // issue order inside the chunk: every MFMA gap carries at most one memory instruction (the ring loads
            // of chunk j+PF, then the LDS reads of chunk j+1) plus, on side-work chunks, a few VALU -- so the
            // matrix pipe never waits behind a cluster of other instructions
            constexpr int nk2 = (j == NCH - 1) ? KSL : 4;
            constexpr int NM = (nk2 * NFO + NSO) * MT;
            constexpr bool has_side = (SIDE_MASK >> j) & 1u;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // 1 MFMA
                if (i < NFO + NSO + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                else if (i < NFO + NSO + 2 + 2 * MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                if (has_side) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);        // up to 4 VALU
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Head layer whose tiles are all K-split, split by CHUNK: wave w owns chunks w, w+4, w+8 of the
// activated input in LDS plus (wave 0 only; the others stream zero weights) the last chunk, which is
// the producer's K-split tile held in registers.  Weights arrive as float4 blocks per (slot, tile):
// 4x fewer load instructions than the k-step split and the whole pass is prefetchable 3 slots ahead.
template <class G, int NX_NCH, int NX_NFO, int NX_NSO, unsigned SIDE_MASK, class Side>
__device__ __forceinline__ void head_pass_csplit(Ring<G>& ring, __amdgpu_buffer_rsrc_t rsrc, unsigned wcur,
                                                 unsigned wnext, const float* lds_in,
                                                 const floatx4 (&bsplit)[G::MT][cmax(G::NS, 1)],
                                                 floatx4 (&accS)[G::MT][cmax(G::NSO, 1)], int wave, int lane,
                                                 Side&& side) {
    constexpr int MT = G::MT, PF = G::PF, NSO = G::NSO, NSL = G::NSL, NLDS = G::NT - G::NS;
    constexpr int NSLPAD = rup(NSL, PF);
    constexpr unsigned SLOTB = NSO * 1024;
    constexpr unsigned NX_SLOTB = (NX_NFO * 4 + NX_NSO) * 256;
    static_assert(G::NS == 1 && NLDS == 4 * (NSL - 1), "csplit head pass expects exactly one trailing K-split input chunk");
    floatx4 b[NSL - 1][MT];
#pragma unroll
    for (int j = 0; j < NSL - 1; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            b[j][mt] = *reinterpret_cast<const floatx4*>(lds_in + ((mt * NLDS + wave + 4 * j) * 64 + lane) * 4);
    __builtin_amdgcn_sched_barrier(0);
    static_for(std::make_integer_sequence<int, NSLPAD>{}, [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int slot = j % PF;
        if constexpr (j < NSL) {
            constexpr int nk = (j == NSL - 1) ? G::KSLH : 4;
            side(jc);
#pragma unroll
            for (int r = 0; r < nk; ++r)
#pragma unroll
                for (int s = 0; s < NSO; ++s)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float bv = j < NSL - 1 ? b[j < NSL - 1 ? j : 0][mt][r] : bsplit[mt][0][r];
                        accS[mt][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.f[slot][s][r], bv, accS[mt][s], 0, 0, 0);
                    }
            if constexpr ((SIDE_MASK >> j) & 1u) {
#pragma unroll
                for (int i = 0; i < nk * NSO * MT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
            }
        }
        constexpr int jj = j + PF;
        if constexpr (jj < NSL) {
            ring_load<slot, NSO, 0>(ring, rsrc, wcur + jj * SLOTB, lane);
        } else if constexpr (jj >= NSLPAD && (jj - NSLPAD) < NX_NCH) {
            ring_load<slot, NX_NFO, NX_NSO>(ring, rsrc, wnext + (jj - NSLPAD) * NX_SLOTB, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <int N>
__device__ __forceinline__ void zero_acc(floatx4 (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = floatx4{0.f, 0.f, 0.f, 0.f};
}

template <class G, int NOISE>
__global__ __launch_bounds__(256) void rollout_kernel(const RolloutArgs a) {
    constexpr int MT = G::MT, D = G::D, A = G::A, P = G::P, C = G::C, K0 = G::K0, NC0 = G::NC0;
    constexpr int NT = G::NT, NF = G::NF, NS = G::NS, NFO = G::NFO, NSO = G::NSO;
    constexpr int NP = G::NP, NPI = G::NPI, NAI = G::NAI;
    constexpr int ENV = G::ENV;
    constexpr int NFT = NT - NS;                                       // full hidden tiles
    static_assert(MT == 1, "MT > 1 needs per-tile return accumulators");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float wsel[4];                                                      // one-hot(wave) as per-lane floats: branch-free selects
#pragma unroll
    for (int r = 0; r < 4; ++r) wsel[r] = (tid >> 6) == r ? 1.0f : 0.0f;
    const int e = blockIdx.x / a.wgs_per_member;
    const int grp = blockIdx.x % a.wgs_per_member;
    const int H = a.H;

    float* x_in = smem + G::X_IN;
    float* ctrl_s = smem + G::CTRL_S;
    float* opart = smem + G::OPART;
    float* ofull = smem + G::OFULL;

    // ---- this thread's role: row arow = tid & 15, slot fg = tid >> 4:
    //      dim pairs dp = fg + 16*pi (state, head math, noise, reward part, derived input features),
    //      action features a = ((fg - NP) mod 16) + 16*ai, static features fg + 16k >= P + A (prologue only)
    const int arow = tid & 15, fg = tid >> 4;
    int lr[MT];          // local row index (returns / eps / traj)
    unsigned grow[MT];   // global row id (RNG counter)
    int abase[MT];       // offset of actions[mi, ni_global, 0, 0]
    bool valid[MT];
    int ctx_off[MT], mi_[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int re = (grp * MT + mt) * 16 + arow;
        valid[mt] = re < a.rows_per_member;
        if (!valid[mt]) re = a.rows_per_member - 1;
        const int cidx = re / a.PE, jl = re % a.PE;
        const int mi = cidx / a.n_local, nl = cidx % a.n_local;
        const int j = e * a.PE + jl;
        mi_[mt] = mi;
        lr[mt] = (mi * a.n_local + nl) * a.p + j;
        grow[mt] = (unsigned)((mi * a.n_global + a.cand_offset + nl) * a.p + j);
        abase[mt] = ((mi * a.n_global + a.cand_offset + nl) * H) * A;
        const int ep = j % a.E;
        if (!a.quirks) ctx_off[mt] = ((j / a.PE) * a.m + mi) * C;        // own member's context
        else if (a.it & 1) ctx_off[mt] = (mi * a.E + ep) * C;            // Q2: [E,m] memory reread as [m,E]
        else ctx_off[mt] = (ep * a.m + mi) * C;                          // Q1: encoder j % E
    }

    // ---- prologue: stats, start state, static input features, control costs ----
    for (int i = tid; i < P; i += 256) {
        smem[G::ST_OBS_MEAN + i] = a.obs_mean[i];
        smem[G::ST_OBS_DEN + i] = 1.0f / (a.obs_std[i] + 1e-10f);   // reciprocal: one multiply per feature per step
    }
    for (int i = tid; i < A; i += 256) {
        smem[G::ST_ACT_MEAN + i] = a.act_mean[i];
        smem[G::ST_ACT_DEN + i] = 1.0f / (a.act_std[i] + 1e-10f);
    }
    float po[MT][NPI][2];                 // tracked observation dims (2dp, 2dp+1)
    float pz[MT][NPI][2];                 // Gaussian-head noise of the step in flight
    float st_dmean[NPI][2], st_dden[NPI][2], st_dl2s[NPI][2], st_mx[NPI][2], st_mn[NPI][2];
    float areg[MT][NAI];                  // raw action of the NEXT step for this thread's action features
#pragma unroll
    for (int pi = 0; pi < NPI; ++pi) {
        const int dp = fg + 16 * pi;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = 2 * dp + h;
            const bool ok = d < D;
            const int dc = ok ? d : 0;
            st_dmean[pi][h] = a.delta_mean[dc];
            st_dden[pi][h] = a.delta_std[dc] + 1e-10f;
            st_dl2s[pi][h] = 2.0f * logf(a.delta_std[dc]);              // core/utils.py:360
            st_mx[pi][h] = a.maxlv[dc];
            st_mn[pi][h] = a.minlv[dc];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                po[mt][pi][h] = a.obs_rows ? a.obs_rows[(size_t)lr[mt] * D + dc] : a.obs[mi_[mt] * D + dc];   // :432
                pz[mt][pi][h] = 0.0f;
            }
        }
    }
    // input features derived from this thread's dims: LDS slot offset (-1: none), op, mean, 1/std
    int fx_off[NPI][2][2], fx_op[NPI][2][2];
    float fx_mean[NPI][2][2], fx_inv[NPI][2][2];
#pragma unroll
    for (int pi = 0; pi < NPI; ++pi)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = 2 * (fg + 16 * pi) + h;
            int ff[2], fop[2];
            const int nf = d < D ? dim_feats<ENV>(d, ff, fop) : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool on = i < nf;
                const int f = on ? ff[i] : 0;
                fx_off[pi][h][i] = on ? (((f >> 4) * 64 + (f & 3) * 16 + arow) * 4 + ((f & 15) >> 2)) : -1;
                fx_op[pi][h][i] = on ? fop[i] : 0;
                fx_mean[pi][h][i] = a.obs_mean[f];
                fx_inv[pi][h][i] = 1.0f / (a.obs_std[f] + 1e-10f);
            }
        }
    const int a0 = (fg - (NP & 15) + 16) & 15;                          // first action feature of this thread
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai) {
            const int ac = a0 + 16 * ai;
            areg[mt][ai] = ac < A ? a.actions[abase[mt] + ac] : 0.0f;
        }
        for (int f = fg; f < NC0 * 16; f += 16) {
            if (f >= P + A) {                                           // static: context (:433-439) / zero pad
                float v = 0.0f;
                if constexpr (C > 0) { if (f < K0) v = a.ctx_vec[ctx_off[mt] + f - P - A]; }
                x_in[((mt * NC0 + (f >> 4)) * 64 + (f & 3) * 16 + arow) * 4 + ((f & 15) >> 2)] = v;
            }
        }
        for (int t = fg; t < H; t += 16)
            ctrl_s[(mt * 16 + arow) * H + t] = ctrl_term<ENV>(a.actions + abase[mt] + t * A, A);
    }
    auto put_x = [&](int mt, int f, float v) {      // input slot of feature f: chunk f/16, k-step (f%16)/4, k-slot f%4
        x_in[((mt * NC0 + (f >> 4)) * 64 + (f & 3) * 16 + arow) * 4 + ((f & 15) >> 2)] = v;
    };

    // ---- weight stream: one buffer descriptor, scalar byte offsets ----
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wstream, 0, a.wbytes, 0x00020000);
    const unsigned wmem_b = e * a.wmember_b;
    const unsigned w0 = wmem_b + wave * (a.w_l0_b / 4);
    const unsigned wo = wmem_b + a.w_l0_b + (a.NH - 1) * a.w_lh_b + wave * (a.w_lo_b / 4);
    const float* bmem = a.bstream + (size_t)e * a.bmember;
    const float* bo = bmem + a.b_l0 + (size_t)(a.NH - 1) * a.b_lh;

    Ring<G> ring;
    static_for(std::make_integer_sequence<int, G::PF>{}, [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < NC0) ring_load<s, NF, NS>(ring, rsrc, w0 + s * ((NF * 4 + NS) * 256), lane);
    });

    float ret = 0.0f;
    __syncthreads();
    TS_DECL

    for (int t = 0; t <= H; ++t) {
        // ===== state update from step t-1's head (:348-365,463-466) + reward (:469-471) + input assembly (:442-460) =====
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int pi = 0; pi < NPI; ++pi) {
                const int dp = fg + 16 * pi;
                if (dp < NP) {
                    if (t > 0) {
                        const int jt = dp >> 2, lt = (dp & 3) * 16 + arow;
                        floatx4 v;                                       // (mu0, mu1, lv0, lv1) of this pair
                        if (NFO > 0 && jt < 4 * NFO) {
                            v = *reinterpret_cast<const floatx4*>(ofull + ((mt * 4 * NFO + jt) * 64 + lt) * 4);
                        } else {
                            const float* b = opart + (((mt * NSO + (jt - 4 * NFO)) * 4) * 64 + lt) * 4;
                            v = *reinterpret_cast<const floatx4*>(b);
                            v += *reinterpret_cast<const floatx4*>(b + 256);
                            v += *reinterpret_cast<const floatx4*>(b + 512);
                            v += *reinterpret_cast<const floatx4*>(b + 768);
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float delta = v[h] * st_dden[pi][h] + st_dmean[pi][h];              // denormalize, :349
                            if constexpr (NOISE != CADM_NOISE_NONE) {
                                float lv = st_mx[pi][h] - softplus_fast(st_mx[pi][h] - v[2 + h]);   // :356
                                lv = st_mn[pi][h] + softplus_fast(lv - st_mn[pi][h]);               // :357
                                const float sd = __expf((lv + st_dl2s[pi][h]) * 0.5f);              // :360-363
                                delta = delta + pz[mt][pi][h] * sd;                                 // :365
                            }
                            po[mt][pi][h] = postproc<ENV>(2 * dp + h, po[mt][pi][h], delta);        // :466
                        }
                        if (a.traj && valid[mt]) {
                            float* tp = a.traj + ((size_t)(t - 1) * a.m * a.n_local * a.p + lr[mt]) * D + 2 * dp;
                            tp[0] = po[mt][pi][0];
                            if (2 * dp + 1 < D) tp[1] = po[mt][pi][1];
                        }
                    }
                    if constexpr (ENV == CADM_ENV_CARTPOLE) {
                        if (t > 0) ret += reward_part<ENV>(dp, po[mt][pi][0], po[mt][pi][1], 0.0f);   // reads NEXT obs
                    } else {
                        if (t < H) ret += reward_part<ENV>(dp, po[mt][pi][0], po[mt][pi][1], ctrl_s[(mt * 16 + arow) * H + t]);
                    }
                    if (t < H) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float sn = 0.0f, cs = 0.0f;
                            if constexpr (ENV == CADM_ENV_HALFCHEETAH) {                 // the one trig pair (obs dim 2)
                                if (fx_op[pi][h][0] != 0) sincos_cw(po[mt][pi][h], &sn, &cs);
                            }
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                if (fx_off[pi][h][i] >= 0) {
                                    const float pv = fx_op[pi][h][i] == 1 ? sn : fx_op[pi][h][i] == 2 ? cs : po[mt][pi][h];
                                    x_in[mt * NC0 * 256 + fx_off[pi][h][i]] = (pv - fx_mean[pi][h][i]) * fx_inv[pi][h][i];   // :450-451
                                }
                            }
                        }
                    }
                }
            }
            if (t < H) {
#pragma unroll
                for (int ai = 0; ai < NAI; ++ai) {
                    const int ac = a0 + 16 * ai;
                    if (ac < A) {
                        float v = areg[mt][ai];
                        if (a.norm_actions) v = (v - smem[G::ST_ACT_MEAN + ac]) * smem[G::ST_ACT_DEN + ac];   // :443
                        put_x(mt, P + ac, v);
                        if (t + 1 < H) areg[mt][ai] = a.actions[abase[mt] + (t + 1) * A + ac];
                    }
                }
            }
        }
        if (t == H) break;
        TS(0)
        __syncthreads();
        TS(1)

        // ================= dense layers =================
        float* act_out = smem + G::ACT_A;
        float* act_in = nullptr;
        float* part_out = smem + G::PART_A;
        float* part_in = nullptr;
        floatx4 accF[MT][cmax(NF, 1)], accS[MT][cmax(NS, 1)];
        floatx4 bsplit[MT][cmax(NS, 1)];
        float bsplit_sel[MT][cmax(NS, 1)];
        floatx4 biasF[cmax(NF, 1)], biasS[cmax(NS, 1)];

        auto load_bias = [&](const float* btiles) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
                biasF[i] = *reinterpret_cast<const floatx4*>(btiles + ((wave * NF + i) * 64 + lane) * 4);
#pragma unroll
            for (int s = 0; s < NS; ++s)
                biasS[s] = *reinterpret_cast<const floatx4*>(btiles + ((4 * NF + s) * 64 + lane) * 4);
        };
        auto hidden_epilogue = [&]() {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    floatx4 v = accF[mt][i] + biasF[i];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = swish_f(v[r]);
                    *reinterpret_cast<floatx4*>(act_out + ((mt * NFT + wave * NF + i) * 64 + lane) * 4) = v;
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    floatx4 v = accS[mt][s];
                    if (wave == 0) v += biasS[s];
                    *reinterpret_cast<floatx4*>(part_out + (((mt * NS + s) * 4 + wave) * 64 + lane) * 4) = v;
                }
            }
        };
        // producer's K-split tiles: sum the 4 partials, activate.  Loads are issued one chunk before the
        // arithmetic; the per-wave k-step select is a masked sum (branch-free, interleaves between the MFMAs).
        floatx4 rb[MT][cmax(NS, 1)][4];
        floatx4 rv[MT][cmax(NS, 1)];
        auto rebuild_step = [&](int part) {     // 0: issue LDS reads, 1: sum, 2: activate r=0,1, 3: r=2,3, 4: k-step select
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (part == 0) {
#pragma unroll
                        for (int w = 0; w < 4; ++w)
                            rb[mt][s][w] = *reinterpret_cast<const floatx4*>(part_in + (((mt * NS + s) * 4 + w) * 64 + lane) * 4);
                    } else if (part == 1) {
                        rv[mt][s] = ((rb[mt][s][0] + rb[mt][s][1]) + rb[mt][s][2]) + rb[mt][s][3];
                    } else if (part == 2) {
                        rv[mt][s][0] = swish_f(rv[mt][s][0]);
                        rv[mt][s][1] = swish_f(rv[mt][s][1]);
                    } else if (part == 3) {
                        rv[mt][s][2] = swish_f(rv[mt][s][2]);
                        rv[mt][s][3] = swish_f(rv[mt][s][3]);
                        bsplit[mt][s] = rv[mt][s];
                    } else {
                        const floatx4 v = rv[mt][s];
                        bsplit_sel[mt][s] = ((v[0] * wsel[0] + v[1] * wsel[1]) + v[2] * wsel[2]) + v[3] * wsel[3];
                    }
                }
        };
        // Gaussian-head noise of THIS step for this thread's pairs, generated between the MFMAs of
        // the head pass (Philox4x32-10 round by round + Box-Muller), or fetched from the injected eps.
        uint32_t pc[MT][NPI][4], pk[MT][NPI][2];
        auto noise_part = [&](auto part_c) {    // 0: setup + load (inject), 1..10: one Philox round each, 11: Box-Muller
            constexpr int part = decltype(part_c)::value;
            if constexpr (NOISE == CADM_NOISE_NONE) return;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int pi = 0; pi < NPI; ++pi) {
                    const int dp = fg + 16 * pi;
                    if constexpr (NOISE == CADM_NOISE_INJECT) {
                        if (part == 0 && dp < NP) {
                            const float* ep = a.eps + ((size_t)t * a.m * a.n_local * a.p + lr[mt]) * D + 2 * dp;
                            pz[mt][pi][0] = ep[0];
                            pz[mt][pi][1] = (2 * dp + 1 < D) ? ep[1] : 0.0f;
                        }
                    } else if constexpr (part == 0) {   // branch-free: idle slots draw a value nobody reads
                        pc[mt][pi][0] = grow[mt]; pc[mt][pi][1] = (uint32_t)t; pc[mt][pi][2] = eps_group(dp);
                        pc[mt][pi][3] = CADM_STREAM_EPS | ((uint32_t)a.it << 8);
                        pk[mt][pi][0] = a.seed; pk[mt][pi][1] = a.call;
                    } else if constexpr (part <= 10) {
                        philox_rounds<part - 1, part>(pc[mt][pi], pk[mt][pi]);
                    } else {
                        const bool hi = eps_sub(dp) != 0;      // (rollout_env.h: one call = the noise of pairs dp and dp + 4)
                        box_muller(u01(hi ? pc[mt][pi][2] : pc[mt][pi][0]), u01(hi ? pc[mt][pi][3] : pc[mt][pi][1]), pz[mt][pi][0], pz[mt][pi][1]);
                    }
                }
        };
        auto no_side = [&](auto) {};

        {   // layer 0
            const unsigned wn = wmem_b + a.w_l0_b + wave * (a.w_lh_b / 4);
            load_bias(bmem);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { zero_acc(accF[mt]); zero_acc(accS[mt]); }
            mfma_pass<G, NC0, G::KSL0, NF, NS, NT, NF, NS, 0, 0u>(ring, rsrc, w0, wn, x_in, bsplit, bsplit_sel, accF, accS, wave,
                                                              lane, no_side);
            TS(2)
            hidden_epilogue();
            TS(3)
        }
        __syncthreads();
        TS(4)

        // hidden layers 1 .. NH-1 (the last one prefetches the head layer's stream, so it is peeled)
        auto hidden_layer = [&](int l, auto last_c) {
            constexpr bool LAST = decltype(last_c)::value;
            act_in = act_out;
            part_in = part_out;
            act_out = (act_in == smem + G::ACT_A) ? smem + G::ACT_B : smem + G::ACT_A;
            part_out = (part_in == smem + G::PART_A) ? smem + G::PART_B : smem + G::PART_A;
            const unsigned wc = wmem_b + a.w_l0_b + (l - 1) * a.w_lh_b + wave * (a.w_lh_b / 4);
            load_bias(bmem + a.b_l0 + (size_t)(l - 1) * a.b_lh);
            auto side = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j == 0) rebuild_step(0);
                if constexpr (j == 1) { rebuild_step(1); rebuild_step(2); rebuild_step(3); rebuild_step(4); }
            };
            TS(5)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { zero_acc(accF[mt]); zero_acc(accS[mt]); }
            if constexpr (!LAST) {
                mfma_pass<G, NT, G::KSLH, NF, NS, NT, NF, NS, NS, 0x2u>(ring, rsrc, wc, wc + a.w_lh_b, act_in, bsplit, bsplit_sel,
                                                                     accF, accS, wave, lane, side);
            } else {
                mfma_pass<G, NT, G::KSLH, NF, NS, G::OX_NCH, G::OX_NFO, G::OX_NSO, NS, 0x2u>(ring, rsrc, wc, wo, act_in, bsplit,
                                                                                          bsplit_sel, accF, accS, wave, lane, side);
            }
            TS(6)
            hidden_epilogue();
            TS(7)
            __syncthreads();
            TS(8)
        };
        for (int l = 1; l + 1 < a.NH; ++l) hidden_layer(l, std::false_type{});
        hidden_layer(a.NH - 1, std::true_type{});

        // ================= output heads (mu | logvar tiles) =================
        act_in = act_out;
        part_in = part_out;
        floatx4 hF[MT][cmax(NFO, 1)], hS[MT][cmax(NSO, 1)];
        floatx4 hbF[cmax(NFO, 1)], hbS[cmax(NSO, 1)];
#pragma unroll
        for (int i = 0; i < NFO; ++i)
            hbF[i] = *reinterpret_cast<const floatx4*>(bo + ((wave * NFO + i) * 64 + lane) * 4);
#pragma unroll
        for (int s = 0; s < NSO; ++s)
            hbS[s] = *reinterpret_cast<const floatx4*>(bo + ((4 * NFO + s) * 64 + lane) * 4);
        TS(9)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { zero_acc(hF[mt]); zero_acc(hS[mt]); }
        if constexpr (G::OCS) {
            auto side = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                auto np = [&](auto... ps) { (noise_part(std::integral_constant<int, decltype(ps)::value>{}), ...); };
                using std::integral_constant;
                if constexpr (j == 0) { rebuild_step(0); np(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{}, integral_constant<int, 3>{}); }
                if constexpr (j == 1) { rebuild_step(1); rebuild_step(2); rebuild_step(3); np(integral_constant<int, 4>{}, integral_constant<int, 5>{}, integral_constant<int, 6>{}); }
                if constexpr (j == 2) np(integral_constant<int, 7>{}, integral_constant<int, 8>{}, integral_constant<int, 9>{}, integral_constant<int, 10>{});
                if constexpr (j == 3) np(integral_constant<int, 11>{});
            };
            head_pass_csplit<G, NC0, NF, NS, 0xFu>(ring, rsrc, wo, w0, act_in, bsplit, hS, wave, lane, side);
        } else {
            auto side = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j <= 4) rebuild_step(j);
                if constexpr (j >= 1 && j <= 12) noise_part(std::integral_constant<int, j - 1>{});
            };
            mfma_pass<G, NT, G::KSLH, NFO, NSO, NC0, NF, NS, NS, 0x1FFEu>(ring, rsrc, wo, w0, act_in, bsplit, bsplit_sel, hF, hS, wave,
                                                                 lane, side);
        }
        TS(11)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int i = 0; i < NFO; ++i)
                *reinterpret_cast<floatx4*>(ofull + ((mt * 4 * NFO + wave * NFO + i) * 64 + lane) * 4) = hF[mt][i] + hbF[i];
#pragma unroll
            for (int s = 0; s < NSO; ++s) {
                floatx4 v = hS[mt][s];
                if (wave == 0) v += hbS[s];
                *reinterpret_cast<floatx4*>(opart + (((mt * NSO + s) * 4 + wave) * 64 + lane) * 4) = v;
            }
        }
        TS(12)
        __syncthreads();
        TS(13)
    }
    TS_DUMP

    // ---- a row's return = sum of its threads' reward parts, in fixed slot order ----
    float* ret_s = smem + G::ACT_A;
    ret_s[arow * 16 + fg] = ret;
    __syncthreads();
    if (fg == 0 && valid[0]) {
        float r = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) r += ret_s[arow * 16 + i];
        a.returns_rows[lr[0]] = r;
    }
}

template <class G, int NOISE>
int launch_noise(cadm_ctx* ctx, const RolloutArgs& a, int rows_per_member, hipStream_t s) {
    RolloutArgs args = a;
    const int tiles = (rows_per_member + 15) / 16;
    args.wgs_per_member = (tiles + G::MT - 1) / G::MT;
    args.rows_per_member = rows_per_member;
    const size_t lds = ((size_t)G::CTRL_S + (size_t)G::MT * 16 * a.H) * sizeof(float);
    if (lds > 160 * 1024) {
        cadm_set_error("rollout: horizon %d needs %zu B of LDS (> 160 KiB)", a.H, lds);
        return CADM_EINVAL;
    }
    const void* fn = reinterpret_cast<const void*>(&rollout_kernel<G, NOISE>);
    if (!ctx->attr_done.count(fn)) {    // per ctx = per device (the attribute is a per-device property)
        CADM_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_done.insert(fn);
    }
    hipLaunchKernelGGL((rollout_kernel<G, NOISE>), dim3(args.wgs_per_member * ctx->E), dim3(256), lds, s, args);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

template <class G>
int launch(cadm_ctx* ctx, const RolloutArgs& a, int rows_per_member, hipStream_t s) {
    if (a.deterministic) return launch_noise<G, CADM_NOISE_NONE>(ctx, a, rows_per_member, s);
    if (a.eps) return launch_noise<G, CADM_NOISE_INJECT>(ctx, a, rows_per_member, s);
    return launch_noise<G, CADM_NOISE_PHILOX>(ctx, a, rows_per_member, s);
}

}  // namespace
