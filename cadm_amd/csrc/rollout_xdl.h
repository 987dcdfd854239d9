#pragma once
// Fused trajectory-sampling rollout on the f16 matrix pipe with fp32-equivalent accuracy ("xdl" kernel).
// ONE launch advances every (candidate, particle) row through the whole horizon -- input assembly, the 6-matmul
// ensemble MLP, Gaussian head, state update and reward accumulation (reference core/utils.py:431-472).
//
// Why not v_mfma_f32_16x16x4_f32: on gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate (1/16 of the f16 rate) and
// a wave cannot issue any VALU work in its shadow (tools/issue_bench: +15 cycles for the first VALU op after an MFMA, +4 for
// each further one).  Every fp32 operand is split in two f16 numbers (xdl_geo.h): 3 v_mfma_f32_16x16x32_f16 per 16x16x32
// block with fp32 accumulation reproduce the fp32 product to 2^-22 in 3/16 of the matrix-pipe time.
//
// Mapping:
//   * workgroup = 8 waves (two per SIMD, 256 registers each) = MT x 16 rows of ONE ensemble member (MT = 1, or 2 for
//     large batches); a workgroup walks over groups of MT row tiles grp, grp + wgs_per_member, ..  of its member (one
//     tile at BASELINE cfg2).  Two waves per SIMD: while one waits (LDS operand loads, the weight stream, an epilogue's
//     exp/rcp chain) the other one's MFMAs run;
//   * a layer is evaluated transposed, OUT^T = W^T IN^T: weights are the A operand (a third register-resident, the rest
//     streamed from L2 in consumption order through a register ring), the 16 rows of a tile are the B / D columns.  A
//     lane's D fragments of tiles (2c, 2c+1) are its B fragment of chunk c of the next layer: activations cross layers
//     through LDS with lane-linear accesses;
//   * wave w owns BASE + (w < EXTRA) hidden tiles and computes them CADM_XDL_GROUP at a time over the whole K (the B
//     operand is read chunk by chunk from LDS, ahead of its use); the last group's epilogue (swish, f16 split, LDS store)
//     overlaps with the SIMD's other wave, earlier groups' with the next group's MFMAs;
//   * MT = 1: the rollout state lives in registers of the 256 "feature threads" (waves 0-3), their twins in waves 4-7
//     produce the Gaussian-head noise; MT = 2: all 512 threads hold state (waves 4-7: the second row tile).
#include "rollout_args.h"
#include "rollout_env.h"
#include "xdl_geo.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

#ifndef CADM_XDL_RING
#define CADM_XDL_RING 4
#endif
// TIMING EXPERIMENTS ONLY (tools/build_variant.sh; wrong results, never in the product build): what a part of the kernel costs.
//   CADM_XDL_EXPERIMENT_NOBAR   no barriers between the dense layers (an upper bound for any finer-grained layer hand-off)
//   CADM_XDL_EXPERIMENT_NOMFMA  the sweeps issue no MFMAs (everything else -- streams, LDS traffic, epilogues, state phase -- stays)
//   CADM_XDL_EXPERIMENT_NOSTREAM the streamed weight fragments are not loaded (the ring keeps stale registers)
//   CADM_XDL_EXPERIMENT_NOEPI   the hidden tiles' epilogue arithmetic is skipped (stages 0-4: nonlinearity and f16 split; the LDS store stays)
#ifdef CADM_XDL_EXPERIMENT_NOBAR
#define XDL_LAYER_SYNC() ((void)0)
#else
#define XDL_LAYER_SYNC() __syncthreads()
#endif

// MT = row tiles (of 16 rows) a workgroup advances together.  2 for large batches: every weight fragment then feeds two sets
// of MFMAs (half the weight stream, half the barriers and sweep start-ups per row), all 512 threads hold rollout state.
// NH = number of hidden layers, compile-time: the per-layer stream offsets, residency tables and the layer loop fold to
// constants (runtime NH cost 3 % at cfg2: ~100 scalar instructions per sweep boundary and SGPR spills).  ACT = hidden
// nonlinearity (CADM_ACT_*; the reference's `_activations`, dynamics.py:17-24).  Geometries that are not compiled in are
// built on demand from rollout_jit.hip (cadm_amd/jit.py).
template <int ENV_, int C_, int HID_, int MT_, int NH_, int ACT_>
struct XC {
    static constexpr int ENV = ENV_, C = C_, HID = HID_, MT = MT_, NHC = NH_, ACT = ACT_;
    static_assert(NH_ >= 1 && NH_ <= CADM_MAX_HIDDEN_LAYERS, "number of hidden layers");
    static constexpr int D = env_D(ENV), A = env_A(ENV), P = env_P(ENV);
    static constexpr int K0 = P + A + C;
    static constexpr int NC0 = (K0 + 31) / 32;            // chunks of layer 0
    // A feature slot behind the last real input feature (its packed weights are zeros: pack_xdl.hip): the state phase's feature writes
    // are UNCONDITIONAL -- a dim that feeds fewer than two features writes 0.0 there instead of branching around the write (hipcc turned
    // the guarded writes + the id / sin / cos select into nested exec-mask regions with out-of-line blocks: ~25 control instructions per
    // feature, 300 per step of the wave-tile kernel).  Exists unless K0 is a multiple of 32.
#ifndef CADM_XDL_SPARE
#define CADM_XDL_SPARE 1
#endif
    static constexpr bool SPARE = CADM_XDL_SPARE && (K0 % 32) != 0;
    static constexpr int NT = (HID + 15) / 16;            // hidden tiles
    // invariant last chunk of layer 0 (xdl_geo.h): accumulated FIRST by every flavour, once per row tile by this kernel (NC0S chunks per step)
    static constexpr bool INV = xdl_inv0(K0, P + A, HID);
    static constexpr int NC0S = NC0 - (INV ? 1 : 0);
    static constexpr int NCH = (NT + 1) / 2;              // chunks of a layer that consumes a hidden layer
    static constexpr int NTO = (D + 7) / 8;               // head tiles (8 dims: mu | lv)
    static constexpr int NW = CADM_XDL_WAVES, NTHR = NW * 64;
    static constexpr int BASE = NT / NW, EXTRA = NT % NW;
    static constexpr int NTOW = (NTO + NW - 1) / NW;      // head tile slots per wave
    static_assert(NTOW == 1, "one head tile per wave at most (obs dim <= 64)");
    static constexpr int R = CADM_XDL_RING;               // ring depth (fragments)
    // products per 16x16x32 block: hi += w1 x1 + w1 x2, lo += w2 x1 [+ w2 x2] (xdl_geo.h).  The dropped w2 x2 term is 2^-22 of a
    // product; wide layers (K > 256) accumulate enough of them to show at the 1e-5 parity bar, so they take the 4th product.
    static constexpr int NPROD = HID > 256 ? 4 : 3;
    static constexpr int NP = (D + 1) / 2, NPI = (NP + 15) / 16, NAI = (A + 15) / 16;
    static_assert(BASE >= 1, "hidden width too small for the 8-wave tile split (>= 128)");
    // LDS carve (bytes)
    static constexpr int XIN = 0;                                  // [MT][2 parts][NC0][64 lanes] x 16 B
    static constexpr int XIN_T = 2 * NC0 * 1024, ACT_T = 2 * NCH * 1024, OFULL_T = NTO * 1024;      // bytes per row tile
    static constexpr int ACTA = XIN + MT * XIN_T;                  // [MT][2][NCH][64] x 16 B
    static constexpr int ACTB = ACTA + MT * ACT_T;
    static constexpr int OFULL = ACTB + MT * ACT_T;                // [MT][NTO][64] x float4
    static constexpr int STATS = OFULL + MT * OFULL_T;             // floats
    static constexpr int ST_OBS_MEAN = 0, ST_OBS_DEN = P, ST_ACT_MEAN = 2 * P, ST_ACT_DEN = 2 * P + A;
    // per feature-slot constants (head statistics, derived-feature slots): read from LDS in every state phase instead of
    // being held in ~26 registers across the MFMA sweeps
    static constexpr int TABW = 28;                                     // floats per (pair slot, fg) entry
    static constexpr int TAB = STATS + rup((2 * P + 2 * A) * 4, 16);
    // One observation dim per thread (instead of a pair) for the state update of small observation spaces, one row tile:
    // thread (row, d) with d = tid / 16 < D owns dim d, the LAST NP of the 32 dim slots make the Gaussian-head noise of one pair
    // each.  The state update is one wave's dependent chain (softplus -> exp -> .. -> f16 split): half the dims per thread halve
    // it, over 4.5 of the 8 waves instead of 2.25.  A row's arithmetic is unchanged (bit-identical to the pair layout).
#ifndef CADM_XDL_ONED
#define CADM_XDL_ONED 1
#endif
    static constexpr bool ONED = CADM_XDL_ONED && MT == 1 && NPI == 1 && D + NP <= 32 && A <= D;
    static constexpr int TABD = 16;                                     // ONED: floats per dim entry
    static constexpr int TAB_BYTES = ONED ? 32 * TABD * 4 : NPI * 16 * TABW * 4;
    // Gaussian-head noise, produced by waves 4-7 while waves 0-3 run the state update: [step parity][pair slot][fg][row] x 2
    static constexpr int ZB = TAB + TAB_BYTES;
    static constexpr int ZB_T = 2 * NPI * 256 * 8;                      // bytes per row tile
    static constexpr int CTRL = ZB + MT * ZB_T;                         // + MT * 16 * H floats (dynamic)
    // Bias tiles (fp32, D layout) live in LDS when they fit next to the rest (a bias read from global memory in a tile's
    // epilogue would sit BEHIND the ring's weight loads in the in-order vmcnt queue and drain the whole ring);
    // otherwise they are fetched at the start of a sweep, ahead of that sweep's ring loads.
#ifndef CADM_XDL_RES
#define CADM_XDL_RES 1
#endif
    // register-resident weights (loaded once per workgroup, never re-read from L2): hidden layer 1 on every wave and
    // the head tile on the waves that own BASE hidden tiles -- ~100 of a wave's 256 registers at HID = 200.
    // RES_FRAGS leaves room for the ring, the accumulators and what hipcc parks in AGPRs itself
    // (tests/test_isa_hygiene.py checks the ISA for AGPR<->VGPR shuffles of resident fragments, which would also be an
    // undetected MFMA operand hazard).
#ifndef CADM_XDL_RES_FRAGS
#define CADM_XDL_RES_FRAGS 16       // waves with one hidden tile (+ a head tile)
#endif
#ifndef CADM_XDL_RES_MT2_LESS
#define CADM_XDL_RES_MT2_LESS 6     // two row tiles: twice the accumulators, operand registers and rollout state
#endif
    // waves with two or more hidden tiles (more accumulators / epilogue state live): 14 until round 5.  Round 6 took the f16-range clamps and
    // the exec-mask regions out of the epilogue and the state phase; what that freed holds one more fragment in every compiled-in geometry
    // (15: all 240 kernels of the library free of scratch and of AGPR copies, tests/test_isa_hygiene.py) and two more in the reference's
    // own geometry (halfcheetah, context 10, hidden 200: 16; the same count spills 2-24 registers in four other geometries).  Same-box:
    // cfg2 153.2 -> 152.2 -> 151.1 us per launch.  Geometries built on demand (jit.py) keep 14: nobody has looked at their code.
#ifdef CADM_XDL_RES_FRAGS_X
    static constexpr int RES_X = CADM_XDL_RES_FRAGS_X;
#elif defined(CADM_JIT_MODULE)
    static constexpr int RES_X = 14;
#else
    static constexpr int RES_X = (INV && NPI > 1) ? 14 : 15;      // (the reference's own geometry held 16 until the invariant-chunk prologue took the registers' last slack; wide observations with it: one less)
#endif
    static constexpr bool ASM_MFMA = CADM_XDL_RES && NCH <= 8;     // asm MFMAs (AGPR-resident operands) vs builtins
    static constexpr int res_frags(int ntw) {      // (wide observations keep two pair slots of rollout state per thread)
        return (!CADM_XDL_RES || NCH > 8) ? 0
               : MT > 1 ? (ntw >= 2 ? RES_X : CADM_XDL_RES_FRAGS) - (NPI > 1 ? 2 : 0) - CADM_XDL_RES_MT2_LESS - (NT >= 15 ? 1 : 0)
                        : (ntw >= 2 ? RES_X : CADM_XDL_RES_FRAGS) - (NPI > 1 ? 2 : 0) - (NT >= 15 ? 2 : 0);
    }
    static constexpr int MAX_NH_LDS = NH_;
    // (a bias tile in D layout repeats each of its 16 values over the tile's 16 data rows: LDS keeps one copy, 64 B per tile)
    static constexpr int BIAS_TILE_B = 64;
    static constexpr int BIAS_BYTES = (MAX_NH_LDS * NT + NTO) * BIAS_TILE_B;
    static constexpr bool BIAS_LDS = CTRL + MT * 16 * 64 * 4 + BIAS_BYTES <= 154 * 1024;
    // (one row tile: a lookahead of 2 chunks bought nothing measurable, and its 8 registers are worth one more resident
    //  fragment: every streamed fragment costs ~0.6 us per launch at cfg2 -- the L2 -> CU weight stream is what the one-tile kernel waits for)
    static constexpr int XDEPTH = 2;                       // B-operand chunks in registers (lookahead XDEPTH - 1)
    // One-tile sweeps (the head tile; the hidden tiles of the waves that own one): a chunk is 3 MFMAs = 48 cycles of this wave's
    // own work against ~130 cycles of LDS latency, so with one chunk of lookahead the sweep runs at the LDS latency (7 x 130
    // cycles where the MFMAs need 336) -- and the HEAD sweep is on the step's critical path (every other wave waits for it:
    // profiles/r4_phase_timing_skeleton.txt).  They keep more operand chunks in flight.
#ifndef CADM_XDL_XD1
#define CADM_XDL_XD1 2      // (measured, same box: 3 / 4 chunks +-0, 7 chunks +1 %: the head is not waiting for its operands)
#endif
    static constexpr int XDEPTH1 = MT > 1 ? 2 : CADM_XDL_XD1;
    // LDS-resident weight fragments: what is left of the 160 KiB (at horizons <= 128; one row tile per workgroup) holds the LAST
    // LQ_SLOTS fragments of hidden layers 1..3 of every wave with two or more tiles -- the waves a layer waits for.  The one-tile
    // kernel is bound by the L2 -> CU weight stream (every streamed fragment costs ~0.6 us per launch at cfg2).
#ifndef CADM_XDL_LQ_MAX
#define CADM_XDL_LQ_MAX 2       // (a third slot measured +-0: the stream is no longer what the layer waits for)
#endif
    static constexpr int NEL = BASE >= 2 ? NW : EXTRA;                 // waves with >= 2 tiles (waves 0 .. NEL-1)
    static constexpr int INV_BYTES = INV ? MT * NT * CADM_XDL_FRAG_BYTES : 0;      // (HI, LO) of "bias + invariant chunk" per (row tile, tile): 64 lanes x 16 B each
    static constexpr int LQ_FREE = 160 * 1024 - (CTRL + rup(MT * 16 * 128 * 4, 16) + (BIAS_LDS ? BIAS_BYTES : 0)) - INV_BYTES;
    static constexpr int LQ_SLOTS = (MT > 1 || NEL == 0 || !ASM_MFMA || LQ_FREE <= 0) ? 0 : cmin(CADM_XDL_LQ_MAX, LQ_FREE / (NEL * 3 * CADM_XDL_FRAG_BYTES));
    static constexpr int LQ_BYTES = NEL * 3 * LQ_SLOTS * CADM_XDL_FRAG_BYTES;
    static size_t lds_bytes(int H) {                       // dynamic LDS of a launch
        return (size_t)CTRL + (size_t)rup(MT * 16 * H * 4, 16) + (BIAS_LDS ? (size_t)BIAS_BYTES : 0) + LQ_BYTES + INV_BYTES;
    }
};

template <class G>
struct XRing {
    uintx4 w[G::R][2];
};

template <int SLOT, class G>
__device__ __forceinline__ void xring_load(XRing<G>& ring, __amdgpu_buffer_rsrc_t rsrc, unsigned soff, int lane) {
#ifdef CADM_XDL_EXPERIMENT_NOSTREAM      // (timing experiment: the L2 -> CU weight stream is not issued; the ring holds whatever it held)
    asm volatile("" : "+v"(ring.w[SLOT][0]), "+v"(ring.w[SLOT][1]));
#else
    ring.w[SLOT][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, soff, 0);
    ring.w[SLOT][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + 1024, soff, 0);
#endif
}

__device__ __forceinline__ floatx4 xmfma(uintx4 a, f16x8 b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), b, c, 0, 0, 0);
}

// Register-RESIDENT fragments live in AGPRs for the whole kernel.  hipcc treats an AGPR-held MFMA operand as a VGPR value
// "spilled to AGPR" and reloads it (4 x v_accvgpr_read) before every use, so the resident path names the register class
// itself: the load writes AGPRs ("=a") and the MFMA reads its A operand from them ("a").  The compiler sees neither the
// load (the caller waits with an explicit vmcnt(0)) nor the MFMA's latency (callers keep the accumulator's first VALU
// read a whole chunk of MFMAs away, or pad with xdl_result_nops).
__device__ __forceinline__ void xres_load(uintx4& dst, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    // (readfirstlane: under SGPR pressure hipcc keeps a uniform offset in a VGPR, which the "s" constraint does not move
    //  back.  s_nop 4: the hazard recognizer does not look into inline asm, and a v_readfirstlane'd SGPR needs 5 wait
    //  states before a VMEM instruction may read it -- without them the load used the PREVIOUS offset now and then)
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen"
                 : "=a"(dst) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
__device__ __forceinline__ void xmfma_res(floatx4& acc, const uintx4& w, const f16x8& x) {
#ifdef CADM_XDL_EXPERIMENT_NOMFMA
    asm volatile("" : "+v"(acc) : "a"(w), "v"(x));
#else
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
#endif
}
// Streamed fragments (ring registers, VGPRs) go through the same asm form so that ALL MFMAs of a sweep keep their
// accumulators in VGPRs: a mix of asm and builtin MFMAs makes hipcc shuttle accumulators between VGPRs and AGPRs
// right behind an MFMA whose latency it cannot see.  (Loads feeding the asm are still tracked: hipcc places the
// s_waitcnt for any register an asm statement reads.)
// ASM = false (geometries without resident fragments): plain builtin, hipcc then handles every hazard itself.
// "s_nop 1" in front of every asm MFMA: under register pressure hipcc parks VGPR values (a ring slot, an operand chunk, an
// accumulator) in spare AGPRs and copies them back with v_accvgpr_read right in front of the statement that reads them; a
// VALU-written VGPR needs 2 wait states before an MFMA may read it, and the hazard recognizer does not look into inline asm.
// (Found by tools/fuzz_rollout.py: cart-pole / pendulum at HID = 256 with two row tiles read a stale ring slot.)  The two
// cycles hide behind the SIMD's other wave: measured neutral.
template <bool ASM>
__device__ __forceinline__ void xmfma_ring(floatx4& acc, const uintx4& w, const f16x8& x) {
#ifdef CADM_XDL_EXPERIMENT_NOMFMA
    asm volatile("" : "+v"(acc) : "v"(w), "v"(x));
#else
    if constexpr (ASM) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
    else acc = xmfma(w, x, acc);
#endif
}
// Hazard padding the compiler cannot place for asm MFMAs.  The accumulators are "+v" operands of the padding statement,
// so every instruction that defines them (the zeroing moves) stays before it and every reader (the epilogue) after it.
// (the third accumulator exists only with 4 products per block: naming it here would keep 4 dead registers per tile alive)
template <bool LL>
__device__ __forceinline__ void xdl_result_nops(floatx4& a, floatx4& b, floatx4& c) {      // XDL write -> VALU read (4-pass op)
    if constexpr (LL) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c));
    else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(a), "+v"(b));
}
template <bool LL>
__device__ __forceinline__ void xdl_operand_nops(floatx4& a, floatx4& b, floatx4& c) {     // VALU write -> XDL read as srcC
    if constexpr (LL) asm volatile("s_nop 4" : "+v"(a), "+v"(b), "+v"(c));
    else asm volatile("s_nop 4" : "+v"(a), "+v"(b));
}

// split an ACTIVATION for the f16 pipe: hi = f16(v), lo = f16(v - hi), UNSCALED (xdl_geo.h: the products w1 * lo accumulate next to
// w1 * hi; a low part below the f16 normal range is a subnormal with 2^-24 absolute spacing: an absolute error of 3e-8 on an
// activation that small)
__device__ __forceinline__ void xsplit(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// Epilogue of a hidden tile: swish, f16 split, store as (half of) a B fragment of the next layer (the bias tile is the
// accumulator's initial value).  Cut in STAGES of mutually independent instructions (stage s of every value before
// stage s+1 of any): a wave issues in order, so a VALU op waiting for its predecessor's result (v_exp -> v_add -> v_rcp ..)
// would hold up the MFMAs queued behind it.
// PACKED: the arithmetic on pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: 24 VALU instructions per tile) or on scalars (30).  Packed fp32 does
// NOT co-execute with the matrix pipe on gfx950 -- one v_pk_* behind a v_mfma_f32_16x16x32_f16 costs the wave +19 cycles, a wave of packed work
// runs at the SUM of its own time and its SIMD partner's MFMA time -- plain v_fma_f32 / v_exp_f32 / v_cvt overlap 0.6-0.9 with the partner's
// MFMAs and two of them hide behind each of the wave's own (tools/micro/gen_coissue_bench.py, profiles/r5_coissue_microbench.md; rounds 3-4
// concluded "VALU and MFMA do not overlap" from micro-benchmarks hipcc had SLP-packed).  Which form is faster is a property of the kernel:
// the wave-tile kernel's waves drift apart inside a block, an epilogue meets the partner's MFMAs: scalar, cfg3 -1.8 %; the cooperative
// kernel's waves run the same phase between the same barriers, the shorter epilogue wins: packed, cfg2 -0.5..1.5 % (same-box A/B,
// profiles/r5_wave_tile_experiments.md).  The rollout translation units are compiled with -fno-slp-vectorize so hipcc does not re-pack scalars.
#ifndef CADM_EPI_PACKED_COOP
#define CADM_EPI_PACKED_COOP 1
#endif
#ifndef CADM_EPI_PACKED_WT
#define CADM_EPI_PACKED_WT 0
#endif
template <class G, bool PACKED = CADM_EPI_PACKED_COOP>
struct XHiddenEpi {
    static constexpr int NSTAGE = 6;
    struct StateP { floatx2 v[2], s[2]; f16x4 h1, h2; };
    struct StateS { float v[4], s[4]; f16x4 h1, h2; };      // (scalars: a pair type invites the instruction selector to re-pack)
    using State = std::conditional_t<PACKED, StateP, StateS>;
    unsigned char* xsmem;
    const float* xb;
    int bias_off;
    int layer, out, tstart, lane;
    const unsigned char* inv = nullptr;      // layer 0 of a geometry with an invariant last chunk (xdl_geo.h): (HI, LO) of "bias + that chunk" per (row tile, tile)
    __device__ __forceinline__ floatx4 init(int ti, int hh) const {       // accumulator HI: bias tile (fp32, D layout) of local tile ti, or the tile's invariant start
        if (inv) return *reinterpret_cast<const floatx4*>(inv + ((hh * G::NT + tstart + ti) * 2 + 0) * 1024 + lane * 16);
        const int tile = layer * G::NT + tstart + ti;
        const int bt = tile * 64 + lane;
        // (a select between an LDS and a global POINTER would become a flat load with a full vmcnt/lgkmcnt drain)
        if constexpr (G::BIAS_LDS) return *reinterpret_cast<const floatx4*>(xsmem + bias_off + tile * G::BIAS_TILE_B + (lane >> 4) * 16);
        else return *reinterpret_cast<const floatx4*>(xb + bt * 4);
    }
    __device__ __forceinline__ floatx4 init_lo(int ti, int hh) const {    // accumulator LO: zero, or the tile's invariant start
        if (inv) return *reinterpret_cast<const floatx4*>(inv + ((hh * G::NT + tstart + ti) * 2 + 1) * 1024 + lane * 16);
        return floatx4{0.f, 0.f, 0.f, 0.f};
    }
    static __device__ __forceinline__ floatx2 lo2(const floatx4& x) { return __builtin_shufflevector(x, x, 0, 1); }
    static __device__ __forceinline__ floatx2 hi2(const floatx4& x) { return __builtin_shufflevector(x, x, 2, 3); }
    template <int S>
    __device__ __forceinline__ void stage(int ti, int hh, const floatx4& hi, const floatx4& lo, const floatx4& ll, State& st) const {
#ifdef CADM_XDL_EXPERIMENT_NOEPI
        if constexpr (S < 5) {      // (one compare keeps the wait for the accumulators; the stored activations are zeros)
            if constexpr (S == 0) { const _Float16 z = (_Float16)((hi[0] == 12345.678f && lo[0] == 1.5f) ? 1.0f : 0.0f); st.h1 = f16x4{z, z, z, z}; st.h2 = st.h1; }
            return;
        }
#endif
        constexpr float KE = G::ACT == CADM_ACT_TANH ? -2.0f * 1.4426950408889634f : -1.4426950408889634f;
        constexpr bool SIG = G::ACT != CADM_ACT_RELU && G::ACT != CADM_ACT_NONE;       // nonlinearities built on sigmoid
        if constexpr (PACKED) {
        if constexpr (S == 0) {            // pre-activation (hi + 2^-11 lo), exp2 argument
            const floatx2 c11 = {4.8828125e-4f, 4.8828125e-4f};
            const floatx2 ke = {KE, KE};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const floatx2 pre = __builtin_elementwise_fma(q ? hi2(lo) : lo2(lo), c11, q ? hi2(hi) : lo2(hi));
                st.v[q] = pre;                 // (no f16-range clamp: the conversions below saturate, fp16_saturate_on)
                // swish: the packed weights carry log2(e) (xdl_geo.h: CADM_XDL_SWISH_FOLD), pre IS the exp2 argument up to its sign
                if constexpr (G::ACT == CADM_ACT_SWISH) st.s[q] = -pre;
                else st.s[q] = pre * ke;
            }
        } else if constexpr (S == 1) {
            if constexpr (SIG) {
#pragma unroll
                for (int q = 0; q < 2; ++q) { st.s[q][0] = __builtin_amdgcn_exp2f(st.s[q][0]); st.s[q][1] = __builtin_amdgcn_exp2f(st.s[q][1]); }
            }
        } else if constexpr (S == 2) {
            if constexpr (SIG) {
                const floatx2 one = {1.0f, 1.0f};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    st.s[q] = st.s[q] + one;
                    st.s[q][0] = __builtin_amdgcn_rcpf(st.s[q][0]);
                    st.s[q][1] = __builtin_amdgcn_rcpf(st.s[q][1]);
                }
            }
        } else if constexpr (S == 3) {     // the nonlinearity (dynamics.py:17-24) and the high f16 part
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if constexpr (G::ACT == CADM_ACT_SWISH) st.v[q] = st.v[q] * st.s[q];                 // x * sigmoid(x), :23
                else if constexpr (G::ACT == CADM_ACT_SIGMOID) st.v[q] = st.s[q];                    // 1 / (1 + e^-x)
                else if constexpr (G::ACT == CADM_ACT_TANH) {      // 2 sigmoid(2x) - 1 (s was built from 2x); odd series near 0, where that cancels
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const float x = st.v[q][r], x2 = x * x;
                        const float ser = x * fmaf(x2, fmaf(x2, fmaf(x2, -0.05396825396825397f, 0.13333333333333333f), -0.3333333333333333f), 1.0f);
                        st.v[q][r] = fabsf(x) < 0.1f ? ser : fmaf(2.0f, st.s[q][r], -1.0f);
                    }
                } else if constexpr (G::ACT == CADM_ACT_RELU) { st.v[q][0] = fmaxf(st.v[q][0], 0.0f); st.v[q][1] = fmaxf(st.v[q][1], 0.0f); }
                st.h1[2 * q] = (_Float16)st.v[q][0];
                st.h1[2 * q + 1] = (_Float16)st.v[q][1];
            }
        } else if constexpr (S == 4) {     // low part: h - hi = fma(hi, -1, h), exact in fp32, rounded once to f16 (unscaled: xsplit)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const floatx2 t = st.v[q];
                const f16x2 hp = {st.h1[2 * q], st.h1[2 * q + 1]};
                f16x2 lp;
                asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                    : "=&v"(lp) : "v"(hp), "s"(-1.0f), "v"(t[0]), "v"(t[1]));
                st.h2[2 * q] = lp[0];
                st.h2[2 * q + 1] = lp[1];
            }
        }
        } else {
        if constexpr (S == 0) {            // pre-activation (hi + 2^-11 lo); the exp2 argument of tanh / sigmoid
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st.v[r] = fmaf(lo[r], 4.8828125e-4f, hi[r]);      // (no f16-range clamp: the conversions below saturate, fp16_saturate_on)
                if constexpr (SIG && G::ACT != CADM_ACT_SWISH) st.s[r] = st.v[r] * KE;
            }
        } else if constexpr (S == 1) {
            // swish: the packed weights carry log2(e) (xdl_geo.h), the pre-activation IS the exp2 argument up to its sign (a source modifier)
            if constexpr (SIG) {
#pragma unroll
                for (int r = 0; r < 4; ++r) st.s[r] = __builtin_amdgcn_exp2f(G::ACT == CADM_ACT_SWISH ? -st.v[r] : st.s[r]);
            }
        } else if constexpr (S == 2) {
            if constexpr (SIG) {
#pragma unroll
                for (int r = 0; r < 4; ++r) st.s[r] = __builtin_amdgcn_rcpf(st.s[r] + 1.0f);
            }
        } else if constexpr (S == 3) {     // the nonlinearity (dynamics.py:17-24) and the high f16 part
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (G::ACT == CADM_ACT_SWISH) st.v[r] = st.v[r] * st.s[r];                 // x * sigmoid(x), :23
                else if constexpr (G::ACT == CADM_ACT_SIGMOID) st.v[r] = st.s[r];                    // 1 / (1 + e^-x)
                else if constexpr (G::ACT == CADM_ACT_TANH) {      // 2 sigmoid(2x) - 1 (s was built from 2x); odd series near 0, where that cancels
                    const float x = st.v[r], x2 = x * x;
                    const float ser = x * fmaf(x2, fmaf(x2, fmaf(x2, -0.05396825396825397f, 0.13333333333333333f), -0.3333333333333333f), 1.0f);
                    st.v[r] = fabsf(x) < 0.1f ? ser : fmaf(2.0f, st.s[r], -1.0f);
                } else if constexpr (G::ACT == CADM_ACT_RELU) st.v[r] = fmaxf(st.v[r], 0.0f);
                st.h1[r] = (_Float16)st.v[r];
            }
        } else if constexpr (S == 4) {     // low part: h - hi = fma(hi, -1, h), exact in fp32, rounded once to f16 (unscaled: xsplit)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f16x2 hp = {st.h1[2 * q], st.h1[2 * q + 1]};
                f16x2 lp;
                asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                    : "=&v"(lp) : "v"(hp), "s"(-1.0f), "v"(st.v[2 * q]), "v"(st.v[2 * q + 1]));
                st.h2[2 * q] = lp[0];
                st.h2[2 * q + 1] = lp[1];
            }
        }
        }
        if constexpr (S == 5) {
            const int Tg = tstart + ti;
            unsigned char* dst = xsmem + out + hh * G::ACT_T + ((Tg >> 1) * 64 + lane) * 16 + (Tg & 1) * 8;
            *reinterpret_cast<f16x4*>(dst) = st.h1;
            *reinterpret_cast<f16x4*>(dst + G::NCH * 1024) = st.h2;
        }
    }
};

// Epilogue of a head tile (mu0 mu1 lv0 lv1 of 2 dims per lane): recombine and hand to the state phase through LDS.
template <class G>
struct XHeadEpi {
    static constexpr int NSTAGE = 1;
    struct State {};
    unsigned char* xsmem;
    const float* xb;
    int bias_off, bias_tile, ht, lane;
    __device__ __forceinline__ floatx4 init_lo(int, int) const { return floatx4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ floatx4 init(int, int) const {
        const int bt = bias_tile * 64 + lane;
        if constexpr (G::BIAS_LDS) return *reinterpret_cast<const floatx4*>(xsmem + bias_off + bias_tile * G::BIAS_TILE_B + (lane >> 4) * 16);
        else return *reinterpret_cast<const floatx4*>(xb + bt * 4);
    }
    template <int S>
    __device__ __forceinline__ void stage(int, int hh, const floatx4& hi, const floatx4& lo, const floatx4& ll, State&) const {
        floatx4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = fmaf(lo[r], 4.8828125e-4f, hi[r]);
        }
        *reinterpret_cast<floatx4*>(xsmem + G::OFULL + hh * G::OFULL_T + (ht * 64 + lane) * 16) = v;
    }
};

// One layer sweep of this wave: NTW tiles x NCHL chunks, GS tiles at a time.  SIDE: a group's epilogue runs stage by stage
// between the next group's MFMAs (only the last one is exposed); !SIDE: every group's epilogue right behind its MFMAs.
//   The first NRES fragments (consumption order) are register-resident (res[j][part], AGPRs, loaded once per workgroup);
//   the other NFS = NF - NRES come through the ring: it holds streamed fragments 0..R-1 of this layer on entry and 0..R-1
//   of the NEXT streamed layer (nx_nf of them exist) on exit.  wcur = byte offset of this layer's first STREAMED fragment.
//   The B operand (the 16 rows' activations) is read from LDS chunk by chunk, XD-1 chunks ahead of its use.
//   The epilogue of a tile group runs stage by stage between the MFMAs of the NEXT group (f16 MFMAs hide independent
//   VALU work of the same wave); the last group's epilogue overlaps with the SIMD's other wave.
//   NLDS: the LAST NLDS fragments of the layer (consumption order) are LDS-resident (lq: this wave's copy, made once per workgroup):
//   they go through the ring like streamed ones, from LDS instead of L2.  The first R ring fragments stay streamed.
template <class G, int NTW, int NCHL, int NRES, int GS, bool SIDE, int NLDS, int NCB = NCHL, class Epi>
__device__ __forceinline__ void xdl_sweep(XRing<G>& ring, const uintx4 (*res)[2], __amdgpu_buffer_rsrc_t rsrc, unsigned wcur,
                                          unsigned wnext, int nx_nf, const unsigned char* lds_in, int lane, const Epi& epi,
                                          const unsigned char* lq TS_PARAMS) {
    constexpr int R = G::R, NF = NTW * NCHL, NFS = NF - NRES, NFSPAD = rup(NFS, R);
    static_assert(NRES >= 0 && NRES <= NF, "bad resident fragment count");
    static_assert(NLDS == 0 || NFS - NLDS >= R, "the first R ring fragments of a layer are streamed");
    constexpr int MT = G::MT, XDW = (NTW == 1 && GS == 1 && !SIDE) ? G::XDEPTH1 : G::XDEPTH,      // (the head sweep)
                   XD = NCHL < XDW ? NCHL : XDW;
    constexpr int IN_T = 2 * NCB * 1024;                   // bytes of one row tile's operand block (NCB chunks per split part, the first NCHL of them swept)
    f16x8 X1[XD][MT], X2[XD][MT];
    auto xload = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int h = 0; h < MT; ++h) {
            X1[c % XD][h] = *reinterpret_cast<const f16x8*>(lds_in + h * IN_T + ((0 * NCB + c) * 64 + lane) * 16);
            X2[c % XD][h] = *reinterpret_cast<const f16x8*>(lds_in + h * IN_T + ((1 * NCB + c) * 64 + lane) * 16);
        }
    };
    auto prefetch = [&](auto jsc) {      // after streamed time slot js: refill its ring slot
        constexpr int js = decltype(jsc)::value;
        constexpr int jj = js + R;
        if constexpr (jj < NFS - NLDS) {
            xring_load<js % R>(ring, rsrc, wcur + jj * CADM_XDL_FRAG_BYTES, lane);
        } else if constexpr (jj < NFS) {
            const unsigned char* p = lq + (jj - (NFS - NLDS)) * CADM_XDL_FRAG_BYTES + lane * 16;
            ring.w[js % R][0] = *reinterpret_cast<const uintx4*>(p);
            ring.w[js % R][1] = *reinterpret_cast<const uintx4*>(p + 1024);
        } else if constexpr (jj >= NFSPAD) {
            if (jj - NFSPAD < nx_nf) xring_load<js % R>(ring, rsrc, wnext + (jj - NFSPAD) * CADM_XDL_FRAG_BYTES, lane);
        }
    };
    constexpr int NG = (NTW + GS - 1) / GS, NST = Epi::NSTAGE;
    constexpr int NPR = G::NPROD;
    floatx4 hi[2][GS][MT], lo[2][GS][MT], ll[2][GS][MT];   // [group parity][tile of the group][row tile]: the previous group is
    typename Epi::State pst[GS][MT];                       // finished as side work while this group accumulates (no register moves)
    // stage s of the side epilogue goes to chunks >= 1, i.e. at least one chunk of MFMAs after the accumulators were
    // last written (the compiler cannot see asm MFMA latency)
    auto stage_chunk = [](int st) constexpr { return NCHL == 1 ? 0 : NST == 1 ? 1 : 1 + st * (NCHL - 2) / (NST - 1); };
    TR_DECL
    TR(0)
    static_for(std::make_integer_sequence<int, NG>{}, [&](auto gc) {
        constexpr int g = decltype(gc)::value, gp = g & 1, pp = gp ^ 1;
        constexpr int gs = (NTW - GS * g) < GS ? (NTW - GS * g) : GS;
        constexpr int pgs = (SIDE && g > 0) ? GS : 0;            // tiles of the previous group (groups before the last are full)
        static_for(std::make_integer_sequence<int, XD - 1>{}, [&](auto cc) { xload(cc); });
#pragma unroll
        for (int k = 0; k < gs; ++k)
#pragma unroll
            for (int h = 0; h < MT; ++h) {
                hi[gp][k][h] = epi.init(GS * g + k, h); lo[gp][k][h] = epi.init_lo(GS * g + k, h); ll[gp][k][h] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int k = 0; k < gs; ++k)
#pragma unroll
            for (int h = 0; h < MT; ++h) xdl_operand_nops<false>(hi[gp][k][h], lo[gp][k][h], ll[gp][k][h]);      // VALU-zeroed accumulators -> MFMA srcC
        static_for(std::make_integer_sequence<int, NCHL>{}, [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int j0 = GS * g * NCHL + c * gs;
            if constexpr (c == 1 && g == 0) { TS(10) }
            if constexpr (c + XD - 1 < NCHL) xload(std::integral_constant<int, c + XD - 1>{});
            static_for(std::make_integer_sequence<int, NPR * gs * MT>{}, [&](auto mc) {      // hi(k,h).. lo(k,h).. lo'(k,h).. [ll(k,h)..]
                constexpr int h = decltype(mc)::value % MT, k = (decltype(mc)::value / MT) % gs, prod = decltype(mc)::value / (MT * gs);
                constexpr int j = j0 + k;
                floatx4& acc = (prod == 0 || prod == 2) ? hi[gp][k][h] : lo[gp][k][h];      // w1 x1, w1 x2 | w2 x1 [, w2 x2] (x2 unscaled: xdl_geo.h)
                const f16x8& x = prod >= 2 ? X2[c % XD][h] : X1[c % XD][h];
                constexpr int part = (prod == 1 || prod == 3) ? 1 : 0;
                if constexpr (j < NRES) xmfma_res(acc, res[j][part], x);
                else xmfma_ring<G::ASM_MFMA>(acc, ring.w[(j - NRES) % R][part], x);
            });
            static_for(std::make_integer_sequence<int, gs>{}, [&](auto kc) {
                constexpr int j = j0 + decltype(kc)::value;
                if constexpr (j >= NRES) prefetch(std::integral_constant<int, j - NRES>{});
            });
            if constexpr (pgs > 0) {
                static_for(std::make_integer_sequence<int, NST>{}, [&](auto sc) {
                    constexpr int st = decltype(sc)::value;
                    if constexpr (stage_chunk(st) == c) {
#pragma unroll
                        for (int k = 0; k < pgs; ++k)
#pragma unroll
                            for (int h = 0; h < MT; ++h)
                                epi.template stage<st>(GS * (g - 1) + k, h, hi[pp][k][h], lo[pp][k][h], ll[pp][k][h], pst[k][h]);
                    }
                });
            }
            if constexpr (g == NG - 1 && c < 8) { TR(1 + c) }
            __builtin_amdgcn_sched_barrier(0);      // pin the software pipeline: no load hoisting across chunks
        });
        if constexpr (g == NG - 1 || !SIDE) {       // the last group's epilogue has no MFMAs of this wave left to hide behind
            TS(11)
#pragma unroll
            for (int k = 0; k < gs; ++k)
#pragma unroll
                for (int h = 0; h < MT; ++h) xdl_result_nops<false>(hi[gp][k][h], lo[gp][k][h], ll[gp][k][h]);
            static_for(std::make_integer_sequence<int, NST>{}, [&](auto sc) {
#pragma unroll
                for (int k = 0; k < gs; ++k)
#pragma unroll
                    for (int h = 0; h < MT; ++h)
                        epi.template stage<decltype(sc)::value>(GS * g + k, h, hi[gp][k][h], lo[gp][k][h], ll[gp][k][h], pst[k][h]);
            });
        }
    });
    TS(12)
    TR(10)
    TR_FLUSH
    static_for(std::make_integer_sequence<int, NFSPAD - NFS>{}, [&](auto jc) {
        prefetch(std::integral_constant<int, NFS + decltype(jc)::value>{});
    });
}

template <class G, int NOISE, int NTW, bool SEQ>
__device__ __forceinline__ void xdl_run(const RolloutArgs& a, unsigned char* xsmem) {
    constexpr int D = G::D, A = G::A, P = G::P, C = G::C, K0 = G::K0, NC0 = G::NC0, NCH = G::NCH, NTO = G::NTO;
    constexpr int NP = G::NP, NPI = G::NPI, NAI = G::NAI, ENV = G::ENV, R = G::R;
    constexpr int MT = G::MT;
    constexpr int XNH = G::NHC;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = rollout_item();
    const int e = item / a.wgs_per_member;
    const int grp = item % a.wgs_per_member;
    const int H = a.H;
    const int arow = tid & 15, fg = (tid >> 4) & 15;
    // feature threads (rollout state, input assembly): MT = 1: waves 0-3, their twins in waves 4-7 make the noise;
    // MT = 2: everybody -- waves 0-3 hold row tile 0, waves 4-7 row tile 1, and make their own noise
    constexpr bool ONED = G::ONED;
    const int sd = tid >> 4;                                    // ONED: dim slot 0..31 of this thread (arow = its row)
    const bool feat = ONED ? sd < D : (MT > 1 || wave < 4);
    const bool nzt = ONED && sd >= 32 - NP;                     // ONED: noise thread of pair sd - (32 - NP)
    // ONED: the action features of x_in are not written in the state phase (where they lengthened the chain of the waves that
    // own dims 0..A-1 -- the slowest waves of the phase every other wave waits for: 1640-1780 cycles against 1250 for the waves
    // without them, profiles/r4_phase_timing_skeleton.txt) but one step AHEAD, by the last A dim slots (waves 6-7: one-tile
    // waves) in the slack behind their hidden-layer-1 sweep: x_in is free from the barrier behind layer 0 on.
    const bool actt = ONED && sd >= 32 - A;
    const int rt = MT > 1 ? (wave >> 2) : 0;                   // row tile of this thread's state
    const int ntiles = a.tile_count;                            // row tiles of this launch: [tile0, tile0 + tile_count) of the member
    float* stats = reinterpret_cast<float*>(xsmem + G::STATS);
    float* ctrl_s = reinterpret_cast<float*>(xsmem + G::CTRL) + rt * 16 * H;
    float* ofull = reinterpret_cast<float*>(xsmem + G::OFULL + rt * G::OFULL_T);
    float2* zb = reinterpret_cast<float2*>(xsmem + G::ZB + rt * G::ZB_T);
    const int xin_rt = G::XIN + rt * G::XIN_T;                 // this thread's row tile inside x_in
    const int bias_off = G::CTRL + rup(MT * 16 * a.H * 4, 16);      // LDS byte offset of the bias tiles (BIAS_LDS only)
    const bool bias_lds = G::BIAS_LDS && a.bias_lds;

    // ---- once per workgroup: stats, zero padding of the operand buffers ----
    for (int i = tid; i < P; i += G::NTHR) {
        stats[G::ST_OBS_MEAN + i] = a.obs_mean[i];
        stats[G::ST_OBS_DEN + i] = 1.0f / (a.obs_std[i] + 1e-10f);
    }
    for (int i = tid; i < A; i += G::NTHR) {
        stats[G::ST_ACT_MEAN + i] = a.act_mean[i];
        stats[G::ST_ACT_DEN + i] = 1.0f / (a.act_std[i] + 1e-10f);
    }
    for (int i = tid; i < (G::OFULL - G::XIN) / 16; i += G::NTHR) reinterpret_cast<uintx4*>(xsmem + G::XIN)[i] = uintx4{0u, 0u, 0u, 0u};
    if (bias_lds) {
        const uintx4* src = reinterpret_cast<const uintx4*>(a.xb + (size_t)e * a.xb_member);
        // one float4 per (tile, lane group): the copy of data row 0
        for (int i = tid; i < (XNH * G::NT + NTO) * 4; i += G::NTHR) reinterpret_cast<uintx4*>(xsmem + bias_off)[i] = src[(i >> 2) * 64 + (i & 3) * 16];
    }

    __syncthreads();      // (the tile prologue writes context / action features into x_in from OTHER threads than the ones that zeroed it, and reads stats)

    // byte offset (part 0) of input feature f of row 0 inside x_in; row arow adds arow * 16
    auto xin_base = [&](int f) { return ((f >> 5) * 64 + ((f & 31) >> 3) * 16) * 16 + (f & 7) * 2; };
    const int arow16 = arow * 16;
    auto xin_off = [&](int f) { return xin_base(f) + arow16; };
    float* tab = reinterpret_cast<float*>(xsmem + G::TAB);
    if constexpr (ONED) {
        if (arow == 0 && sd < D) {      // per-dim constants: head statistics, the (up to two) input features this dim feeds
            float* te = tab + sd * G::TABD;
            te[0] = a.delta_mean[sd];
            te[1] = a.delta_std[sd] + 1e-10f;
            te[2] = a.delta_std[sd];                                    // core/utils.py:360-363 through head_sd (rollout_env.h)
            te[3] = expf(-a.maxlv[sd]);
            te[4] = expf(a.minlv[sd]);
            int ff[2], fop[2];
            const int nf = dim_feats<ENV>(sd, ff, fop);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool on = i < nf;
                const int f = on ? ff[i] : 0;
                te[5 + i] = a.obs_mean[f];
                te[7 + i] = 1.0f / (a.obs_std[f] + 1e-10f);
                te[9 + i] = __builtin_bit_cast(float, on ? xin_base(f) : G::SPARE ? xin_base(K0) : -1);
                te[11 + i] = __builtin_bit_cast(float, on ? fop[i] : G::SPARE ? 3 : 0);      // (op 3: no feature, 0.0 into the spare slot)
            }
        }
    } else if (arow == 0 && wave < 4) {
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi) {
            float* te = tab + (pi * 16 + fg) * G::TABW;
            const int dp = fg + 16 * pi;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = 2 * dp + h;
                const int dc = d < D ? d : 0;
                te[0 + h] = a.delta_mean[dc];
                te[2 + h] = a.delta_std[dc] + 1e-10f;
                te[4 + h] = a.delta_std[dc];                           // core/utils.py:360-363 through head_sd (rollout_env.h)
                te[6 + h] = expf(-a.maxlv[dc]);
                te[8 + h] = expf(a.minlv[dc]);
                int ff[2], fop[2];
                const int nf = d < D ? dim_feats<ENV>(d, ff, fop) : 0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool on = i < nf;
                    const int f = on ? ff[i] : 0;
                    te[10 + 2 * h + i] = a.obs_mean[f];
                    te[14 + 2 * h + i] = 1.0f / (a.obs_std[f] + 1e-10f);
                    te[18 + 2 * h + i] = __builtin_bit_cast(float, on ? xin_base(f) : -1);
                    te[22 + 2 * h + i] = __builtin_bit_cast(float, on ? fop[i] : 0);
                }
            }
        }
    }
    const int a0 = (fg - (NP & 15) + 16) & 15;                          // first action feature of this thread
    // Non-finite inputs: the saturating f16 split below would turn an inf feature into a finite one, so the row would carry on with a
    // plausible-looking return.  Every feature is folded into the thread's reward sum first (0 * v = NaN iff v is NaN or inf,
    // else +-0): a row that ever saw a non-finite observation, action or context value returns NaN -- what the reference's
    // matmuls do to it (core/utils.py:441-472).  tests/test_gpu_precision.py pins this.
    float ret = 0.0f;
    auto put_x = [&](int off, float v) {                                // both split parts of one input feature
        ret = fmaf(0.0f, v, ret);
        // (the feature as an fp32 number, whatever expression produced it: hipcc otherwise folds a multiply into the conversion --
        //  v_fma_mixlo_f16, ONE rounding instead of two -- where the expression's shape allows, and the flavours' state phases, written
        //  differently around the same arithmetic, would then differ in the last bit of a rare f16 tie)
        asm volatile("" : "+v"(v));
        _Float16 h1, h2;                                                // (a value beyond the f16 range saturates in the split: fp16_saturate_on)
        xsplit(v, h1, h2);
        *reinterpret_cast<_Float16*>(xsmem + xin_rt + off) = h1;
        *reinterpret_cast<_Float16*>(xsmem + xin_rt + NC0 * 1024 + off) = h2;
    };

    // ---- weight stream of this (member, wave) ----
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xw, 0, a.xw_bytes, 0x00020000);
    const unsigned wbase = e * a.xw_member_b + a.xw_wave_b[wave];
    const float* xb = a.xb + (size_t)e * a.xb_member;
    const int my_ntw = G::BASE + (wave < G::EXTRA ? 1 : 0);
    const int tstart = wave * G::BASE + (wave < G::EXTRA ? wave : G::EXTRA);
    const int ht = G::NW - 1 - wave;                       // this wave's head tile (if < NTO)
    const int nhead = ht < NTO ? 1 : 0;
    const int l0_nf = my_ntw * G::NC0S, lh_nf = my_ntw * NCH, hd_nf = nhead * NCH;      // (NC0S: without layer 0's invariant chunk, xdl_geo.h)
    const unsigned w_l0 = wbase, w_h1 = w_l0 + l0_nf * CADM_XDL_FRAG_BYTES;
    const unsigned w_hd = w_h1 + (XNH - 1) * lh_nf * CADM_XDL_FRAG_BYTES;
    const unsigned w_inv = w_hd + hd_nf * CADM_XDL_FRAG_BYTES;      // the invariant chunk's fragments, one per tile (G::INV)
    const int inv_off = bias_off + (G::BIAS_LDS ? G::BIAS_BYTES : 0) + G::LQ_BYTES;      // LDS: (HI, LO) of "bias + invariant chunk" per (row tile, tile)

    // layer ids: 0 = layer 0, 1 .. NH-1 = hidden, NH = head.
    // Register-resident fragments (never re-read from L2): the whole head tile on the waves that have one and registers to
    // spare, and the first Q_l fragments of hidden layers l = 1..3 -- spread EVENLY, so that every layer streams about the
    // same number of bytes: the L2 -> CU path (~50 B/clk) is the scarce resource, and a layer that streams nothing
    // leaves it idle while its neighbours wait for it.  The STREAMED part of a layer is the tail of its stream region.
    constexpr int GSZ = SEQ ? 1 : CADM_XDL_GROUP;                       // tiles per group (xdl_geo.h: xdl_group(wave))
    constexpr int NFH = NTW * NCH;                                      // fragments of a hidden layer (this wave)
    constexpr bool RESO = NTW == G::BASE && G::res_frags(NTW) >= NCH + 3;
    constexpr int RREM = G::res_frags(NTW) - (RESO ? NCH : 0);
    constexpr int Q1 = cmin((RREM + 2) / 3, NFH), Q2 = cmin((RREM - Q1 + 1) / 2, NFH), Q3 = cmin(RREM - Q1 - Q2, NFH);
    constexpr int NRESH = Q1 + Q2 + Q3;
    auto hq = [&](int l) { return l == 1 ? Q1 : l == 2 ? Q2 : l == 3 ? Q3 : 0; };
    auto lay_res = [&](int l) { return l == 0 ? 0 : l < XNH ? hq(l) : RESO ? hd_nf : 0; };
    auto lay_off = [&](int l) {          // first streamed fragment of layer l
        return (l == 0 ? w_l0 : l < XNH ? w_h1 + (l - 1) * lh_nf * CADM_XDL_FRAG_BYTES : w_hd) + lay_res(l) * CADM_XDL_FRAG_BYTES;
    };
    auto lay_nf = [&](int l) { return (l == 0 ? l0_nf : l < XNH ? lh_nf : hd_nf) - lay_res(l); };
    auto next_streamed = [&](int l) {    // next layer (cyclically over steps) with a streamed part
        for (int k = 0; k <= XNH; ++k) {
            l = l == XNH ? 0 : l + 1;
            if (lay_nf(l) > 0) break;
        }
        return l;                         // (nothing streamed at all: a layer with lay_nf = 0, no loads are issued for it)
    };
    // LDS-resident tail of hidden layers 1..3 (XC::LQ_SLOTS): as many as leave the first R ring fragments of every layer streamed
    constexpr int LQ = (NTW >= 2 && G::LQ_SLOTS > 0) ? cmax(0, cmin(G::LQ_SLOTS, NFH - cmax(Q1, cmax(Q2, Q3)) - R)) : 0;
    const int lq_off = bias_off + (G::BIAS_LDS ? G::BIAS_BYTES : 0) + wave * 3 * G::LQ_SLOTS * CADM_XDL_FRAG_BYTES;
    if constexpr (LQ > 0) {      // once per workgroup: L2 -> LDS (a wave-private region: no barrier)
#pragma unroll
        for (int l = 1; l < (XNH < 4 ? XNH : 4); ++l)
#pragma unroll
            for (int i = 0; i < LQ; ++i)
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const uintx4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + part * 1024,
                                                                           w_h1 + ((l - 1) * lh_nf + NFH - LQ + i) * CADM_XDL_FRAG_BYTES, 0);
                    *reinterpret_cast<uintx4*>(xsmem + lq_off + ((l - 1) * LQ + i) * CADM_XDL_FRAG_BYTES + part * 1024 + lane * 16) = v;
                }
    }
    uintx4 resH[NRESH > 0 ? NRESH : 1][2], resO[RESO ? NCH : 1][2];
    static_for(std::make_integer_sequence<int, NRESH>{}, [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int l = q < Q1 ? 1 : q < Q1 + Q2 ? 2 : 3, ql = q - (l == 1 ? 0 : l == 2 ? Q1 : Q1 + Q2);
        // (a layer the model does not have gets no load: its registers would be dead behind an asynchronous asm load, hipcc reuses dead
        //  AGPRs as spill space, and a spill written before the load lands would be overwritten -- found by the JIT's ISA scan on a
        //  two-layer net, cadm_amd/isa_check.py)
        if constexpr (l < XNH) {
            const unsigned so = w_h1 + ((l - 1) * lh_nf + ql) * CADM_XDL_FRAG_BYTES;
            xres_load(resH[q][0], rsrc, lane * 16, so);
            xres_load(resH[q][1], rsrc, lane * 16 + 1024, so);
        }
    });
    if constexpr (RESO) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            // (a wave without a head tile loads in-bounds garbage that is never used)
            const unsigned so = q < hd_nf ? w_hd + q * CADM_XDL_FRAG_BYTES : wbase;
            xres_load(resO[q][0], rsrc, lane * 16, so);
            xres_load(resO[q][1], rsrc, lane * 16 + 1024, so);
        }
    }
    if constexpr (NRESH > 0 || RESO) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XRing<G> ring;
    {
        const int first = next_streamed(XNH);
        const unsigned fo = lay_off(first);
        const int fn = lay_nf(first);
        static_for(std::make_integer_sequence<int, R>{}, [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (s < fn) xring_load<s>(ring, rsrc, fo + s * CADM_XDL_FRAG_BYTES, lane);
        });
    }

    for (int tile = grp; tile < (ntiles + MT - 1) / MT; tile += a.wgs_per_member) {      // groups of MT row tiles
        // ---- this thread's row ----
        int re = (a.tile0 + tile * MT + rt) * 16 + arow;
        const bool valid = re < a.rows_per_member;
        if (!valid) re = a.rows_per_member - 1;
        const int cidx = re / a.PE, jl = re % a.PE;
        const int mi = cidx / a.n_local, nl = cidx % a.n_local;
        const int j = e * a.PE + jl;
        const int lr = (mi * a.n_local + nl) * a.p + j;                                  // local row (returns / eps / traj)
        const unsigned grow = (unsigned)((mi * a.n_global + a.cand_offset + nl) * a.p + j);   // global row (RNG counter)
        const int abase = ((mi * a.n_global + a.cand_offset + nl) * H) * A;
        const int ep = j % a.E;
        int ctx_off;
        if (!a.quirks) ctx_off = ((j / a.PE) * a.m + mi) * C;            // own member's context
        else if (a.it & 1) ctx_off = (mi * a.E + ep) * C;                // Q2: [E,m] memory reread as [m,E]
        else ctx_off = (ep * a.m + mi) * C;                              // Q1: encoder j % E

        float po[NPI][2], areg[NAI];
        // ONED action threads: write step tt's (normalised, split) action feature from areg, then fetch step tt + 1's.
        // (everything is recomputed from an opaque copy of the dim slot: nothing of it is hoisted out of the step loop to sit in
        //  registers through the sweeps, where there are none to spare)
        auto act_put = [&](int tt) {
            int sda = tid >> 4;
            asm volatile("" : "+v"(sda));
            const int ai = sda - (32 - A);
            float v = areg[0];
            if (a.norm_actions) v = (v - stats[G::ST_ACT_MEAN + ai]) * stats[G::ST_ACT_DEN + ai];   // :443
            put_x(xin_off(P + ai), v);
            if (tt + 1 < H) areg[0] = a.actions[abase + (tt + 1) * A + ai];
        };
        if constexpr (ONED) {
            po[0][0] = feat ? (a.obs_rows ? a.obs_rows[(size_t)lr * D + sd] : a.obs[mi * D + sd]) : 0.0f;      // :432
            po[0][1] = 0.0f;
            areg[0] = actt ? a.actions[abase + sd - (32 - A)] : 0.0f;
            if constexpr (C > 0) {
                for (int f = P + A + sd; f < K0; f += 32) put_x(xin_off(f), a.ctx_vec[ctx_off + f - P - A]);      // static: context (:433-439)
            }
            if (actt) act_put(0);      // step 0's action features; areg then holds step 1's
            for (int t = sd; t < H; t += 32) ctrl_s[arow * H + t] = ctrl_term<ENV>(a.actions + abase + t * A, A);
        } else {
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = 2 * (fg + 16 * pi) + h;
                const int dc = d < D ? d : 0;
                po[pi][h] = !feat ? 0.0f : a.obs_rows ? a.obs_rows[(size_t)lr * D + dc] : a.obs[mi * D + dc];   // :432
            }
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai) {
            const int ac = a0 + 16 * ai;
            areg[ai] = (feat && ac < A) ? a.actions[abase + ac] : 0.0f;
        }
        if (feat) {
            if constexpr (C > 0) {
                for (int f = P + A + fg; f < K0; f += 16) put_x(xin_off(f), a.ctx_vec[ctx_off + f - P - A]);   // static: context (:433-439)
            }
            for (int t = fg; t < H; t += 16) ctrl_s[arow * H + t] = ctrl_term<ENV>(a.actions + abase + t * A, A);
        }
        }
        auto gen_noise = [&](int t) {
#pragma unroll
            for (int pi = 0; pi < NPI; ++pi) {
                const int dp = ONED ? sd - (32 - NP) : fg + 16 * pi;      // (ONED: called by the noise threads only)
                if (dp >= NP) continue;                   // (wave-uniform for whole waves of unused pair slots)
                float2 z;
                if constexpr (NOISE == CADM_NOISE_INJECT) {
                    const float* epp = a.eps + ((size_t)t * a.m * a.n_local * a.p + lr) * D + 2 * dp;
                    z.x = epp[0];
                    z.y = (2 * dp + 1 < D) ? epp[1] : 0.0f;
                } else {
                    uint32_t pc[4] = {grow, (uint32_t)t, eps_group(dp), CADM_STREAM_EPS | ((uint32_t)a.it << 8)};
                    uint32_t pk[2] = {a.seed, a.call};
                    philox_rounds<0, 10>(pc, pk);
                    const bool hi = eps_sub(dp) != 0;      // (the call's other two words are pair dp +- 4's: rollout_env.h)
                    box_muller(u01(hi ? pc[2] : pc[0]), u01(hi ? pc[3] : pc[1]), z.x, z.y);
                }
                zb[((t & 1) * NPI + pi) * 256 + (dp & 15) * 16 + arow] = z;
            }
        };
        __syncthreads();
        if constexpr (G::INV) {
            // Layer 0's invariant last chunk (xdl_geo.h): context features only, the same in every step.  Every flavour accumulates it FIRST;
            // this kernel does so here, once per row tile -- bias -> HI += w1 x1, LO += w2 x1, HI += w1 x2, the instructions a sweep would issue --
            // and leaves (HI, LO) per tile in LDS: the step loop's layer-0 accumulators start from them.  A wave reads back only its own tiles.
            floatx4 ll0 = floatx4{0.f, 0.f, 0.f, 0.f};
            const XHiddenEpi<G> epib{xsmem, xb, bias_off, 0, 0, tstart, lane};
            static_for(std::make_integer_sequence<int, NTW>{}, [&](auto tc) {
                constexpr int ti = decltype(tc)::value;
                const uintx4 w0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, w_inv + ti * CADM_XDL_FRAG_BYTES, 0);
                const uintx4 w1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + 1024, w_inv + ti * CADM_XDL_FRAG_BYTES, 0);
#pragma unroll
                for (int h = 0; h < MT; ++h) {
                    const unsigned char* xin = xsmem + G::XIN + h * G::XIN_T;
                    const f16x8 x1 = *reinterpret_cast<const f16x8*>(xin + ((0 * NC0 + NC0 - 1) * 64 + lane) * 16);
                    const f16x8 x2 = *reinterpret_cast<const f16x8*>(xin + ((1 * NC0 + NC0 - 1) * 64 + lane) * 16);
                    floatx4 hi = epib.init(ti, h), lo = floatx4{0.f, 0.f, 0.f, 0.f};
                    xdl_operand_nops<false>(hi, lo, ll0);
                    xmfma_ring<G::ASM_MFMA>(hi, w0, x1);
                    xmfma_ring<G::ASM_MFMA>(lo, w1, x1);
                    xmfma_ring<G::ASM_MFMA>(hi, w0, x2);
                    xdl_result_nops<false>(hi, lo, ll0);
                    unsigned char* dst = xsmem + inv_off + ((h * G::NT + tstart + ti) * 2) * 1024 + lane * 16;
                    *reinterpret_cast<floatx4*>(dst) = hi;
                    *reinterpret_cast<floatx4*>(dst + 1024) = lo;
                }
            });
        }
        TS_DECL

        for (int t = 0; t <= H; ++t) {
            // ===== state update from step t-1's head (:348-365,463-466) + reward (:469-471) + input assembly (:442-460) =====
            if constexpr (ONED) {
                // one observation dim per thread: thread (arow, sd) owns dim sd of its row; its pair partner is 16 lanes up
                const int dp = sd >> 1, hh = sd & 1;
                if (feat) {
                    floatx4 tq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) tq[q] = *reinterpret_cast<const floatx4*>(tab + sd * G::TABD + 4 * q);
                    auto tv = [&](int w) { return tq[w >> 2][w & 3]; };
                    auto ti_ = [&](int w) { const float fv = tq[w >> 2][w & 3]; return __float_as_int(fv); };
                    if (t > 0) {
                        const int jt = dp >> 2, lt = (dp & 3) * 16 + arow;
                        const float* vp = ofull + (jt * 64 + lt) * 4;                            // (mu0, mu1, lv0, lv1) of the pair
                        float delta = vp[hh] * tv(1) + tv(0);                                    // denormalize, :349
                        if constexpr (NOISE != CADM_NOISE_NONE) {
                            const float z = reinterpret_cast<const float*>(zb + ((t - 1) & 1) * 256 + dp * 16 + arow)[hh];
                            const float sdv = head_sd(vp[2 + hh], tv(3), tv(4), tv(2));          // :356-363
                            delta = delta + z * sdv;                                             // :365
                        }
                        po[0][0] = postproc<ENV>(sd, po[0][0], delta);                           // :466
                        if (a.traj && valid) a.traj[((size_t)(t - 1) * a.m * a.n_local * a.p + lr) * D + sd] = po[0][0];
                    }
                    if (t < H) {
                        float sn = 0.0f, cs = 0.0f;
                        if constexpr (ENV == CADM_ENV_HALFCHEETAH) {                             // the one trig pair (obs dim 2)
                            if (ti_(11) == 1) sincos_cw(po[0][0], &sn, &cs);
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int off = ti_(9 + i), op = ti_(11 + i);
                            if constexpr (G::SPARE) {      // unconditional: selects, no exec-mask regions (XC::SPARE)
                                float pv = po[0][0];
                                pv = op == 1 ? sn : pv;
                                pv = op == 2 ? cs : pv;
                                const float xv = (pv - tv(5 + i)) * tv(7 + i);                   // :450-451
                                put_x(off + arow16, op == 3 ? 0.0f : xv);
                            } else if (off >= 0) {
                                const float pv = op == 1 ? sn : op == 2 ? cs : po[0][0];
                                put_x(off + arow16, (pv - tv(5 + i)) * tv(7 + i));               // :450-451
                            }
                        }
                    }
                }
                // reward of the pair (dims 2dp, 2dp+1): the even dim's thread adds it, reading its partner's dim 16 lanes up
                // (every lane of the wave takes part in the exchange; an env whose reward reads one dim compiles it away)
                const float o1 = __shfl_down(po[0][0], 16);
                if (feat && hh == 0) {
                    if constexpr (ENV == CADM_ENV_CARTPOLE) {
                        if (t > 0) ret += reward_part<ENV>(dp, po[0][0], o1, 0.0f);              // reads NEXT obs
                    } else {
                        if (t < H) ret += reward_part<ENV>(dp, po[0][0], o1, ctrl_s[arow * H + t]);
                    }
                }
            } else if (feat) {
#pragma unroll
            for (int pi = 0; pi < NPI; ++pi) {
                const int dp = fg + 16 * pi;
                if (dp < NP) {
                    floatx4 tq[7];
#pragma unroll
                    for (int q = 0; q < 7; ++q) tq[q] = *reinterpret_cast<const floatx4*>(tab + (pi * 16 + fg) * G::TABW + 4 * q);
                    auto tv = [&](int w) { return tq[w >> 2][w & 3]; };
                    // (bit_cast of an ext-vector ELEMENT expression is miscompiled by hipcc 7.2: go through a scalar)
                    auto ti_ = [&](int w) { const float fv = tq[w >> 2][w & 3]; return __float_as_int(fv); };
                    if (t > 0) {
                        const int jt = dp >> 2, lt = (dp & 3) * 16 + arow;
                        const floatx4 v = *reinterpret_cast<const floatx4*>(ofull + (jt * 64 + lt) * 4);   // (mu0, mu1, lv0, lv1)
                        float2 z = make_float2(0.0f, 0.0f);
                        if constexpr (NOISE != CADM_NOISE_NONE) z = zb[(((t - 1) & 1) * NPI + pi) * 256 + fg * 16 + arow];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float delta = v[h] * tv(2 + h) + tv(0 + h);                          // denormalize, :349
                            if constexpr (NOISE != CADM_NOISE_NONE) {
                                const float sd = head_sd(v[2 + h], tv(6 + h), tv(8 + h), tv(4 + h));   // :356-363
                                delta = delta + (h ? z.y : z.x) * sd;                               // :365
                            }
                            po[pi][h] = postproc<ENV>(2 * dp + h, po[pi][h], delta);                // :466
                        }
                        if (a.traj && valid) {
                            float* tp = a.traj + ((size_t)(t - 1) * a.m * a.n_local * a.p + lr) * D + 2 * dp;
                            tp[0] = po[pi][0];
                            if (2 * dp + 1 < D) tp[1] = po[pi][1];
                        }
                    }
                    if constexpr (ENV == CADM_ENV_CARTPOLE) {
                        if (t > 0) ret += reward_part<ENV>(dp, po[pi][0], po[pi][1], 0.0f);   // reads NEXT obs
                    } else {
                        if (t < H) ret += reward_part<ENV>(dp, po[pi][0], po[pi][1], ctrl_s[arow * H + t]);
                    }
                    if (t < H) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float sn = 0.0f, cs = 0.0f;
                            if constexpr (ENV == CADM_ENV_HALFCHEETAH) {                 // the one trig pair (obs dim 2)
                                if (ti_(22 + 2 * h) == 1) sincos_cw(po[pi][h], &sn, &cs);
                            }
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int off = ti_(18 + 2 * h + i), op = ti_(22 + 2 * h + i);
                                // (guarded writes here: the unconditional form of the one-dim layout above and of the wave-tile kernel made ONE
                                //  instantiation of this layout -- halfcheetah without context, two row tiles, injected noise -- return wrong rows for
                                //  its second row tile and fault, with or without the extra writes; found by tools/fuzz_rollout.py, not understood)
                                if (off >= 0) {
                                    const float pv = op == 1 ? sn : op == 2 ? cs : po[pi][h];
                                    put_x(off + arow16, (pv - tv(10 + 2 * h + i)) * tv(14 + 2 * h + i));   // :450-451
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
                        float v = areg[ai];
                        if (a.norm_actions) v = (v - stats[G::ST_ACT_MEAN + ac]) * stats[G::ST_ACT_DEN + ac];   // :443
                        put_x(xin_off(P + ac), v);
                        if (t + 1 < H) areg[ai] = a.actions[abase + (t + 1) * A + ac];
                    }
                }
            }
            }
            if (t == H) break;
            // Gaussian-head noise of THIS step (consumed by the next state phase, through the double-buffered zb).
            // One row tile: made here by the twin thread in waves 4-7, which have nothing else to do while waves 0-3 update
            // the state.  Two row tiles: every thread owns state and makes its own noise, but not here, where it would
            // lengthen the state phase that everything waits for -- see the hidden layers / the head below.
#ifndef CADM_XDL_NOISE_IN_HIDDEN
#define CADM_XDL_NOISE_IN_HIDDEN 0
#endif
            if constexpr (NOISE != CADM_NOISE_NONE && MT == 1 && !CADM_XDL_NOISE_IN_HIDDEN) {
                if (ONED ? nzt : !feat) gen_noise(t);
            }
            TS(0)
            __syncthreads();
            TS(1)

            // ================= dense layers =================
            {
                int act_out = G::ACTA, act_in = G::XIN;      // LDS byte offsets (not pointers: keeps every access a ds_ op)
                // hidden epilogue: bias, swish, f16 split, store as the next layer's B operand
                auto hidden_epi = [&](int layer, int out) { return XHiddenEpi<G>{xsmem, xb, bias_off, layer, out, tstart, lane}; };
                // layer 0
                {
                    const int nx = next_streamed(0);
                    XHiddenEpi<G> epi0 = hidden_epi(0, act_out);
                    if constexpr (G::INV) epi0.inv = xsmem + inv_off;      // accumulators start from "bias + invariant chunk" (made once per row tile, below the tile prologue)
                    xdl_sweep<G, NTW, G::NC0S, 0, GSZ, !SEQ, 0, NC0>(ring, nullptr, rsrc, w_l0, lay_off(nx), lay_nf(nx), xsmem + act_in, lane,
                                              epi0, nullptr TS_ARGS);
                }
                TS(2)
                XDL_LAYER_SYNC();
                TS(3)
                if constexpr (ONED && XNH == 1) {      // one hidden layer: no hidden-layer-1 sweep to carry the next step's action features (see actt)
                    if (actt && t + 1 < H) act_put(t + 1);
                }
                // hidden layers 1 .. NH-1: the first three are unrolled (distinct resident registers), the rest loop
                auto hidden = [&](int l, auto res_c, auto base_c, auto lq_c) {
                    constexpr int NRES = decltype(res_c)::value, RBASE = decltype(base_c)::value;
                    act_in = act_out;
                    act_out = (act_in == G::ACTA) ? G::ACTB : G::ACTA;
                    const int nx = next_streamed(l);
                    // (layers 1..3: distinct instantiations with their LDS-resident tail; deeper layers stream everything)
                    constexpr int NL = decltype(lq_c)::value ? LQ : 0;
                    TR_ON(t == 10 && l == 2, wave)      // (CADM_PHASE_TIMING builds: raw stamps of this one sweep, tools/sweep_trace.py)
                    xdl_sweep<G, NTW, NCH, NRES, GSZ, !SEQ, NL>(ring, resH + RBASE, rsrc, lay_off(l), lay_off(nx), lay_nf(nx), xsmem + act_in, lane,
                                                 hidden_epi(l, act_out), xsmem + lq_off + (l - 1) * LQ * CADM_XDL_FRAG_BYTES TS_ARGS);
                    // two row tiles: the waves that have a head tile (and one hidden tile less than the others: they would
                    // wait at this barrier anyway) make their noise now
                    if constexpr (NOISE != CADM_NOISE_NONE && MT > 1) {
                        if (l == 1 && nhead) gen_noise(t);
                    }
                    if constexpr (NOISE != CADM_NOISE_NONE && MT == 1 && CADM_XDL_NOISE_IN_HIDDEN) {      // (experiment: noise in the one-tile waves' slack)
                        if (l == CADM_XDL_NOISE_IN_HIDDEN && (ONED ? nzt : !feat)) gen_noise(t);
                    }
                    if constexpr (ONED) {      // next step's action features (see actt)
                        if (l == 1 && actt && t + 1 < H) act_put(t + 1);
                    }
                    if (l == 1) { TS(4) } else if (l == 2) { TS(8) } else { TS(9) }
                    XDL_LAYER_SYNC();
                    TR_LATE(11)
                    TR_ON(false, 0)
                    TS(5)
                };
                using IC0 = std::integral_constant<int, 0>;
                using IC1 = std::integral_constant<int, 1>;
                if (1 < XNH) hidden(1, std::integral_constant<int, Q1>{}, IC0{}, IC1{});
                if (2 < XNH) hidden(2, std::integral_constant<int, Q2>{}, std::integral_constant<int, Q1>{}, IC1{});
                if (3 < XNH) hidden(3, std::integral_constant<int, Q3>{}, std::integral_constant<int, Q1 + Q2>{}, IC1{});
                for (int l = 4; l < XNH; ++l) hidden(l, IC0{}, IC0{}, IC0{});
                act_in = act_out;
                // two row tiles: the waves without a head tile make their noise while the others run the head
                if constexpr (NOISE != CADM_NOISE_NONE && MT > 1) {
                    if (!nhead || XNH < 2) gen_noise(t);
                }
                // ================= output head tile (mu | logvar of 8 dims) =================
                if (nhead) {
                    const int nx = next_streamed(XNH);
                    xdl_sweep<G, 1, NCH, RESO ? NCH : 0, 1, false, 0>(ring, resO, rsrc, lay_off(XNH), lay_off(nx), lay_nf(nx), xsmem + act_in, lane,
                                                         XHeadEpi<G>{xsmem, xb, bias_off, XNH * G::NT + ht, ht, lane}, nullptr TS_ARGS);
                }
            }
            TS(6)
            __syncthreads();
            TS(7)
        }
        TS_DUMP

        // ---- a row's return = sum of its threads' reward parts, in fixed slot order ----
        __syncthreads();
        if constexpr (ONED) {
            // 32 dim slots per row; pair slot i = (2i, 2i+1): the same 16 partial sums, added in the same order, as the pair layout
            float* ret_s = reinterpret_cast<float*>(xsmem + G::ACTA);      // (activation buffer, free between tiles)
            ret_s[arow * 32 + sd] = ret;
            ret = 0.0f;
            __syncthreads();
            if (sd == 0 && valid) {
                float r = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) r += ret_s[arow * 32 + 2 * i] + ret_s[arow * 32 + 2 * i + 1];
                a.returns_rows[lr] = r;
            }
        } else {
            float* ret_s = ofull;                      // (this row tile's head buffer, free between tiles)
            if (feat) ret_s[arow * 16 + fg] = ret;
            ret = 0.0f;
            __syncthreads();
            if (feat && fg == 0 && valid) {
                float r = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) r += ret_s[arow * 16 + i];
                a.returns_rows[lr] = r;
            }
        }
        __syncthreads();
    }
}

template <class G, int NOISE>
__global__ __launch_bounds__(G::NTHR) void rollout_xdl_kernel(const RolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsmem_raw[];
    fp16_saturate_on();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves [0, EXTRA) own one hidden tile more than the others: two specialisations of the whole body, chosen per wave
    // (a scalar branch; every wave executes the same number of barriers)
    // and waves 4-7 go tile by tile (xdl_geo.h: xdl_group)
    constexpr int HALF = G::NW / 2;
    const bool seq = CADM_XDL_SEQ && wave >= HALF;
    // Static issue priority for the second-dispatched wave of every SIMD (waves 4-7): VALU issue between the two waves of a SIMD is
    // arbitrated by priority, then AGE, so the younger wave loses every contested slot and is the last to reach each layer's barrier
    // (profiles/r3_c_phase_timing_cfg3.txt: wave 4).  Measured, same box, interleaved (profiles/r4_s3_experiments.md): two row tiles
    // -1.5 % (1343 -> 1322 us at cfg3), one row tile +-0.3 % (left alone there); priority for the OLDER half instead: +0.5 % / -0.2 %.
#ifndef CADM_XDL_PRIO
#define CADM_XDL_PRIO (G::MT > 1 ? 1 : 0)
#endif
    if constexpr ((CADM_XDL_PRIO) != 0) {
        if ((CADM_XDL_PRIO) > 0 ? wave >= HALF : wave < HALF) __builtin_amdgcn_s_setprio((CADM_XDL_PRIO) > 0 ? (CADM_XDL_PRIO) : -(CADM_XDL_PRIO));
    }
    if (wave < G::EXTRA) {
        if constexpr (G::EXTRA > 0) {
            if (!seq) xdl_run<G, NOISE, G::BASE + 1, false>(a, xsmem_raw);
            else if constexpr (CADM_XDL_SEQ && G::EXTRA > HALF) xdl_run<G, NOISE, G::BASE + 1, true>(a, xsmem_raw);
        }
    } else {
        if (seq) { if constexpr (CADM_XDL_SEQ) xdl_run<G, NOISE, G::BASE, true>(a, xsmem_raw); }
        else if constexpr (!CADM_XDL_SEQ || G::EXTRA < HALF) xdl_run<G, NOISE, G::BASE, false>(a, xsmem_raw);
    }
}

template <class G, int NOISE>
int xdl_launch_noise(cadm_ctx* ctx, const RolloutArgs& a, int rows_per_member, hipStream_t s) {
    RolloutArgs args = a;
    // this launch covers row tiles [a.tile0, a.tile0 + a.tile_count) of every member, in groups of MT;
    // one workgroup of 8 waves per CU (256 registers per wave); workgroups walk over their member's groups
    const int tiles = (a.tile_count + G::MT - 1) / G::MT;
    int per_member = ctx->n_cus / ctx->E;
    if (per_member < 1) per_member = 1;
    args.wgs_per_member = tiles < per_member ? tiles : per_member;
    args.rows_per_member = rows_per_member;
    const size_t lds = G::lds_bytes(a.H);
    args.bias_lds = G::BIAS_LDS;
    if (lds > 160 * 1024) {
        cadm_set_error("rollout: horizon %d with %d hidden layers of %d units needs %zu B of LDS (> 160 KiB)", a.H, G::NHC, G::HID, lds);
        return CADM_EINVAL;
    }
    const void* fn = reinterpret_cast<const void*>(&rollout_xdl_kernel<G, NOISE>);
    if (!ctx->attr_done.count(fn)) {
        CADM_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_done.insert(fn);
    }
    if (a.dry_run) return CADM_OK;            // cadm_rollout_check: geometry / LDS validation only
    hipLaunchKernelGGL((rollout_xdl_kernel<G, NOISE>), dim3(args.wgs_per_member * ctx->E), dim3(G::NTHR), lds, s, args);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

// NOISE < 0: all three noise modes of the geometry (the library's compiled-in set); else only that one (a JIT module is built
// per noise mode, in parallel / on first use: rollout_jit.hip)
template <class G, int NOISE>
int xdl_launch_mt(cadm_ctx* ctx, const RolloutArgs& a, int rows_per_member, hipStream_t s) {
    const int mode = a.deterministic ? CADM_NOISE_NONE : a.eps ? CADM_NOISE_INJECT : CADM_NOISE_PHILOX;
    if constexpr (NOISE >= 0) {
        if (mode != NOISE) { cadm_set_error("rollout: this module holds noise mode %d, the launch needs %d", NOISE, mode); return CADM_EINVAL; }
        return xdl_launch_noise<G, NOISE>(ctx, a, rows_per_member, s);
    } else {
        if (mode == CADM_NOISE_NONE) return xdl_launch_noise<G, CADM_NOISE_NONE>(ctx, a, rows_per_member, s);
        if (mode == CADM_NOISE_INJECT) return xdl_launch_noise<G, CADM_NOISE_INJECT>(ctx, a, rows_per_member, s);
        return xdl_launch_noise<G, CADM_NOISE_PHILOX>(ctx, a, rows_per_member, s);
    }
}

}  // namespace

#include "rollout_wt.h"

namespace {

// A member's row tiles are cut between the kernel flavours, launch after launch (rows are independent and the flavours agree bit
// for bit, so the cut changes no result):
//   cooperative, one row tile per workgroup  (cap = one tile per CU of the member's share: cfg2)         cost 1
//   cooperative, two row tiles per workgroup (cap 2x)                                                    cost xdl_costs(ENV, HID).c[1]
//   wave-tile, 4 tiles per workgroup (one wave per SIMD; cap 4x)                                         cost .c[2]
//   wave-tile, 8 tiles per workgroup (cap 8x, any number of rounds)                                      cost .c[3] per round
// (xdl_geo.h: xdl_plan_units -- costs in units of the one-tile launch, per instantiation; a small dynamic programme over units of one CU share.)
template <int ENV, int C, int HID, int NH, int ACT, int NOISE = -1>
int xdl_launch(cadm_ctx* ctx, const RolloutArgs& a0, int rows_per_member, hipStream_t s) {
    using G1 = XC<ENV, C, HID, 1, NH, ACT>;
    using G2 = XC<ENV, C, HID, 2, NH, ACT>;
    using W = WT<G1>;
    RolloutArgs a = a0;
    const int tiles = (rows_per_member + 15) / 16;
    int per_member = ctx->n_cus / ctx->E;
    if (per_member < 1) per_member = 1;
    // (wide layers / long horizons: two tiles' activation buffers / the wave-tile kernel's ring do not fit the 160 KiB of LDS)
    const bool mt2_ok = G2::lds_bytes(a0.H) <= 160 * 1024;
    const bool wt_ok = W::AVAILABLE && W::lds_bytes(a0.H) <= 160 * 1024;
    a.tile0 = 0;
    a.tile_count = tiles;
    if (ctx->dev_force_mt) {            // developer library only (dev/dev_api.hip): ONE launch of the forced flavour
        const int f = ctx->dev_force_mt;
        if (f >= 3) {
            if constexpr (W::AVAILABLE) { if (wt_ok) return wt_launch<G1, NOISE>(ctx, a, rows_per_member, f == 3 ? 8 : 4, s); }
            cadm_set_error("rollout: the wave-tile kernel does not exist for this geometry (hidden %d, horizon %d)", HID, a0.H);
            return CADM_EINVAL;
        }
        if (f == 2 && mt2_ok) return xdl_launch_mt<G2, NOISE>(ctx, a, rows_per_member, s);
        return xdl_launch_mt<G1, NOISE>(ctx, a, rows_per_member, s);
    }
    if (a0.dry_run) {                   // cadm_rollout_check: every flavour the launcher may pick must exist and fit
        int rc = xdl_launch_mt<G1, NOISE>(ctx, a, rows_per_member, s);
        if (!rc && mt2_ok) rc = xdl_launch_mt<G2, NOISE>(ctx, a, rows_per_member, s);
        if constexpr (W::AVAILABLE) { if (!rc && wt_ok) rc = wt_launch<G1, NOISE>(ctx, a, rows_per_member, 8, s); }
        return rc;
    }
    // cheapest cover of the member's tiles, in units of per_member tiles (xdl_geo.h: xdl_plan_units)
    const int units = (tiles + per_member - 1) / per_member;
    const int capu[4] = {1, 2, 4, 8};
    int count[4];
    xdl_plan_units(units, mt2_ok, wt_ok, count, xdl_costs(ENV, HID));
    int rem = tiles, t0 = 0;
    for (int o = 3; o >= 0 && rem > 0; --o) {      // biggest flavour first: the last launch takes the ragged rest
        if (!count[o]) continue;
        const int cover = rem < count[o] * capu[o] * per_member ? rem : count[o] * capu[o] * per_member;
        a.tile0 = t0;
        a.tile_count = cover;
        int rc = CADM_EINVAL;
        if (o == 0) rc = xdl_launch_mt<G1, NOISE>(ctx, a, rows_per_member, s);
        else if (o == 1) rc = xdl_launch_mt<G2, NOISE>(ctx, a, rows_per_member, s);
        else {
            if constexpr (W::AVAILABLE) rc = wt_launch<G1, NOISE>(ctx, a, rows_per_member, o == 2 ? 4 : 8, s);
        }
        if (rc) return rc;
        t0 += cover;
        rem -= cover;
    }
    return CADM_OK;
}

}  // namespace
