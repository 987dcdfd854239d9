// Geometry of the split-f16 ("xdl") rollout weight stream, shared by the packer (host + pack kernel) and the
// rollout kernel so that the two can never disagree.  See DESIGN.md "rollout kernel".
//
//   * A dense layer is evaluated transposed (OUT^T = W^T * IN^T) with v_mfma_f32_16x16x32_f16: the weights are
//     the A operand (16 output units x 32 input features), 16 data rows are the B / D columns.
//   * fp32 accuracy on the f16 matrix pipe: a WEIGHT is split as  w = w1 + 2^-11 * w2,  w1 = f16_rn(w), w2 = f16_rn((w - w1) * 2^11)
//     (the 2^11 keeps the low part a NORMAL f16 number); an ACTIVATION as  x = x1 + x2,  x1 = f16_rn(x), x2 = f16_rn(x - x1), unscaled
//     (round 4: the scaling cost the epilogue 2 of its 28 instructions per tile, and the kernels are bound by their VALU instructions as
//     much as by their MFMAs; a low part below the f16 normal range is a subnormal: 3e-8 ABSOLUTE error on an activation below 0.25).
//     A product  w * x  is evaluated as  w1*x1 + w1*x2  (accumulator HI)  +  2^-11 * w2*x1  (accumulator LO); the dropped term
//     w2*x2 is <= 2^-22 |w x| (layers wider than 256 add it to LO).  3 MFMAs at 16x the fp32-MFMA rate instead of 8.
//   * swish layers (the reference's default): log2(e) is folded into the packed weights -- layer 0's W and every hidden bias are
//     multiplied by it, the head's W divided -- so a hidden pre-activation arrives as  p' = log2(e) * p,  sigmoid(p) = 1 / (1 + exp2(-p'))
//     needs no multiply, and the activation handed on is  p' * sigmoid(p) = log2(e) * swish(p)  (2 more instructions per tile saved).
//   * An output tile is 16 units; the K dimension of the layers that consume a hidden layer is cut in chunks of
//     32 = one PAIR of producer tiles, so a lane's D fragments (units 4g..4g+3 of tiles 2c, 2c+1) ARE its B fragment
//     for chunk c: activations cross layers through LDS without any cross-lane movement.
//   * Hidden tiles are distributed over the CADM_XDL_WAVES waves of a workgroup contiguously (wave w: BASE + (w < EXTRA)
//     tiles); head tiles (8 obs dims: mu0 mu1 lv0 lv1 per lane) go to the waves with the fewest hidden tiles first.
//   * A "fragment" = one (tile, chunk) of weights = 2 split parts x 64 lanes x 16 B = 2 KB.  Per (member, wave) the
//     stream stores the fragments in EXACTLY the order that wave consumes them during one rollout step:
//     [layer 0][hidden 1 .. NH-1][head]; inside a layer tiles go in groups of xdl_group(wave), chunk-major inside a group
//     (xdl_frag_index), so the kernel addresses the stream linearly.
#pragma once

#define CADM_XDL_FRAG_BYTES 2048
#define CADM_XDL_LOG2E 1.4426950408889634f
// Tiles a wave accumulates at a time.  Waves 0-3 (the first wave of each SIMD) take their tiles two at a time and run
// the epilogues at the end; waves 4-7 go tile by tile, each tile's epilogue right behind its MFMAs.  The two waves of a
// SIMD are thereby out of phase: one wave's epilogue (VALU) meets the other's MFMAs instead of its epilogue
// (a 16-cycle f16 MFMA hides only ~1 VALU op of its own wave, profiles/r2_issue_microbench.md).
#ifndef CADM_XDL_GROUP
#define CADM_XDL_GROUP 2
#endif
#ifndef CADM_XDL_SEQ
#define CADM_XDL_SEQ 0                // measured: 211 vs 203 us per launch at cfg2 -- off
#endif
#define CADM_XDL_WAVES 8          // two waves per SIMD: one wave's waits (LDS, L2, epilogue chains) hide behind the other's MFMAs

// Narrow nets: the 8-wave tile split needs at least 8 hidden tiles (113 units).  A narrower net (the reference accepts any
// --hidden_size, e.g. 64) runs on the 128-wide instantiation with zero-padded units: a padded unit has zero weights and bias on
// both sides, so whatever the nonlinearity makes of its zero pre-activation is multiplied by zero in the next layer.  HID is the
// KERNEL's width (tiles, chunks, stream layout), HIDR the model's (bounds and strides of the master weights in the packer).
#define CADM_XDL_MIN_HID 113
#define CADM_XDL_NARROW_HID 128
__host__ __device__ constexpr int xdl_kernel_hid(int hid) { return hid < CADM_XDL_MIN_HID ? CADM_XDL_NARROW_HID : hid; }

// INVARIANT LAST CHUNK OF LAYER 0 (round 6).  Layer 0's inputs are [obs features | action | context]; the context vector does not change
// over a rollout, so a layer-0 chunk that holds ONLY context features (halfcheetah CaDM: K0 = 18 + 6 + 10 = 34, chunk 1 = context features
// 8 and 9; slim humanoid: chunk 2 of 3) contributes the same products in all H steps.  Where that holds (xdl_inv0) EVERY flavour accumulates
// that chunk FIRST -- (bias + chunk NC0-1) + chunk 0 + .. -- and the cooperative kernel runs it ONCE per row tile: the tile prologue leaves
// (HI, LO) of "bias + last chunk" in LDS per tile, the step loop starts its layer-0 accumulators from them and sweeps NC0 - 1 chunks.  The same
// instructions on the same operands in the same order as a flavour that runs the chunk in every step (the wave-tile kernel): bit-identical.
// Stream of the cooperative kernels: layer 0 holds NC0 - 1 chunks per tile; the invariant chunk's fragments (one per tile) follow the head's.
// A timing build on a vanilla model (K0 = 24: one chunk): cfg2 152.8 -> 144.9 us per rollout (tools/ctx_fold_bound.py).
#ifndef CADM_XDL_INV0
#define CADM_XDL_INV0 1
#endif
__host__ __device__ constexpr bool xdl_inv0(int k0, int kdyn, int hid_kernel) {      // kdyn = P + A: the inputs that change from step to step
    return CADM_XDL_INV0 && k0 > kdyn && (k0 + 31) / 32 >= 2 && ((k0 + 31) / 32 - 1) * 32 >= kdyn && (hid_kernel + 15) / 16 <= 13;
}

struct XdlGeo {
    int K0, HID, D, NH, HIDR;
    int INV;     // 1: the cooperative kernels' stream keeps layer 0's invariant chunk apart (above); the wave-tile stream (NW = 1) never does
    int NC0, NT, NCH, NTO, BASE, EXTRA, NTOW;
    int NW;      // waves the tiles are dealt to: CADM_XDL_WAVES (cooperative kernel), 1 (wave-tile kernel, rollout_wt.h: every wave owns ALL tiles)
    __host__ __device__ int ntw(int w) const { return BASE + (w < EXTRA ? 1 : 0); }
    __host__ __device__ int tstart(int w) const { return w * BASE + (w < EXTRA ? w : EXTRA); }
    __host__ __device__ int head_tile(int w, int s) const { return (NW - 1 - w) + NW * s; }          // slot s of wave w (valid if < NTO)
    __host__ __device__ int nhead(int w) const {
        int n = 0;
        for (int s = 0; s < NTOW; ++s) n += head_tile(w, s) < NTO ? 1 : 0;
        return n;
    }
    __host__ __device__ int wave_frags(int w) const { return ntw(w) * NC0 + (NH - 1) * ntw(w) * NCH + nhead(w) * NCH; }      // (INV: NC0 - 1 chunks in front, 1 behind the head)
    __host__ __device__ int member_frags() const {
        int n = 0;
        for (int w = 0; w < NW; ++w) n += wave_frags(w);
        return n;
    }
    __host__ __device__ int bias_tiles() const { return NH * NT + NTO; }
};

inline XdlGeo make_xdl_geo(int K0, int hid_model, int D, int NH, int nw = CADM_XDL_WAVES, int kdyn = -1) {
    XdlGeo g;
    g.NW = nw;
    g.INV = (nw == CADM_XDL_WAVES && kdyn >= 0 && xdl_inv0(K0, kdyn, xdl_kernel_hid(hid_model))) ? 1 : 0;
    const int HID = xdl_kernel_hid(hid_model);
    g.K0 = K0; g.HID = HID; g.HIDR = hid_model; g.D = D; g.NH = NH;
    g.NC0 = (K0 + 31) / 32;
    g.NT = (HID + 15) / 16;
    g.NCH = (g.NT + 1) / 2;
    g.NTO = (D + 7) / 8;
    g.BASE = g.NT / nw;
    g.EXTRA = g.NT % nw;
    g.NTOW = (g.NTO + nw - 1) / nw;
    return g;
}

__host__ __device__ constexpr int xdl_group(int w) { return (CADM_XDL_SEQ && w >= CADM_XDL_WAVES / 2) ? 1 : CADM_XDL_GROUP; }

// position of fragment (local tile ti, chunk c) in a layer's consumption order, for a wave with ntw tiles taken in
// groups of gsz: chunk-major inside a group
__host__ __device__ constexpr int xdl_frag_index(int ntw, int nchl, int ti, int c, int gsz) {
    const int g = ti / gsz;
    const int gs = (ntw - gsz * g) < gsz ? (ntw - gsz * g) : gsz;
    return gsz * g * nchl + c * gs + (ti - gsz * g);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The launcher's plan (rollout_xdl.h: xdl_launch): how a member's row tiles are cut between the kernel flavours.
//   flavour 0: cooperative kernel, one row tile per workgroup   cap 1 unit   cost 1
//           1: cooperative kernel, two row tiles per workgroup  cap 2        CADM_COST_MT2
//           2: wave-tile kernel, 4 tiles per workgroup           cap 4        CADM_COST_WT4   (one wave per SIMD: never wins, exists for the tests)
//           3: wave-tile kernel, 8 tiles per workgroup           cap 8        CADM_COST_WT8   (per round)
// 1 unit = one CU share of the member (n_cus / E tiles); costs in units of the one-tile launch (halfcheetah at the cfg2 / cfg3 geometry,
// 51 workgroups per member: 165 / 261 / 600 / 876 us, profiles/r4_s3_flavour_table.txt).  The cheapest cover is a small
// dynamic programme, f(u) = min over flavours (cost + f(u - cap)); count[o] = launches (rounds) of flavour o.
// The costs are a property of the INSTANTIATION (env kind -> observation width -> state phase and head; hidden width -> tiles per wave):
// xdl_costs(env, hid) is a constexpr of the launcher's template arguments, filled from tools/flavour_table.py runs (round 6,
// profiles/r6_flavour_table.txt; us per rollout / the one-tile launch's):
//   halfcheetah, HID 200:    two tiles 1.59 (260.7 per full round)   wave-tile 4: 3.15-3.2 (517)   wave-tile 8: 5.19 (851.9)
//   slim humanoid, HID 200:  two tiles 1.62 (304.0)                  wave-tile 4: 3.62   wave-tile 8: 5.5
// A wave-tile round that is NOT full is cheaper than a full one -- its workgroups run 5, 6 or 7 waves instead of 8 (wt_launch gives a
// partly filled round the member's whole CU share): wt8p[k - 5] = cost of a round that covers k = 5, 6, 7 units (halfcheetah 735.6 /
// 741.8 / 794.0 us: 4.48 / 4.52 / 4.84; at 6 units one partial round beats three two-tile launches, 741.8 against 776.9 us).
// Geometries nobody measured use halfcheetah's.  -DCADM_COST_MT2=.. etc. override every instantiation (tools/build_variant.sh experiments).
struct XdlCosts { float c[4]; float wt8p[3]; };
__host__ __device__ constexpr XdlCosts xdl_costs(int env_kind, int hid) {
#if defined(CADM_COST_MT2) || defined(CADM_COST_WT4) || defined(CADM_COST_WT8)
#ifndef CADM_COST_MT2
#define CADM_COST_MT2 1.6f
#endif
#ifndef CADM_COST_WT4
#define CADM_COST_WT4 3.6f
#endif
#ifndef CADM_COST_WT8
#define CADM_COST_WT8 5.3f
#endif
    return XdlCosts{{1.0f, CADM_COST_MT2, CADM_COST_WT4, CADM_COST_WT8}, {CADM_COST_WT8, CADM_COST_WT8, CADM_COST_WT8}};
#else
    return env_kind == 2 /* CADM_ENV_SLIM_HUMANOID */ ? XdlCosts{{1.0f, 1.62f, 3.62f, 5.5f}, {4.78f, 4.80f, 5.15f}}
                                                      : XdlCosts{{1.0f, 1.59f, 3.2f, 5.19f}, {4.48f, 4.52f, 4.84f}};
#endif
}
inline void xdl_plan_units(int units, bool mt2_ok, bool wt_ok, int (&count)[4], XdlCosts costs = xdl_costs(0, 200)) {
    const int capu[4] = {1, mt2_ok ? 2 : 0, wt_ok ? 4 : 0, wt_ok ? 8 : 0};
    const float* cost = costs.c;
    for (int o = 0; o < 4; ++o) count[o] = 0;
    // beyond 64 units the answer is "rounds of the biggest flavour" plus the plan of the rest
    const int big = capu[3] ? 3 : capu[1] ? 1 : 0;
    int u = units;
    if (u > 64) { const int k = (u - 56) / capu[big]; count[big] += k; u -= k * capu[big]; }
    float f[65];
    int pick[65];
    f[0] = 0.0f;
    pick[0] = -1;
    for (int v = 1; v <= u; ++v) {
        f[v] = 1e30f;
        for (int o = 0; o < 4; ++o) {
            if (!capu[o]) continue;
            // (a wave-tile-8 round that covers only 5, 6 or 7 units -- the LAST launch of a plan takes the ragged rest -- at its own cost)
            const float co = (o == 3 && v >= 5 && v <= 7) ? costs.wt8p[v - 5] : cost[o];
            const float c = co + f[v > capu[o] ? v - capu[o] : 0];
            if (c < f[v] - 1e-6f) { f[v] = c; pick[v] = o; }
        }
    }
    for (int v = u; v > 0; v -= capu[pick[v]]) ++count[pick[v]];
}
