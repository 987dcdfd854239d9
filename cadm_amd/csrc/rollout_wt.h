#pragma once
// "Wave-tile" flavour of the split-f16 rollout kernel (rollout_xdl.h) for LARGE batches: every wave advances its OWN 16-row
// tile through the whole horizon (reference core/utils.py:431-472), the 8 waves of a workgroup share nothing but the weights.
//
// Why a second flavour.  The cooperative kernel (8 waves on one or two row tiles) is built for BASELINE cfg2, where a CU has 16 rows
// and the step is a latency chain: every layer is a barrier-delimited sweep, each wave re-reads the tile's whole activation
// block from LDS, and the 13 hidden tiles fall 4 / 3 / 3 / 3 on the SIMDs.  With many rows per CU none of that is needed:
//   * a lane's D fragments of output tiles (2c, 2c+1) ARE its B fragment of chunk c of the next layer (xdl_geo.h), so a wave that
//     computes ALL output tiles of its 16 rows carries the activations from layer to layer IN REGISTERS -- no LDS traffic, no
//     barrier between the layers, no cross-wave dependency in the data path at all;
//   * the rollout state (observation, return) stays in the lanes that receive the head's output for it, so the Gaussian head, the
//     state update, the reward and the input assembly are in-lane (the input features take one round trip through a wave-private
//     1-4 KB LDS image to get from "dims per lane" to the B-operand layout);
//   * every SIMD hosts the same work (two waves x 13 tiles), whatever the hidden width;
//   * the weights are what the waves share: the member's fragment stream (ONE consumption order for all waves: layer by layer,
//     tile pair by tile pair) goes L2 -> LDS once per workgroup and step by LDS-DMA (buffer_load_dwordx4 .. lds, no registers), in
//     BLOCKS of one tile pair (28 KB at HID = 200), double-buffered: at every block boundary one barrier says "block b has
//     landed, block b-1 is no longer read", then the waves request block b+1 and compute block b from LDS.
//     L2 -> CU traffic per row drops 5x against the one-tile cooperative kernel (640 KB per step for 128 rows instead of 404 KB for 16).
// A row's arithmetic -- MFMA order per accumulator, epilogue, head, state update, noise counters, reward summation order -- is that
// of the cooperative kernel: the flavours agree BIT FOR BIT (tests/test_gpu_rowtiles.py, tools/fuzz_rollout.py), so the launcher
// may cut a batch between them freely (xdl_launch).
//
// Included by rollout_xdl.h (uses its geometry class XC, epilogue arithmetic XHiddenEpi and helpers).

// TIMING EXPERIMENTS ONLY (tools/build_variant.sh; wrong results): CADM_WT_EXPERIMENT_NOMFMA / _NOFRAG (fragments not read from LDS) /
// _NOSTATE (no state update, noise, input assembly) / _NOBAR (no block barriers) / _NOGLDS (no weight requests);
// CADM_XDL_EXPERIMENT_NOEPI also applies (profiles/r4_wave_tile.md).
namespace {

template <class G>
struct WT {
    static constexpr int NW = 8, NTHR = NW * 64;
    static constexpr int NT = G::NT, NCH = G::NCH, NC0 = G::NC0, NTO = G::NTO, NH = G::NHC;
    static constexpr int NG = (NT + 1) / 2, NGH = (NTO + 1) / 2;       // tile pairs of a hidden-type layer / of the head
    static constexpr bool AVAILABLE = G::NPROD == 3 && NCH <= 8 && NC0 <= NCH && NH <= 5;      // (the layer loop is unrolled: code size)
    static constexpr int BLK = 2 * NCH * CADM_XDL_FRAG_BYTES;            // bytes of the largest block (one pair of hidden tiles)
    static constexpr int NJ = NTO;                                       // pair slots per lane: slot j <-> head tile j, pair 4 j + (lane >> 4)
    static constexpr int NAJ = (G::A + 3) / 4;                           // action slots per lane
    static constexpr int TABW = 28;
    // two register sets for the weight fragments (chunk c + 1 requested from LDS before chunk c's MFMAs) where the registers allow:
    // wide observation spaces keep more rollout state and head output per lane (slim humanoid: 6 pair slots)
#ifndef CADM_WT_NBUF
#define CADM_WT_NBUF 2
#endif
    static constexpr int DBUF = (NTO > 4 || (NTO > 3 && NCH > 7)) ? 1 : CADM_WT_NBUF;      // register sets of weight fragments
    // LDS carve (bytes)
    static constexpr int SLOTS = 2;                                      // ring depth: the block requested at a boundary is the one right behind the block computed
    static constexpr int RING = 0;                                       // [SLOTS][BLK]
    static constexpr int BIAS = RING + SLOTS * BLK;                          // one float4 per (tile, lane group): 64 B per tile
    static constexpr int BIAS_BYTES = (NH * NT + NTO) * 64;
    static constexpr int TAB = BIAS + rup(BIAS_BYTES, 16);               // per (slot, lane group) constants of the state update
    static constexpr int TAB_BYTES = NJ * 4 * TABW * 4;
    static constexpr int STATS = TAB + TAB_BYTES;                        // act mean [A], 1 / (act std + 1e-10) [A]
    static constexpr int WAVE0 = STATS + rup(2 * G::A * 4, 16);          // wave-private regions: x_in image, control cost of the tile's rows
    static constexpr int XW_BYTES = cmax(2 * NC0 * 1024, 1024);          // (also the scratch of the return's final sum: 16 rows x 16 slots)
    static constexpr int wave_bytes(int H) { return XW_BYTES + rup(16 * H * 4, 16); }
    static size_t lds_bytes(int H) { return (size_t)WAVE0 + (size_t)NW * wave_bytes(H); }
    // fragments of a block
    static constexpr int gs_hidden(int g) { return (NT - 2 * g) < 2 ? (NT - 2 * g) : 2; }
    static constexpr int gs_head(int g) { return (NTO - 2 * g) < 2 ? (NTO - 2 * g) : 2; }
    // the step's block sequence: position q = layer * NG + pair (layers 0 .. NH-1), then the head's pairs; cyclic over steps
    static constexpr int STEP_BLOCKS = NH * NG + NGH;
    static __host__ __device__ constexpr int wrap(int q) { return q >= STEP_BLOCKS ? (q - STEP_BLOCKS >= STEP_BLOCKS ? q - 2 * STEP_BLOCKS : q - STEP_BLOCKS) : q; }
    static __host__ __device__ constexpr int block_frags(int q0) {          // fragments of block q
        const int q = wrap(q0);
        return q < NG ? gs_hidden(q) * NC0 : q < NH * NG ? gs_hidden(q % NG) * NCH : gs_head(q - NH * NG) * NCH;
    }
};

// The ring's bookkeeping: two slots; `par` = slot of the block computed next, `next` = stream offset of the block REQUESTED next.
struct WTRing {
    __amdgpu_buffer_rsrc_t rsrc;   // the whole stream buffer (all members)
    unsigned mbase;                // this member's byte offset in it
    unsigned next, bytes;
    int par;
};

// one LDS-DMA piece: 64 lanes x 16 B from stream offset `off` of this member to lds_dst (wave-uniform) + lane * 16
__device__ __forceinline__ void wt_glds16(const WTRing& rg, unsigned off, int lane, unsigned char* lds_dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rg.rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, lane * 16, rg.mbase + off, 0, 0);
}


// this wave's share of a block of nf fragments (= 2 nf pieces of 1 KB, piece i to wave i mod 8) into slot `slot_idx`
// (NF is a template parameter: whether piece i exists is then decided at compile time for all but the last round of pieces -- with a
//  run-time block size hipcc emitted a compare and a branch per piece and block, 6 % of the launch)
template <class G, int NF>
__device__ __forceinline__ void wt_request(WTRing& rg, unsigned char* sm, int slot_idx, int wave, int lane) {
    using W = WT<G>;
#ifndef CADM_WT_EXPERIMENT_NOGLDS
    unsigned char* dst = sm + W::RING + slot_idx * W::BLK;
    static_for(std::make_integer_sequence<int, (2 * NF + W::NW - 1) / W::NW>{}, [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int piece = wave + W::NW * i;
        // buffer form of the LDS-DMA load: descriptor + scalar offset + the lane's 16 B -- no per-request 64-bit address arithmetic on the VALU
        if constexpr (W::NW * i + W::NW - 1 < 2 * NF) wt_glds16(rg, rg.next + (unsigned)piece * 1024u, lane, dst + piece * 1024);
        else { if (piece < 2 * NF) wt_glds16(rg, rg.next + (unsigned)piece * 1024u, lane, dst + piece * 1024); }
    });
#endif
    rg.next += (unsigned)NF * CADM_XDL_FRAG_BYTES;
    if (rg.next >= rg.bytes) rg.next = 0;
}

// Block boundary.  One barrier says: block b has landed (s_waitcnt vmcnt(0) in front of it: every wave's share of the
// requests issued one block ago is in LDS) and nobody reads block b-1 any more; then block b+1 is requested into the slot b-1 left.
// Measured alternatives (profiles/r4_wave_tile.md, same box): three slots with the next block's first fragments requested from LDS a
// block ahead +-0; a raw s_barrier behind a counted s_waitcnt vmcnt(k) (requests in flight across the barrier) +4 %; the requests
// issued in the VALU-only epilogue instead of behind the barrier +-0.  The requests themselves are 15-22 % of the launch (a build
// without them), the barriers 4 %.
template <class G, int NF_NEXT>
__device__ __forceinline__ const unsigned char* wt_block_boundary(WTRing& rg, unsigned char* sm, int wave, int lane) {
    using W = WT<G>;
#if !defined(CADM_WT_EXPERIMENT_NOBAR)
    // every wave's share of the requests issued one block ago must be IN LDS before any wave reads the block: the wait is stated here and not
    // left to hipcc (a workgroup-scope barrier does not have to drain vmcnt; tests/test_isa_hygiene.py checks the shipped code for it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
    const int ahead = rg.par + W::SLOTS - 1;
    wt_request<G, NF_NEXT>(rg, sm, ahead >= W::SLOTS ? ahead - W::SLOTS : ahead, wave, lane);
    const unsigned char* cur = sm + W::RING + rg.par * W::BLK;
    rg.par = rg.par + 1 == W::SLOTS ? 0 : rg.par + 1;
    return cur;
}

// accumulate GS tiles over NCHL chunks from the block at `slot`: hi += w1 x1 + w1 x2, lo += w2 x1 per chunk, in the cooperative
// kernel's order per accumulator.  HEAD: the block stores its fragments tile-major (xdl_geo.h), else chunk-major.
// LASTFIRST: the chunks are taken in the order NCHL-1, 0, 1, .. (layer 0 of a geometry whose last chunk holds only context features: every
// flavour accumulates that chunk first, xdl_geo.h "invariant last chunk" -- the cooperative kernel runs it once per row tile).
template <int GS, int NCHL, bool HEAD, int DBUF, bool LASTFIRST = false, int NX>
__device__ __forceinline__ void wt_accumulate(const unsigned char* slot, int lane, const f16x8 (&X1)[NX], const f16x8 (&X2)[NX],
                                              floatx4 (&hi)[2], floatx4 (&lo)[2]) {
    // the fragments of chunk c + 1 are requested before chunk c's MFMAs (two register sets): a wave alone on its SIMD otherwise
    // waits out the LDS latency at every chunk
    uintx4 w[DBUF][GS][2];
#ifdef CADM_WT_EXPERIMENT_NOFRAG
#pragma unroll
    for (int q = 0; q < DBUF; ++q)
#pragma unroll
        for (int k = 0; k < GS; ++k) { w[q][k][0] = uintx4{1u, 2u, 3u, 4u}; w[q][k][1] = uintx4{1u, 2u, 3u, 4u}; }
#endif
    auto wload = [&](auto cc) {      // the cc-th chunk in processing order
        constexpr int c = decltype(cc)::value;
        constexpr int co = LASTFIRST ? (c == 0 ? NCHL - 1 : c - 1) : c;
#pragma unroll
        for (int k = 0; k < GS; ++k)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const int fi = HEAD ? k * NCHL + co : co * GS + k;
#ifndef CADM_WT_EXPERIMENT_NOFRAG
                w[c % DBUF][k][part] = *reinterpret_cast<const uintx4*>(slot + (fi * 2 + part) * 1024 + lane * 16);
#else
                (void)fi;
#endif
            }
    };
    static_for(std::make_integer_sequence<int, (DBUF - 1 < NCHL ? DBUF - 1 : NCHL)>{}, [&](auto cc) { wload(cc); });
    static_for(std::make_integer_sequence<int, NCHL>{}, [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int co = LASTFIRST ? (c == 0 ? NCHL - 1 : c - 1) : c;      // the chunk whose operands these MFMAs read
        if constexpr (c + DBUF - 1 < NCHL) wload(std::integral_constant<int, c + DBUF - 1>{});
#ifdef CADM_WT_EXPERIMENT_NOMFMA
#pragma unroll
        for (int k = 0; k < GS; ++k) asm volatile("" : "+v"(hi[k]), "+v"(lo[k]) : "v"(w[c % DBUF][k][0]), "v"(w[c % DBUF][k][1]), "v"(X1[co]), "v"(X2[co]));
#else
#pragma unroll
        for (int k = 0; k < GS; ++k) hi[k] = xmfma(w[c % DBUF][k][0], X1[co], hi[k]);
#pragma unroll
        for (int k = 0; k < GS; ++k) lo[k] = xmfma(w[c % DBUF][k][1], X1[co], lo[k]);
#pragma unroll
        for (int k = 0; k < GS; ++k) hi[k] = xmfma(w[c % DBUF][k][0], X2[co], hi[k]);
#endif
        if constexpr (DBUF > 1) __builtin_amdgcn_sched_barrier(0);      // pin the pipeline: no load sinking / hoisting across chunks
    });
}

template <class G, int NOISE>
__global__ __launch_bounds__(WT<G>::NTHR) void rollout_wt_kernel(const RolloutArgs a) {
    using W = WT<G>;
    constexpr int D = G::D, A = G::A, P = G::P, C = G::C, K0 = G::K0, NC0 = G::NC0, NCH = G::NCH, NT = G::NT, NTO = G::NTO;
    constexpr int NP = G::NP, ENV = G::ENV, XNH = G::NHC, NJ = W::NJ, NAJ = W::NAJ, TABW = W::TABW;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    fp16_saturate_on();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, fg = lane >> 4;                     // row of the tile, lane group (units 4 fg .. 4 fg + 3 of a D tile)
    const int item = rollout_item();
    const int e = item / a.wgs_per_member, grp = item % a.wgs_per_member;
    const int H = a.H;
    const int wbytes = W::wave_bytes(H);
    unsigned char* xw = sm + W::WAVE0 + wave * wbytes;           // this wave's x_in image: [2 parts][NC0][64 lanes] x 16 B
    float* ctrl_s = reinterpret_cast<float*>(xw + W::XW_BYTES);  // [16 rows][H]
    float* tab = reinterpret_cast<float*>(sm + W::TAB);
    float* stats = reinterpret_cast<float*>(sm + W::STATS);

    // ---- once per workgroup: bias tiles (one float4 per tile and lane group), state-update constants, action statistics ----
    {
        const uintx4* src = reinterpret_cast<const uintx4*>(a.xb + (size_t)e * a.xb_member);
        for (int i = tid; i < (XNH * NT + NTO) * 4; i += W::NTHR) reinterpret_cast<uintx4*>(sm + W::BIAS)[i] = src[(i >> 2) * 64 + (i & 3) * 16];
    }
    for (int i = tid; i < A; i += W::NTHR) {
        stats[i] = a.act_mean[i];
        stats[A + i] = 1.0f / (a.act_std[i] + 1e-10f);
    }
    auto xin_base = [&](int f) { return ((f >> 5) * 64 + ((f & 31) >> 3) * 16) * 16 + (f & 7) * 2; };   // byte offset (part 0) of feature f of row 0
    if (tid < NJ * 4) {              // entry (slot j, lane group g): pair dp = 4 j + g  (the cooperative kernel's table, rollout_xdl.h)
        float* te = tab + tid * TABW;
        const int dp = tid;          // (= 4 j + g)
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
                te[18 + 2 * h + i] = __builtin_bit_cast(float, on ? xin_base(f) : G::SPARE ? xin_base(K0) : -1);
                te[22 + 2 * h + i] = __builtin_bit_cast(float, on ? fop[i] : G::SPARE ? 3 : 0);      // (op 3: no feature, 0.0 into the spare slot)
            }
        }
    }

    // ---- the weight ring: the member's stream, block by block, forever (a step ends where the next one starts) ----
    WTRing rg;
    rg.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xw1, 0, a.xw1_member_b * (unsigned)a.E, 0x00020000);
    rg.mbase = (unsigned)e * a.xw1_member_b;
    rg.bytes = a.xw1_member_b;
    rg.par = 0;
    rg.next = 0;
    // the first SLOTS - 1 blocks of the step; every block boundary then requests the block SLOTS - 1 ahead
    static_for(std::make_integer_sequence<int, W::SLOTS - 1>{}, [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        wt_request<G, W::block_frags(b)>(rg, sm, b, wave, lane);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (as at every block boundary: this wave's pieces of the first block have landed)
    __syncthreads();

    const int r16 = r * 16;
    float nanfold = 0.0f;              // 0 * v of every input feature: NaN iff the row ever saw a non-finite one (rollout_xdl.h put_x)
    auto put_x = [&](int off, float v) __attribute__((always_inline)) {
        nanfold = fmaf(0.0f, v, nanfold);
        asm volatile("" : "+v"(v));         // (an fp32 number before it is split: rollout_xdl.h put_x)
        _Float16 h1, h2;                    // (saturating split: fp16_saturate_on)
        xsplit(v, h1, h2);
        *reinterpret_cast<_Float16*>(xw + off) = h1;
        *reinterpret_cast<_Float16*>(xw + NC0 * 1024 + off) = h2;
    };

    // rounds: wave w of workgroup grp takes tile  round * (wgs * nwa) + w * wgs + grp  of this launch (nwa = active waves per workgroup)
    const int per_round = a.wgs_per_member * a.wt_waves;
    const int rounds = (a.tile_count + per_round - 1) / per_round;
    for (int round = 0; round < rounds; ++round) {
        const int tile = round * per_round + wave * a.wgs_per_member + grp;
        const bool active = wave < a.wt_waves && tile < a.tile_count;      // (wave-uniform)
        // ---- this lane's row ----
        int re = (a.tile0 + tile) * 16 + r;
        const bool valid = active && re < a.rows_per_member;
        if (!(re < a.rows_per_member)) re = a.rows_per_member - 1;
        if (!active) re = 0;
        const int cidx = re / a.PE, jl = re % a.PE;
        const int mi = cidx / a.n_local, nl = cidx % a.n_local;
        const int j = e * a.PE + jl;
        const int lr = (mi * a.n_local + nl) * a.p + j;                                       // local row (returns / eps / traj)
        const unsigned grow = (unsigned)((mi * a.n_global + a.cand_offset + nl) * a.p + j);   // global row (RNG counter)
        const int abase = ((mi * a.n_global + a.cand_offset + nl) * H) * A;
        const int ep = j % a.E;
        int ctx_off;
        if (!a.quirks) ctx_off = ((j / a.PE) * a.m + mi) * C;            // own member's context
        else if (a.it & 1) ctx_off = (mi * a.E + ep) * C;                // Q2: [E,m] memory reread as [m,E]
        else ctx_off = (ep * a.m + mi) * C;                              // Q1: encoder j % E

        float po[NJ][2], retq[4] = {0.0f, 0.0f, 0.0f, 0.0f}, areg[NAJ];
        nanfold = 0.0f;
        if (active) {
            for (int i = lane; i < W::XW_BYTES / 16; i += 64) reinterpret_cast<uintx4*>(xw)[i] = uintx4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = 2 * (4 * jj + fg) + h;
                    const int dc = d < D ? d : 0;
                    po[jj][h] = a.obs_rows ? a.obs_rows[(size_t)lr * D + dc] : a.obs[mi * D + dc];   // :432
                }
#pragma unroll
            for (int ai = 0; ai < NAJ; ++ai) {
                const int ac = fg + 4 * ai;
                areg[ai] = ac < A ? a.actions[abase + ac] : 0.0f;
            }
            if constexpr (C > 0) {
                for (int f = P + A + fg; f < K0; f += 4) put_x(xin_base(f) + r16, a.ctx_vec[ctx_off + f - P - A]);   // static: context (:433-439)
            }
            for (int t = fg; t < H; t += 4) ctrl_s[r * H + t] = ctrl_term<ENV>(a.actions + abase + t * A, A);
        } else {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) po[jj][0] = po[jj][1] = 0.0f;
#pragma unroll
            for (int ai = 0; ai < NAJ; ++ai) areg[ai] = 0.0f;
        }

        floatx4 hv[NTO];                                   // head output of the previous step: (mu0, mu1, lv0, lv1) of pair 4 j + fg
#pragma unroll
        for (int jj = 0; jj < NTO; ++jj) hv[jj] = floatx4{0.f, 0.f, 0.f, 0.f};

        for (int t = 0; t <= H; ++t) {
            f16x8 X1[NCH], X2[NCH];
#ifdef CADM_WT_EXPERIMENT_NOSTATE
            if (active) {
#pragma unroll
                for (int c = 0; c < NC0; ++c) {
                    X1[c] = *reinterpret_cast<const f16x8*>(xw + ((0 * NC0 + c) * 64 + lane) * 16);
                    X2[c] = *reinterpret_cast<const f16x8*>(xw + ((1 * NC0 + c) * 64 + lane) * 16);
                }
            }
            if (false) {
#else
            if (active) {
#endif
                // ===== state update from step t-1's head (:348-365,463-466) + reward (:469-471) + input assembly (:442-460), in-lane =====
                // Everything lane-dependent is derived HERE from opaque copies: hipcc otherwise hoists ~45 loop-invariant LDS
                // addresses / row pointers out of the step loop and keeps them in registers through the layers, where the
                // activations (112 registers) leave none to spare for the weight fragments' lookahead.
                int lane_o = lane, lr_o = lr;
                asm volatile("" : "+v"(lane_o), "+v"(lr_o));
                const int r = lane_o & 15, fg = lane_o >> 4, r16 = r * 16, lr = lr_o;
                // Gaussian-head noise: one Philox call per TWO pair slots (pairs 8 q + fg and 8 q + 4 + fg share a call: rollout_env.h eps_group /
                // eps_sub) -- made at the even slot, the odd slot's pair carried over in two registers
                float2 z_odd = make_float2(0.0f, 0.0f);
                static_for(std::make_integer_sequence<int, NJ>{}, [&](auto jc) {
                    constexpr int jj = decltype(jc)::value;
                    const int dp = 4 * jj + fg;
                    if (dp < NP) {
                        floatx4 tq[7];
#pragma unroll
                        for (int q = 0; q < 7; ++q) tq[q] = *reinterpret_cast<const floatx4*>(tab + dp * TABW + 4 * q);
                        auto tv = [&](int w) { return tq[w >> 2][w & 3]; };
                        auto ti_ = [&](int w) { const float fv = tq[w >> 2][w & 3]; return __float_as_int(fv); };
                        if (t > 0) {
                            const floatx4 v = hv[jj];
                            float2 z = make_float2(0.0f, 0.0f);
                            if constexpr (NOISE == CADM_NOISE_INJECT) {
                                const float* epp = a.eps + ((size_t)(t - 1) * a.m * a.n_local * a.p + lr) * D + 2 * dp;
                                z.x = epp[0];
                                z.y = (2 * dp + 1 < D) ? epp[1] : 0.0f;
                            } else if constexpr (NOISE == CADM_NOISE_PHILOX) {
                                if constexpr ((jj & 1) == 0) {
                                    uint32_t pc[4] = {grow, (uint32_t)(t - 1), (uint32_t)(fg | ((jj >> 1) << 2)), CADM_STREAM_EPS | ((uint32_t)a.it << 8)};
                                    uint32_t pk[2] = {a.seed, a.call};
                                    philox_rounds<0, 10>(pc, pk);
                                    box_muller(u01(pc[0]), u01(pc[1]), z.x, z.y);
                                    if constexpr (jj + 1 < NJ) box_muller(u01(pc[2]), u01(pc[3]), z_odd.x, z_odd.y);
                                } else {
                                    z = z_odd;
                                }
                            }
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                float delta = v[h] * tv(2 + h) + tv(0 + h);                          // denormalize, :349
                                if constexpr (NOISE != CADM_NOISE_NONE) {
                                    const float sd = head_sd(v[2 + h], tv(6 + h), tv(8 + h), tv(4 + h));   // :356-363
                                    delta = delta + (h ? z.y : z.x) * sd;                               // :365
                                }
                                po[jj][h] = postproc<ENV>(2 * dp + h, po[jj][h], delta);                // :466
                            }
                            if (a.traj && valid) {
                                float* tp = a.traj + ((size_t)(t - 1) * a.m * a.n_local * a.p + lr) * D + 2 * dp;
                                tp[0] = po[jj][0];
                                if (2 * dp + 1 < D) tp[1] = po[jj][1];
                            }
                        }
                        // a row's return is summed over 16 pair slots (pair dp and dp + 16 share slot dp mod 16), in slot order: rollout_xdl.h
                        if constexpr (ENV == CADM_ENV_CARTPOLE) {
                            if (t > 0) retq[jj & 3] += reward_part<ENV>(dp, po[jj][0], po[jj][1], 0.0f);   // reads NEXT obs
                        } else {
                            if (t < H) retq[jj & 3] += reward_part<ENV>(dp, po[jj][0], po[jj][1], ctrl_s[r * H + t]);
                        }
                        if (t < H) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                float sn = 0.0f, cs = 0.0f;
                                if constexpr (ENV == CADM_ENV_HALFCHEETAH) {                 // the one trig pair (obs dim 2)
                                    if (ti_(22 + 2 * h) == 1) sincos_cw(po[jj][h], &sn, &cs);
                                }
#pragma unroll
                                for (int i = 0; i < 2; ++i) {
                                    const int off = ti_(18 + 2 * h + i), op = ti_(22 + 2 * h + i);
                                    if constexpr (G::SPARE) {      // unconditional: selects, no exec-mask regions (XC::SPARE)
                                        float pv = po[jj][h];
                                        pv = op == 1 ? sn : pv;
                                        pv = op == 2 ? cs : pv;
                                        const float xv = (pv - tv(10 + 2 * h + i)) * tv(14 + 2 * h + i);   // :450-451
                                        put_x(off + r16, op == 3 ? 0.0f : xv);
                                    } else if (off >= 0) {
                                        const float pv = op == 1 ? sn : op == 2 ? cs : po[jj][h];
                                        put_x(off + r16, (pv - tv(10 + 2 * h + i)) * tv(14 + 2 * h + i));   // :450-451
                                    }
                                }
                            }
                        }
                    }
                });
                if (t < H) {
#pragma unroll
                    for (int ai = 0; ai < NAJ; ++ai) {
                        const int ac = fg + 4 * ai;
                        if (ac < A) {
                            float v = areg[ai];
                            if (a.norm_actions) v = (v - stats[ac]) * stats[A + ac];   // :443
                            put_x(xin_base(P + ac) + r16, v);
                            if (t + 1 < H) areg[ai] = a.actions[abase + (t + 1) * A + ac];
                        }
                    }
                    // the step's B operand of layer 0 (other lanes of this wave wrote it: LDS executes a wave's accesses in order)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int c = 0; c < NC0; ++c) {
                        X1[c] = *reinterpret_cast<const f16x8*>(xw + ((0 * NC0 + c) * 64 + lane) * 16);
                        X2[c] = *reinterpret_cast<const f16x8*>(xw + ((1 * NC0 + c) * 64 + lane) * 16);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the next step's writes must not overtake these reads in program order)
                }
            }
            if (t == H) break;

            // ================= dense layers: activations stay in registers =================
            f16x8 Y1[NCH], Y2[NCH];
            // One block: barrier, transport of the NEXT block (NFN fragments), GS tiles accumulated over NCHL chunks of XA / XB from the
            // current slot, epilogue.
            auto run_block = [&](auto nfn_c, auto gs_c, auto nchl_c, auto head_c, auto lf_c, const f16x8 (&XA)[NCH], const f16x8 (&XB)[NCH], int bias_tile,
                                 auto&& epilogue) __attribute__((always_inline)) {
                constexpr int NFN = decltype(nfn_c)::value, GS = decltype(gs_c)::value, NCHL = decltype(nchl_c)::value;
                constexpr bool HEAD = decltype(head_c)::value;
                const unsigned char* slot = wt_block_boundary<G, NFN>(rg, sm, wave, lane);
                if (active) {
                    floatx4 hi[2], lo[2];
#pragma unroll
                    for (int k = 0; k < GS; ++k) {
                        hi[k] = *reinterpret_cast<const floatx4*>(sm + W::BIAS + (bias_tile + k) * 64 + fg * 16);
                        lo[k] = floatx4{0.f, 0.f, 0.f, 0.f};
                    }
                    wt_accumulate<GS, NCHL, HEAD, W::DBUF, decltype(lf_c)::value>(slot, lane, XA, XB, hi, lo);
                    epilogue(hi, lo);
                }
            };
            using TrueT = std::integral_constant<bool, true>;
            using FalseT = std::integral_constant<bool, false>;
            // one hidden-type layer: NCHL input chunks in X, all NT output tiles into Y, pair by pair
            auto hidden_layer = [&](auto nchl_c, auto lf_c, int layer, const f16x8 (&IN1)[NCH], const f16x8 (&IN2)[NCH], f16x8 (&OUT1)[NCH], f16x8 (&OUT2)[NCH]) __attribute__((always_inline)) {
                constexpr int NCHL = decltype(nchl_c)::value;
                static_for(std::make_integer_sequence<int, W::NG>{}, [&](auto gc) {
                    constexpr int g = decltype(gc)::value, GS = W::gs_hidden(g);
                    static_assert(W::SLOTS == 2, "a block boundary names the block right behind the current one");
                    auto epilogue = [&](floatx4 (&hi)[2], floatx4 (&lo)[2]) __attribute__((always_inline)) {
                        const XHiddenEpi<G, CADM_EPI_PACKED_WT> epi{nullptr, nullptr, 0, 0, 0, 0, 0};
                        typename XHiddenEpi<G, CADM_EPI_PACKED_WT>::State st[2];
                        const floatx4 zero = floatx4{0.f, 0.f, 0.f, 0.f};
                        static_for(std::make_integer_sequence<int, 5>{}, [&](auto sc) {
#pragma unroll
                            for (int k = 0; k < GS; ++k) epi.template stage<decltype(sc)::value>(0, 0, hi[k], lo[k], zero, st[k]);
                        });
                        const f16x4 z4 = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
                        OUT1[g] = __builtin_shufflevector(st[0].h1, GS > 1 ? st[1].h1 : z4, 0, 1, 2, 3, 4, 5, 6, 7);
                        OUT2[g] = __builtin_shufflevector(st[0].h2, GS > 1 ? st[1].h2 : z4, 0, 1, 2, 3, 4, 5, 6, 7);
                    };
                    using GSc = std::integral_constant<int, GS>;
                    const int bt = layer * NT + 2 * g;
                    // the block behind this one: the layer's next pair; behind a layer's last pair the next layer's first (the head's
                    // first behind the last hidden layer's)
                    if constexpr (g + 1 < W::NG)
                        run_block(std::integral_constant<int, W::gs_hidden(g + 1 < W::NG ? g + 1 : 0) * NCHL>{}, GSc{}, nchl_c, FalseT{}, lf_c, IN1, IN2, bt, epilogue);
                    else if (layer + 1 < XNH) run_block(std::integral_constant<int, W::gs_hidden(0) * NCH>{}, GSc{}, nchl_c, FalseT{}, lf_c, IN1, IN2, bt, epilogue);
                    else run_block(std::integral_constant<int, W::gs_head(0) * NCH>{}, GSc{}, nchl_c, FalseT{}, lf_c, IN1, IN2, bt, epilogue);
                });
            };
            // layer 0
            hidden_layer(std::integral_constant<int, NC0>{}, std::integral_constant<bool, G::INV>{}, 0, X1, X2, Y1, Y2);      // (G::INV: the invariant last chunk first, xdl_geo.h)
            // hidden layers 1 .. NH-1, unrolled: the two activation register sets swap roles from layer to layer (a rolled loop had to move
            // 2 x NCH x 4 registers per layer: -2.7 % of a full round, profiles/r4_wave_tile.md; nets deeper than 5 layers stay on the
            // cooperative kernel, WT::AVAILABLE)
            static_for(std::make_integer_sequence<int, XNH - 1>{}, [&](auto lc) {
                constexpr int l = decltype(lc)::value + 1;
                if constexpr (l & 1) hidden_layer(std::integral_constant<int, NCH>{}, FalseT{}, l, Y1, Y2, X1, X2);
                else hidden_layer(std::integral_constant<int, NCH>{}, FalseT{}, l, X1, X2, Y1, Y2);
            });
            f16x8 (&H1)[NCH] = ((XNH - 1) & 1) ? X1 : Y1;      // the last hidden layer's output: the head's input
            f16x8 (&H2)[NCH] = ((XNH - 1) & 1) ? X2 : Y2;
            // head: (mu | logvar) of 8 dims per tile, recombined in-lane
            static_for(std::make_integer_sequence<int, W::NGH>{}, [&](auto gc) {
                constexpr int g = decltype(gc)::value, GS = W::gs_head(g);
                constexpr int q = XNH * W::NG + g;
                auto epilogue = [&](floatx4 (&hi)[2], floatx4 (&lo)[2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int k = 0; k < GS; ++k)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) hv[2 * g + k][qq] = fmaf(lo[k][qq], 4.8828125e-4f, hi[k][qq]);
                };
                run_block(std::integral_constant<int, W::block_frags(q + 1)>{}, std::integral_constant<int, GS>{}, std::integral_constant<int, NCH>{}, TrueT{}, FalseT{},
                          H1, H2, XNH * NT + 2 * g, epilogue);
            });
        }

        // ---- a row's return = sum of its 16 pair slots, in slot order (the cooperative kernel's reduction) ----
        if (active) {
            retq[0] += nanfold;      // (+-0, or NaN for a row that saw a non-finite input)
            float* rs = reinterpret_cast<float*>(xw);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) rs[r * 16 + 4 * q + fg] = retq[q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (fg == 0 && valid) {
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) s += rs[r * 16 + i];
                a.returns_rows[lr] = s;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the ring's last request)
}

// one launch of the wave-tile kernel: row tiles [a.tile0, a.tile0 + a.tile_count) of every member, `waves` (4 or 8) tiles per workgroup and round
template <class G, int NOISE>
int wt_launch_noise(cadm_ctx* ctx, const RolloutArgs& a, int rows_per_member, int waves, hipStream_t s) {
    using W = WT<G>;
    RolloutArgs args = a;
    int per_member = ctx->n_cus / ctx->E;
    if (per_member < 1) per_member = 1;
    // the member's whole CU share, however few tiles there are: a partly filled round then has fewer active waves per workgroup (a
    // SIMD with one wave finishes its step in less than half the time of one with two), not fewer workgroups
    args.wgs_per_member = a.tile_count < per_member ? a.tile_count : per_member;
    args.rows_per_member = rows_per_member;
    args.wt_waves = waves;
    const size_t lds = W::lds_bytes(a.H);
    if (lds > 160 * 1024) {
        cadm_set_error("rollout (wave-tile): horizon %d needs %zu B of LDS (> 160 KiB)", a.H, lds);
        return CADM_EINVAL;
    }
    const void* fn = reinterpret_cast<const void*>(&rollout_wt_kernel<G, NOISE>);
    if (!ctx->attr_done.count(fn)) {
        CADM_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_done.insert(fn);
    }
    if (a.dry_run) return CADM_OK;
    hipLaunchKernelGGL((rollout_wt_kernel<G, NOISE>), dim3(args.wgs_per_member * ctx->E), dim3(W::NTHR), lds, s, args);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

template <class G, int NOISE>
int wt_launch(cadm_ctx* ctx, const RolloutArgs& a, int rows_per_member, int waves, hipStream_t s) {
    const int mode = a.deterministic ? CADM_NOISE_NONE : a.eps ? CADM_NOISE_INJECT : CADM_NOISE_PHILOX;
    if constexpr (NOISE >= 0) {
        if (mode != NOISE) { cadm_set_error("rollout: this module holds noise mode %d, the launch needs %d", NOISE, mode); return CADM_EINVAL; }
        return wt_launch_noise<G, NOISE>(ctx, a, rows_per_member, waves, s);
    } else {
        if (mode == CADM_NOISE_NONE) return wt_launch_noise<G, CADM_NOISE_NONE>(ctx, a, rows_per_member, waves, s);
        if (mode == CADM_NOISE_INJECT) return wt_launch_noise<G, CADM_NOISE_INJECT>(ctx, a, rows_per_member, waves, s);
        return wt_launch_noise<G, CADM_NOISE_PHILOX>(ctx, a, rows_per_member, waves, s);
    }
}

}  // namespace
