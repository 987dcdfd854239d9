#pragma once
// Fused trajectory-sampling rollout on the f16 matrix pipe with fp32-equivalent accuracy ("xdl" kernel).
// ONE launch advances every (candidate, particle) row through the whole horizon -- input assembly, the 6-matmul
// ensemble MLP, Gaussian head, state update and reward accumulation (reference core/utils.py:431-472).
//
// Why not v_mfma_f32_16x16x4_f32: on gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate (1/16 of the f16 rate) and
// a wave cannot issue any VALU work in its shadow (tools/issue_bench: +12 cycles per MFMA<->VALU switch, +4 per VALU op),
// while v_mfma_f32_16x16x32_f16 hides ~5 single-issue instructions per 32 cycles.  Every fp32 operand is split in two
// f16 numbers (xdl_geo.h): 3 f16 MFMAs per 16x16x32 block with fp32 accumulation reproduce the fp32 product to 2^-22.
//
// Mapping:
//   * workgroup = 4 waves (one per SIMD, 512 VGPRs each) = 16 rows of ONE ensemble member; a workgroup walks over
//     row tiles grp, grp + wgs_per_member, ..  of its member (one tile at BASELINE cfg2);
//   * a layer is evaluated transposed, OUT^T = W^T IN^T: weights are the A operand (streamed from L2 in consumption
//     order through a register ring), the 16 rows are the B / D columns.  A lane's D fragments of tiles (2c, 2c+1) are
//     its B fragment of chunk c of the next layer: activations cross layers through LDS with lane-linear accesses;
//   * wave w owns BASE + (w < EXTRA) hidden tiles and computes them two at a time over the whole K (the B operand of
//     a layer sits in registers), so a tile pair's epilogue (bias, swish, f16 split, LDS store) runs in the shadow of
//     the next pair's MFMAs;
//   * the rollout state lives in registers of the 256 "feature threads" exactly as in the fp32 kernel.
#include "rollout_args.h"
#include "rollout_env.h"
#include "xdl_geo.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#ifndef CADM_XDL_RING
#define CADM_XDL_RING 8
#endif

template <int ENV_, int C_, int HID_>
struct XC {
    static constexpr int ENV = ENV_, C = C_, HID = HID_;
    static constexpr int D = env_D(ENV), A = env_A(ENV), P = env_P(ENV);
    static constexpr int K0 = P + A + C;
    static constexpr int NC0 = (K0 + 31) / 32;            // chunks of layer 0
    static constexpr int NT = (HID + 15) / 16;            // hidden tiles
    static constexpr int NCH = (NT + 1) / 2;              // chunks of a layer that consumes a hidden layer
    static constexpr int NTO = (D + 7) / 8;               // head tiles (8 dims: mu | lv)
    static constexpr int BASE = NT / 4, EXTRA = NT % 4;
    static constexpr int NTOW = (NTO + 3) / 4;            // head tile slots per wave
    static constexpr int R = CADM_XDL_RING;               // ring depth (fragments)
    static constexpr int NP = (D + 1) / 2, NPI = (NP + 15) / 16, NAI = (A + 15) / 16;
    static_assert(BASE >= 1, "hidden width too small for the 4-wave tile split");
    // LDS carve (bytes)
    static constexpr int XIN = 0;                                  // [2 parts][NC0][64 lanes] x 16 B
    static constexpr int ACTA = XIN + 2 * NC0 * 1024;              // [2][NCH][64] x 16 B
    static constexpr int ACTB = ACTA + 2 * NCH * 1024;
    static constexpr int OFULL = ACTB + 2 * NCH * 1024;            // [NTO][64] x float4
    static constexpr int STATS = OFULL + NTO * 1024;               // floats
    static constexpr int ST_OBS_MEAN = 0, ST_OBS_DEN = P, ST_ACT_MEAN = 2 * P, ST_ACT_DEN = 2 * P + A;
    static constexpr int CTRL = STATS + rup((2 * P + 2 * A) * 4, 16);   // + 16 * H floats (dynamic)
};

template <class G>
struct XRing {
    uintx4 w[G::R][2];
};

template <int SLOT, class G>
__device__ __forceinline__ void xring_load(XRing<G>& ring, __amdgpu_buffer_rsrc_t rsrc, unsigned soff, int lane) {
    ring.w[SLOT][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, soff, 0);
    ring.w[SLOT][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + 1024, soff, 0);
}

__device__ __forceinline__ floatx4 xmfma(uintx4 a, f16x8 b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), b, c, 0, 0, 0);
}

// split an fp32 value for the f16 pipe: hi = f16(v), lo = f16((v - hi) * 2^11)
__device__ __forceinline__ void xsplit(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)fmaf((float)hi, -2048.0f, v * 2048.0f);
}

// One hidden-type layer sweep of this wave: NTW tiles x NCHL chunks, tiles two at a time.
//   ring holds fragments 0..R-1 of this layer on entry and 0..R-1 of the NEXT layer (nx_nf of them exist) on exit.
//   epi(ti, hi, lo) consumes a finished tile.
template <class G, int NTW, int NCHL, class Epi>
__device__ __forceinline__ void xdl_sweep(XRing<G>& ring, __amdgpu_buffer_rsrc_t rsrc, unsigned wcur, unsigned wnext,
                                          int nx_nf, const unsigned char* lds_in, int lane, Epi&& epi) {
    constexpr int R = G::R, NF = NTW * NCHL, NFPAD = rup(NF, R);
    f16x8 X1[NCHL], X2[NCHL];
#pragma unroll
    for (int c = 0; c < NCHL; ++c) {
        X1[c] = *reinterpret_cast<const f16x8*>(lds_in + ((0 * NCHL + c) * 64 + lane) * 16);
        X2[c] = *reinterpret_cast<const f16x8*>(lds_in + ((1 * NCHL + c) * 64 + lane) * 16);
    }
    auto prefetch = [&](auto jc) {      // after time slot j: refill its ring slot
        constexpr int j = decltype(jc)::value;
        constexpr int jj = j + R;
        if constexpr (jj < NF) {
            xring_load<j % R>(ring, rsrc, wcur + jj * CADM_XDL_FRAG_BYTES, lane);
        } else if constexpr (jj >= NFPAD) {
            if (jj - NFPAD < nx_nf) xring_load<j % R>(ring, rsrc, wnext + (jj - NFPAD) * CADM_XDL_FRAG_BYTES, lane);
        }
    };
    constexpr int NG = (NTW + 1) / 2;
    static_for(std::make_integer_sequence<int, NG>{}, [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int gs = (NTW - 2 * g) < 2 ? (NTW - 2 * g) : 2;
        floatx4 hi[gs], lo[gs];
#pragma unroll
        for (int k = 0; k < gs; ++k) { hi[k] = floatx4{0.f, 0.f, 0.f, 0.f}; lo[k] = floatx4{0.f, 0.f, 0.f, 0.f}; }
        static_for(std::make_integer_sequence<int, NCHL>{}, [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int j0 = 2 * g * NCHL + c * gs;
#pragma unroll
            for (int k = 0; k < gs; ++k) hi[k] = xmfma(ring.w[(j0 + k) % R][0], X1[c], hi[k]);
#pragma unroll
            for (int k = 0; k < gs; ++k) lo[k] = xmfma(ring.w[(j0 + k) % R][1], X1[c], lo[k]);
#pragma unroll
            for (int k = 0; k < gs; ++k) lo[k] = xmfma(ring.w[(j0 + k) % R][0], X2[c], lo[k]);
            static_for(std::make_integer_sequence<int, gs>{}, [&](auto kc) {
                prefetch(std::integral_constant<int, j0 + decltype(kc)::value>{});
            });
        });
#pragma unroll
        for (int k = 0; k < gs; ++k) epi(2 * g + k, hi[k], lo[k]);
    });
    static_for(std::make_integer_sequence<int, NFPAD - NF>{}, [&](auto jc) {
        prefetch(std::integral_constant<int, NF + decltype(jc)::value>{});
    });
}

template <class G, int NOISE>
__global__ __launch_bounds__(256) void rollout_xdl_kernel(const RolloutArgs a) {
    constexpr int D = G::D, A = G::A, P = G::P, C = G::C, K0 = G::K0, NC0 = G::NC0, NCH = G::NCH, NTO = G::NTO;
    constexpr int NP = G::NP, NPI = G::NPI, NAI = G::NAI, ENV = G::ENV, R = G::R;
    extern __shared__ __attribute__((aligned(16))) unsigned char xsmem[];
    float* stats = reinterpret_cast<float*>(xsmem + G::STATS);
    float* ctrl_s = reinterpret_cast<float*>(xsmem + G::CTRL);
    float* ofull = reinterpret_cast<float*>(xsmem + G::OFULL);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = blockIdx.x / a.wgs_per_member;
    const int grp = blockIdx.x % a.wgs_per_member;
    const int H = a.H;
    const int arow = tid & 15, fg = tid >> 4;
    const int ntiles = (a.rows_per_member + 15) / 16;

    // ---- once per workgroup: stats, zero padding of the operand buffers ----
    for (int i = tid; i < P; i += 256) {
        stats[G::ST_OBS_MEAN + i] = a.obs_mean[i];
        stats[G::ST_OBS_DEN + i] = 1.0f / (a.obs_std[i] + 1e-10f);
    }
    for (int i = tid; i < A; i += 256) {
        stats[G::ST_ACT_MEAN + i] = a.act_mean[i];
        stats[G::ST_ACT_DEN + i] = 1.0f / (a.act_std[i] + 1e-10f);
    }
    for (int i = tid; i < (G::OFULL - G::XIN) / 16; i += 256) reinterpret_cast<uintx4*>(xsmem + G::XIN)[i] = uintx4{0u, 0u, 0u, 0u};

    float st_dmean[NPI][2], st_dden[NPI][2], st_dl2s[NPI][2], st_mx[NPI][2], st_mn[NPI][2];
    int fx_off[NPI][2][2], fx_op[NPI][2][2];
    float fx_mean[NPI][2][2], fx_inv[NPI][2][2];
    // byte offset (part 0) of input feature f of row arow inside x_in
    auto xin_off = [&](int f) { return ((f >> 5) * 64 + ((f & 31) >> 3) * 16 + arow) * 16 + (f & 7) * 2; };
#pragma unroll
    for (int pi = 0; pi < NPI; ++pi) {
        const int dp = fg + 16 * pi;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = 2 * dp + h;
            const int dc = d < D ? d : 0;
            st_dmean[pi][h] = a.delta_mean[dc];
            st_dden[pi][h] = a.delta_std[dc] + 1e-10f;
            st_dl2s[pi][h] = 2.0f * logf(a.delta_std[dc]);              // core/utils.py:360
            st_mx[pi][h] = a.maxlv[dc];
            st_mn[pi][h] = a.minlv[dc];
            int ff[2], fop[2];
            const int nf = d < D ? dim_feats<ENV>(d, ff, fop) : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool on = i < nf;
                const int f = on ? ff[i] : 0;
                fx_off[pi][h][i] = on ? xin_off(f) : -1;
                fx_op[pi][h][i] = on ? fop[i] : 0;
                fx_mean[pi][h][i] = a.obs_mean[f];
                fx_inv[pi][h][i] = 1.0f / (a.obs_std[f] + 1e-10f);
            }
        }
    }
    const int a0 = (fg - (NP & 15) + 16) & 15;                          // first action feature of this thread
    auto put_x = [&](int off, float v) {                                // both split parts of one input feature
        v = fminf(fmaxf(v, -65000.0f), 65000.0f);                       // f16 range (only diverged rows ever get here)
        _Float16 h1, h2;
        xsplit(v, h1, h2);
        *reinterpret_cast<_Float16*>(xsmem + G::XIN + off) = h1;
        *reinterpret_cast<_Float16*>(xsmem + G::XIN + NC0 * 1024 + off) = h2;
    };

    // ---- weight stream of this (member, wave) ----
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xw, 0, a.xw_bytes, 0x00020000);
    const unsigned wbase = e * a.xw_member_b + a.xw_wave_b[wave];
    const float* xb = a.xb + (size_t)e * a.xb_member;
    const int my_ntw = G::BASE + (wave < G::EXTRA ? 1 : 0);
    const int tstart = wave * G::BASE + (wave < G::EXTRA ? wave : G::EXTRA);
    int nhead = 0;
#pragma unroll
    for (int s = 0; s < G::NTOW; ++s) nhead += ((3 - wave) + 4 * s) < NTO ? 1 : 0;
    const int l0_nf = my_ntw * NC0, lh_nf = my_ntw * NCH, hd_nf = nhead * NCH;
    const unsigned w_l0 = wbase, w_h1 = w_l0 + l0_nf * CADM_XDL_FRAG_BYTES;
    const unsigned w_hd = w_h1 + (a.NH - 1) * lh_nf * CADM_XDL_FRAG_BYTES;

    XRing<G> ring;
    static_for(std::make_integer_sequence<int, R>{}, [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (s < l0_nf) xring_load<s>(ring, rsrc, w_l0 + s * CADM_XDL_FRAG_BYTES, lane);
    });

    for (int tile = grp; tile < ntiles; tile += a.wgs_per_member) {
        // ---- this thread's row ----
        int re = tile * 16 + arow;
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

        float po[NPI][2], pz[NPI][2], areg[NAI];
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = 2 * (fg + 16 * pi) + h;
                const int dc = d < D ? d : 0;
                po[pi][h] = a.obs_rows ? a.obs_rows[(size_t)lr * D + dc] : a.obs[mi * D + dc];   // :432
                pz[pi][h] = 0.0f;
            }
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai) {
            const int ac = a0 + 16 * ai;
            areg[ai] = ac < A ? a.actions[abase + ac] : 0.0f;
        }
        if constexpr (C > 0) {
            for (int f = P + A + fg; f < K0; f += 16) put_x(xin_off(f), a.ctx_vec[ctx_off + f - P - A]);   // static: context (:433-439)
        }
        for (int t = fg; t < H; t += 16) ctrl_s[arow * H + t] = ctrl_term<ENV>(a.actions + abase + t * A, A);
        float ret = 0.0f;
        __syncthreads();

        for (int t = 0; t <= H; ++t) {
            // ===== state update from step t-1's head (:348-365,463-466) + reward (:469-471) + input assembly (:442-460) =====
#pragma unroll
            for (int pi = 0; pi < NPI; ++pi) {
                const int dp = fg + 16 * pi;
                if (dp < NP) {
                    if (t > 0) {
                        const int jt = dp >> 2, lt = (dp & 3) * 16 + arow;
                        const floatx4 v = *reinterpret_cast<const floatx4*>(ofull + (jt * 64 + lt) * 4);   // (mu0, mu1, lv0, lv1)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float delta = v[h] * st_dden[pi][h] + st_dmean[pi][h];              // denormalize, :349
                            if constexpr (NOISE != CADM_NOISE_NONE) {
                                float lv = st_mx[pi][h] - softplus_fast(st_mx[pi][h] - v[2 + h]);   // :356
                                lv = st_mn[pi][h] + softplus_fast(lv - st_mn[pi][h]);               // :357
                                const float sd = __expf((lv + st_dl2s[pi][h]) * 0.5f);              // :360-363
                                delta = delta + pz[pi][h] * sd;                                     // :365
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
                                if (fx_op[pi][h][0] != 0) sincos_cw(po[pi][h], &sn, &cs);
                            }
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                if (fx_off[pi][h][i] >= 0) {
                                    const float pv = fx_op[pi][h][i] == 1 ? sn : fx_op[pi][h][i] == 2 ? cs : po[pi][h];
                                    put_x(fx_off[pi][h][i], (pv - fx_mean[pi][h][i]) * fx_inv[pi][h][i]);   // :450-451
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
            if (t == H) break;
            // Gaussian-head noise of THIS step for this thread's pairs (consumed by the next state phase)
            if constexpr (NOISE == CADM_NOISE_INJECT) {
#pragma unroll
                for (int pi = 0; pi < NPI; ++pi) {
                    const int dp = fg + 16 * pi;
                    if (dp < NP) {
                        const float* epp = a.eps + ((size_t)t * a.m * a.n_local * a.p + lr) * D + 2 * dp;
                        pz[pi][0] = epp[0];
                        pz[pi][1] = (2 * dp + 1 < D) ? epp[1] : 0.0f;
                    }
                }
            } else if constexpr (NOISE == CADM_NOISE_PHILOX) {
#pragma unroll
                for (int pi = 0; pi < NPI; ++pi) {
                    uint32_t pc[4] = {grow, (uint32_t)t, (uint32_t)(fg + 16 * pi), CADM_STREAM_EPS | ((uint32_t)a.it << 8)};
                    uint32_t pk[2] = {a.seed, a.call};
                    philox_rounds<0, 10>(pc, pk);
                    box_muller(u01(pc[0]), u01(pc[1]), pz[pi][0], pz[pi][1]);
                }
            }
            __syncthreads();

            // ================= dense layers =================
            auto layers = [&](auto ntw_c) {
                constexpr int NTW = decltype(ntw_c)::value;
                int act_out = G::ACTA, act_in = G::XIN;      // LDS byte offsets (not pointers: keeps every access a ds_ op)
                // hidden epilogue: bias, swish, f16 split, store as the next layer's B operand
                auto hidden_epi = [&](const float* btiles, int out) {
                    return [=](int ti, floatx4 hi, floatx4 lo) {
                        const int Tg = tstart + ti;
                        const floatx4 b = *reinterpret_cast<const floatx4*>(btiles + (Tg * 64 + lane) * 4);
                        f16x4 h1, h2;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = fmaf(lo[r], 4.8828125e-4f, hi[r] + b[r]);
                            v = fminf(v, 60000.0f);
                            const float hv = swish_f(v);
                            _Float16 x1, x2;
                            xsplit(hv, x1, x2);
                            h1[r] = x1; h2[r] = x2;
                        }
                        unsigned char* dst = xsmem + out + ((Tg >> 1) * 64 + lane) * 16 + (Tg & 1) * 8;
                        *reinterpret_cast<f16x4*>(dst) = h1;
                        *reinterpret_cast<f16x4*>(dst + NCH * 1024) = h2;
                    };
                };
                // layer 0
                xdl_sweep<G, NTW, NC0>(ring, rsrc, w_l0, w_h1, lh_nf, xsmem + act_in, lane, hidden_epi(xb, act_out));
                __syncthreads();
                // hidden layers 1 .. NH-1
                for (int l = 1; l < a.NH; ++l) {
                    act_in = act_out;
                    act_out = (act_in == G::ACTA) ? G::ACTB : G::ACTA;
                    const unsigned wc = w_h1 + (l - 1) * lh_nf * CADM_XDL_FRAG_BYTES;
                    const bool last = l + 1 == a.NH;
                    xdl_sweep<G, NTW, NCH>(ring, rsrc, wc, last ? w_hd : wc + lh_nf * CADM_XDL_FRAG_BYTES, last ? hd_nf : lh_nf,
                                           xsmem + act_in, lane, hidden_epi(xb + (size_t)l * G::NT * 256, act_out));
                    __syncthreads();
                }
                act_in = act_out;
                // ================= output heads (mu | logvar tiles) =================
                {
                    constexpr int HNF = G::NTOW * NCH, HNFPAD = rup(HNF, R);
                    const float* bo = xb + (size_t)a.NH * G::NT * 256;
                    f16x8 X1[NCH], X2[NCH];
                    if (nhead > 0) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            X1[c] = *reinterpret_cast<const f16x8*>(xsmem + act_in + ((0 * NCH + c) * 64 + lane) * 16);
                            X2[c] = *reinterpret_cast<const f16x8*>(xsmem + act_in + ((1 * NCH + c) * 64 + lane) * 16);
                        }
                    }
                    static_for(std::make_integer_sequence<int, G::NTOW>{}, [&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        const int ht = (3 - wave) + 4 * s;
                        floatx4 hi = floatx4{0.f, 0.f, 0.f, 0.f}, lo = floatx4{0.f, 0.f, 0.f, 0.f};
                        static_for(std::make_integer_sequence<int, NCH>{}, [&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            constexpr int jx = s * NCH + c;
                            if (s < nhead) {
                                hi = xmfma(ring.w[jx % R][0], X1[c], hi);
                                lo = xmfma(ring.w[jx % R][1], X1[c], lo);
                                lo = xmfma(ring.w[jx % R][0], X2[c], lo);
                            }
                            constexpr int jj = jx + R;
                            if (jj < HNF) {
                                if (jj < hd_nf) xring_load<jx % R>(ring, rsrc, w_hd + jj * CADM_XDL_FRAG_BYTES, lane);
                            } else if constexpr (jj >= HNFPAD) {
                                if (jj - HNFPAD < l0_nf) xring_load<jx % R>(ring, rsrc, w_l0 + (jj - HNFPAD) * CADM_XDL_FRAG_BYTES, lane);
                            }
                        });
                        if (s < nhead) {
                            const floatx4 b = *reinterpret_cast<const floatx4*>(bo + (ht * 64 + lane) * 4);
                            floatx4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaf(lo[r], 4.8828125e-4f, hi[r] + b[r]);
                            *reinterpret_cast<floatx4*>(ofull + (ht * 64 + lane) * 4) = v;
                        }
                    });
                    static_for(std::make_integer_sequence<int, HNFPAD - HNF>{}, [&](auto jc) {
                        constexpr int jx = HNF + decltype(jc)::value;
                        constexpr int jj = jx + R;
                        if (jj - HNFPAD < l0_nf) xring_load<jx % R>(ring, rsrc, w_l0 + (jj - HNFPAD) * CADM_XDL_FRAG_BYTES, lane);
                    });
                }
            };
            if (G::EXTRA > 0 && wave < G::EXTRA) layers(std::integral_constant<int, G::BASE + 1>{});
            else layers(std::integral_constant<int, G::BASE>{});
            __syncthreads();
        }

        // ---- a row's return = sum of its threads' reward parts, in fixed slot order ----
        float* ret_s = reinterpret_cast<float*>(xsmem + G::OFULL);
        __syncthreads();
        ret_s[arow * 16 + fg] = ret;
        __syncthreads();
        if (fg == 0 && valid) {
            float r = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) r += ret_s[arow * 16 + i];
            a.returns_rows[lr] = r;
        }
        __syncthreads();
    }
}

template <class G, int NOISE>
int xdl_launch_noise(cadm_ctx* ctx, const RolloutArgs& a, int rows_per_member, hipStream_t s) {
    RolloutArgs args = a;
    const int tiles = (rows_per_member + 15) / 16;
    // one workgroup per CU (512 VGPRs per wave); workgroups walk over their member's row tiles
    int per_member = ctx->n_cus / ctx->E;
    if (per_member < 1) per_member = 1;
    args.wgs_per_member = tiles < per_member ? tiles : per_member;
    args.rows_per_member = rows_per_member;
    const size_t lds = (size_t)G::CTRL + (size_t)16 * a.H * sizeof(float);
    if (lds > 160 * 1024) {
        cadm_set_error("rollout: horizon %d needs %zu B of LDS (> 160 KiB)", a.H, lds);
        return CADM_EINVAL;
    }
    const void* fn = reinterpret_cast<const void*>(&rollout_xdl_kernel<G, NOISE>);
    if (!ctx->attr_done.count(fn)) {
        CADM_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_done.insert(fn);
    }
    hipLaunchKernelGGL((rollout_xdl_kernel<G, NOISE>), dim3(args.wgs_per_member * ctx->E), dim3(256), lds, s, args);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

template <class G>
int xdl_launch(cadm_ctx* ctx, const RolloutArgs& a, int rows_per_member, hipStream_t s) {
    if (a.deterministic) return xdl_launch_noise<G, CADM_NOISE_NONE>(ctx, a, rows_per_member, s);
    if (a.eps) return xdl_launch_noise<G, CADM_NOISE_INJECT>(ctx, a, rows_per_member, s);
    return xdl_launch_noise<G, CADM_NOISE_PHILOX>(ctx, a, rows_per_member, s);
}

}  // namespace
