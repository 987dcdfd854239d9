// Fused trajectory-sampling rollout: ONE launch advances every (candidate, particle) row through
// the whole horizon -- input assembly, the 6-matmul ensemble MLP, Gaussian head, state update and
// reward accumulation (reference core/utils.py:431-472; SURVEY.md groups G3..G6).
//
// Mapping (DESIGN.md "rollout kernel"):
//   * workgroup = 4 waves (one per SIMD) = MT tiles of 16 rows of ONE ensemble member;
//   * every dense layer is evaluated transposed, OUT^T = W^T * IN^T, with v_mfma_f32_16x16x4_f32:
//     weights are the A operand (streamed from L2 in pre-packed fragment order, 3-deep register
//     ring), the 16 data rows are the B/D columns.  The D-layout of a layer's output IS the B-layout
//     of the next layer's input, so activations cross layers through LDS with lane-linear
//     ds_write_b128 / ds_read_b128 and no transposes;
//   * the 13 output tiles of a 200-wide layer are split over the 4 waves as 3 full tiles each plus
//     one tile that is K-split (wave w takes k-step w of every chunk); partial sums meet in LDS.
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

template <int... Js, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Js...>, F&& f) {
    (f(std::integral_constant<int, Js>{}), ...);
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int rup(int a, int b) { return (a + b - 1) / b * b; }

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
    static constexpr int PF = 3;                                       // weight ring depth (chunks)
    static constexpr int NFM = cmax(cmax(NF, NFO), 1), NSM = cmax(cmax(NS, NSO), 1);
    static constexpr int DP = D + 1;
    // LDS carve (floats)
    static constexpr int X_IN = 0;
    static constexpr int ACT_A = X_IN + MT * NC0 * 256;
    static constexpr int ACT_B = ACT_A + MT * NT * 256;
    static constexpr int PART_A = ACT_B + MT * NT * 256;
    static constexpr int PART_B = PART_A + MT * cmax(NS, 1) * 4 * 256;
    static constexpr int OPART = PART_B + MT * cmax(NS, 1) * 4 * 256;
    static constexpr int OBS_S = OPART + MT * cmax(NSO, 1) * 4 * 256;
    static constexpr int STATS = OBS_S + rup(MT * 16 * DP, 4);
    static constexpr int ST_OBS_MEAN = STATS, ST_OBS_DEN = ST_OBS_MEAN + P, ST_ACT_MEAN = ST_OBS_DEN + P,
                         ST_ACT_DEN = ST_ACT_MEAN + A, ST_DMEAN = ST_ACT_DEN + A, ST_DDEN = ST_DMEAN + D,
                         ST_DL2S = ST_DDEN + D, ST_MAXLV = ST_DL2S + D, ST_MINLV = ST_MAXLV + D;
    static constexpr int CTRL_S = rup(ST_MINLV + D, 4);                // + MT*16*H floats (dynamic)
};

struct RolloutArgs {
    const float *wstream, *bstream;
    size_t wmember, bmember;          // floats per member
    size_t w_l0, w_lh, w_lo;          // layer stream sizes (floats): L0, hidden, OUT
    size_t b_l0, b_lh;                // bias tile sizes (floats): L0/hidden, (OUT follows)
    const float *obs, *obs_rows, *ctx_vec, *actions, *eps;
    const float *obs_mean, *obs_std, *act_mean, *act_std, *delta_mean, *delta_std, *maxlv, *minlv;
    float *returns_rows, *traj;
    int m, n_local, n_global, cand_offset, E, p, PE, H, NH, it, quirks, deterministic, norm_actions;
    uint32_t seed, call;
    int wgs_per_member, rows_per_member;
};

template <class G>
struct Ring {
    floatx4 f[G::PF][G::NFM];
    float s[G::PF][G::NSM];
};

template <int SLOT, int NFO, int NSO, class G>
__device__ __forceinline__ void ring_load(Ring<G>& ring, const float* __restrict__ chunk, int lane) {
#pragma unroll
    for (int i = 0; i < NFO; ++i)
        ring.f[SLOT][i] = *reinterpret_cast<const floatx4*>(chunk + (i * 64 + lane) * 4);
#pragma unroll
    for (int s = 0; s < NSO; ++s) ring.s[SLOT][s] = chunk[NFO * 256 + s * 64 + lane];
}

// One dense layer's MFMA sweep for this wave.
//   ring holds chunks 0..PF-1 of this layer on entry and chunks 0..PF-1 of the NEXT layer on exit.
//   B operands: chunks [0, NCH-IN_NS) come from lds_in (already activated), the last IN_NS chunks
//   are the producer's split tiles, reconstructed by the caller into bsplit.
template <class G, int NCH, int KSL, int NFO, int NSO, int NX_NCH, int NX_NFO, int NX_NSO, int IN_NS>
__device__ __forceinline__ void mfma_pass(Ring<G>& ring, const float* __restrict__ wcur,
                                          const float* __restrict__ wnext, const float* lds_in,
                                          const floatx4 (&bsplit)[G::MT][cmax(IN_NS, 1)],
                                          floatx4 (&accF)[G::MT][cmax(NFO, 1)],
                                          floatx4 (&accS)[G::MT][cmax(NSO, 1)], int wave, int lane) {
    constexpr int MT = G::MT, PF = G::PF;
    constexpr int NCHPAD = rup(NCH, PF);
    constexpr int SLOTF = (NFO * 4 + NSO) * 64;
    constexpr int NX_SLOTF = (NX_NFO * 4 + NX_NSO) * 64;
    static_for(std::make_integer_sequence<int, NCHPAD>{}, [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int slot = j % PF;
        if constexpr (j < NCH) {
            constexpr int nk = (j == NCH - 1) ? KSL : 4;
            floatx4 b[MT];
            float bsel[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (j < NCH - IN_NS) {
                    b[mt] = *reinterpret_cast<const floatx4*>(lds_in + ((mt * (NCH - IN_NS) + j) * 64 + lane) * 4);
                } else {
                    b[mt] = bsplit[mt][j - (NCH - IN_NS)];
                }
                bsel[mt] = wave == 0 ? b[mt][0] : wave == 1 ? b[mt][1] : wave == 2 ? b[mt][2] : b[mt][3];
            }
#pragma unroll
            for (int r = 0; r < nk; ++r)
#pragma unroll
                for (int i = 0; i < NFO; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        accF[mt][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.f[slot][i][r], b[mt][r],
                                                                           accF[mt][i], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NSO; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    accS[mt][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.s[slot][s], bsel[mt],
                                                                       accS[mt][s], 0, 0, 0);
        }
        constexpr int jj = j + PF;
        if constexpr (jj < NCH) {
            ring_load<slot, NFO, NSO>(ring, wcur + (size_t)jj * SLOTF, lane);
        } else if constexpr (jj >= NCHPAD && (jj - NCHPAD) < NX_NCH) {
            ring_load<slot, NX_NFO, NX_NSO>(ring, wnext + (size_t)(jj - NCHPAD) * NX_SLOTF, lane);
        }
    });
}

template <int N>
__device__ __forceinline__ void zero_acc(floatx4 (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = floatx4{0.f, 0.f, 0.f, 0.f};
}

template <int ENV>
__device__ __forceinline__ float preproc_feature(const float* o, int pf) {
    if constexpr (ENV == CADM_ENV_HALFCHEETAH) {  // half_cheetah_env.py:46-50
        if (pf == 0) return o[1];
        if (pf == 1) return sinf(o[2]);
        if (pf == 2) return cosf(o[2]);
        return o[pf];
    } else if constexpr (ENV == CADM_ENV_ANT) {   // ant_env.py:52-53
        return o[pf + 1];
    } else {
        return o[pf];
    }
}

// action term of the reward, state independent: precomputed per (row, t)
template <int ENV>
__device__ __forceinline__ float ctrl_term(const float* a, int A) {
    if constexpr (ENV == CADM_ENV_PENDULUM) {     // classic_control.py:214 (max_torque = 2)
        const float tq = fminf(fmaxf(a[0], -2.0f), 2.0f);
        return tq * tq;
    } else if constexpr (ENV == CADM_ENV_CARTPOLE) {
        return 0.0f;
    } else {
        float s = 0.0f;
        for (int i = 0; i < A; ++i) s += a[i] * a[i];
        return s;
    }
}

template <int ENV>
__device__ __forceinline__ float reward_of(const float* o, float ctrl) {
    if constexpr (ENV == CADM_ENV_HALFCHEETAH) {          // half_cheetah_env.py:82-88
        return o[0] - 0.1f * ctrl;
    } else if constexpr (ENV == CADM_ENV_ANT) {           // ant_env.py:89-98
        return ((o[0] + (-0.005f * ctrl)) + 0.0f) + 0.05f;
    } else if constexpr (ENV == CADM_ENV_SLIM_HUMANOID) { // slim_humanoid_env.py:95-111
        const float alive = (o[1] > 1.0f && o[1] < 2.0f) ? 5.0f : 0.0f;
        return ((16.666666666666668f * o[22] - 0.1f * ctrl) - 0.0f) + alive;
    } else if constexpr (ENV == CADM_ENV_CARTPOLE) {      // classic_control.py:154-166 (o = NEXT obs)
        const float th = 0.20943951023931953f;            // 12 * 2 * pi / 360
        const float cond = (o[0] > 2.4f ? 1.f : 0.f) + (o[0] < -2.4f ? 1.f : 0.f) +
                           (o[2] > th ? 1.f : 0.f) + (o[2] < -th ? 1.f : 0.f);
        return 1.0f - cond * 1.0f;
    } else {                                              // pendulum, classic_control.py:209-218
        const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
        const float theta = atan2f(o[1], o[0]);
        float r = fmodf(theta + PI_F, TWO_PI_F);
        if (r != 0.0f && r < 0.0f) r += TWO_PI_F;         // floormod
        const float tn = r - PI_F;
        const float cost = tn * tn + 0.1f * (o[2] * o[2]) + 0.001f * ctrl;
        return -cost;
    }
}

template <class G>
__global__ __launch_bounds__(256) void rollout_kernel(const RolloutArgs a) {
    constexpr int MT = G::MT, D = G::D, A = G::A, P = G::P, C = G::C, K0 = G::K0, NC0 = G::NC0;
    constexpr int NT = G::NT, NF = G::NF, NS = G::NS, NFO = G::NFO, NSO = G::NSO, NTO = G::NTO, DP = G::DP;
    constexpr int ENV = G::ENV;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row16 = lane & 15, q = lane >> 4;
    const int e = blockIdx.x / a.wgs_per_member;
    const int grp = blockIdx.x % a.wgs_per_member;
    const int H = a.H;

    float* x_in = smem + G::X_IN;
    float* obs_s = smem + G::OBS_S;
    float* ctrl_s = smem + G::CTRL_S;

    // ---- row bookkeeping for this thread's assembly role: (arow16 = tid&15, fg = tid>>4) ----
    const int arow = tid & 15, fg = tid >> 4;
    auto row_info = [&](int mt, int r16, int& mi, int& nl, int& j, bool& valid) {
        int re = (grp * MT + mt) * 16 + r16;
        valid = re < a.rows_per_member;
        if (!valid) re = a.rows_per_member - 1;
        const int cidx = re / a.PE, jl = re % a.PE;
        mi = cidx / a.n_local;
        nl = cidx % a.n_local;
        j = e * a.PE + jl;
    };

    // ---- prologue: stats, start state, static input features, control costs ----
    for (int i = tid; i < P; i += 256) {
        smem[G::ST_OBS_MEAN + i] = a.obs_mean[i];
        smem[G::ST_OBS_DEN + i] = a.obs_std[i] + 1e-10f;
    }
    for (int i = tid; i < A; i += 256) {
        smem[G::ST_ACT_MEAN + i] = a.act_mean[i];
        smem[G::ST_ACT_DEN + i] = a.act_std[i] + 1e-10f;
    }
    for (int i = tid; i < D; i += 256) {
        smem[G::ST_DMEAN + i] = a.delta_mean[i];
        smem[G::ST_DDEN + i] = a.delta_std[i] + 1e-10f;
        smem[G::ST_DL2S + i] = 2.0f * logf(a.delta_std[i]);   // core/utils.py:360
        smem[G::ST_MAXLV + i] = a.maxlv[i];
        smem[G::ST_MINLV + i] = a.minlv[i];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int mi, nl, j;
        bool valid;
        row_info(mt, arow, mi, nl, j, valid);
        const size_t lr = ((size_t)mi * a.n_local + nl) * a.p + j;
        // start state (core/utils.py:432) or per-row override
        for (int d = fg; d < D; d += 16)
            obs_s[(mt * 16 + arow) * DP + d] = a.obs_rows ? a.obs_rows[lr * D + d] : a.obs[mi * D + d];
        // static features: context (core/utils.py:433-439) and zero padding
        const float* cvec = nullptr;
        if constexpr (C > 0) {
            const int ep = j % a.E;
            size_t off;
            if (!a.quirks) off = ((size_t)(j / a.PE) * a.m + mi) * C;            // own member's context
            else if (a.it & 1) off = ((size_t)mi * a.E + ep) * C;                // Q2: reinterpreted [E,m]->[m,E]
            else off = ((size_t)ep * a.m + mi) * C;                             // Q1: encoder j % E
            cvec = a.ctx_vec + off;
        }
        for (int f = fg; f < NC0 * 16; f += 16) {
            if (f >= P + A) {
                float v = 0.0f;
                if constexpr (C > 0) { if (f < K0) v = cvec[f - P - A]; }
                x_in[((mt * NC0 + (f >> 4)) * 64 + (f & 3) * 16 + arow) * 4 + ((f & 15) >> 2)] = v;
            }
        }
        // per-(row,t) control cost
        const float* arow_p = a.actions + (((size_t)mi * a.n_global + a.cand_offset + nl) * H) * A;
        for (int t = fg; t < H; t += 16) ctrl_s[(mt * 16 + arow) * H + t] = ctrl_term<ENV>(arow_p + (size_t)t * A, A);
    }

    // ---- weight stream pointers of this wave ----
    const float* wmem = a.wstream + (size_t)e * a.wmember;
    const float* bmem = a.bstream + (size_t)e * a.bmember;
    const float* w0 = wmem + (size_t)wave * (a.w_l0 / 4);
    const float* wo = wmem + a.w_l0 + (size_t)(a.NH - 1) * a.w_lh + (size_t)wave * (a.w_lo / 4);
    const float* bo = bmem + a.b_l0 + (size_t)(a.NH - 1) * a.b_lh;

    Ring<G> ring;
    static_for(std::make_integer_sequence<int, G::PF>{}, [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < NC0) ring_load<s, NF, NS>(ring, w0 + (size_t)s * ((NF * 4 + NS) * 64), lane);
    });

    // action feature prefetch registers (raw action value of the NEXT step for this thread's features)
    float areg[MT][NC0];
    auto fetch_actions = [&](int t) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int mi, nl, j;
            bool valid;
            row_info(mt, arow, mi, nl, j, valid);
            const float* ap = a.actions + (((size_t)mi * a.n_global + a.cand_offset + nl) * H + t) * A;
#pragma unroll
            for (int k = 0; k < NC0; ++k) {
                const int f = fg + 16 * k;
                if (f >= P && f < P + A) areg[mt][k] = ap[f - P];
            }
        }
    };
    fetch_actions(0);

    // reward thread state
    float ret = 0.0f;
    __syncthreads();

    for (int t = 0; t < H; ++t) {
        // ================= input assembly (core/utils.py:442-460) + reward of the pre-step state =================
        if (tid < 16 * MT) {
            const float* o = obs_s + tid * DP;
            if constexpr (ENV == CADM_ENV_CARTPOLE) {
                if (t > 0) ret += reward_of<ENV>(o, 0.0f);
            } else {
                ret += reward_of<ENV>(o, ctrl_s[tid * H + t]);            // core/utils.py:469-471
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float* o = obs_s + (mt * 16 + arow) * DP;
#pragma unroll
            for (int k = 0; k < NC0; ++k) {
                const int f = fg + 16 * k;
                if (f < P + A) {
                    float v;
                    if (f < P) {
                        v = (preproc_feature<ENV>(o, f) - smem[G::ST_OBS_MEAN + f]) / smem[G::ST_OBS_DEN + f];
                    } else {
                        v = areg[mt][k];
                        if (a.norm_actions) v = (v - smem[G::ST_ACT_MEAN + f - P]) / smem[G::ST_ACT_DEN + f - P];
                    }
                    x_in[((mt * NC0 + k) * 64 + (f & 3) * 16 + arow) * 4 + ((f & 15) >> 2)] = v;
                }
            }
        }
        if (t + 1 < H) fetch_actions(t + 1);
        __syncthreads();

        // ================= layer 0 =================
        float* act_out = smem + G::ACT_A;
        float* act_in;
        float* part_out = smem + G::PART_A;
        float* part_in;
        floatx4 accF[MT][cmax(NF, 1)], accS[MT][cmax(NS, 1)];
        floatx4 bsplit[MT][cmax(NS, 1)];
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
                    *reinterpret_cast<floatx4*>(act_out + ((mt * (NT - NS) + wave * NF + i) * 64 + lane) * 4) = v;
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    floatx4 v = accS[mt][s];
                    if (wave == 0) v += biasS[s];
                    *reinterpret_cast<floatx4*>(part_out + (((mt * NS + s) * 4 + wave) * 64 + lane) * 4) = v;
                }
            }
        };
        auto rebuild_split = [&]() {   // producer's K-split tiles: sum the 4 partials, activate
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float* pp = part_in + ((mt * NS + s) * 4 * 64 + lane) * 4;
                    floatx4 v = *reinterpret_cast<const floatx4*>(pp);
                    v += *reinterpret_cast<const floatx4*>(pp + 256);
                    v += *reinterpret_cast<const floatx4*>(pp + 512);
                    v += *reinterpret_cast<const floatx4*>(pp + 768);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = swish_f(v[r]);
                    bsplit[mt][s] = v;
                }
        };

        {
            const float* wn = wmem + a.w_l0 + (size_t)wave * (a.w_lh / 4);   // hidden layer 1
            load_bias(bmem);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { zero_acc(accF[mt]); zero_acc(accS[mt]); }
            mfma_pass<G, NC0, G::KSL0, NF, NS, NT, NF, NS, 0>(ring, w0, wn, x_in, bsplit, accF, accS, wave, lane);
            hidden_epilogue();
        }
        __syncthreads();

        // ================= hidden layers 1 .. NH-1 =================
        for (int l = 1; l < a.NH; ++l) {
            act_in = act_out;
            part_in = part_out;
            act_out = (act_in == smem + G::ACT_A) ? smem + G::ACT_B : smem + G::ACT_A;
            part_out = (part_in == smem + G::PART_A) ? smem + G::PART_B : smem + G::PART_A;
            const float* wc = wmem + a.w_l0 + (size_t)(l - 1) * a.w_lh + (size_t)wave * (a.w_lh / 4);
            load_bias(bmem + a.b_l0 + (size_t)(l - 1) * a.b_lh);
            rebuild_split();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { zero_acc(accF[mt]); zero_acc(accS[mt]); }
            if (l + 1 < a.NH) {
                mfma_pass<G, NT, G::KSLH, NF, NS, NT, NF, NS, NS>(ring, wc, wc + a.w_lh, act_in, bsplit, accF,
                                                                 accS, wave, lane);
            } else {
                mfma_pass<G, NT, G::KSLH, NF, NS, NT, NFO, NSO, NS>(ring, wc, wo, act_in, bsplit, accF, accS,
                                                                   wave, lane);
            }
            hidden_epilogue();
            __syncthreads();
        }

        // ================= output heads (mu | logvar tiles) =================
        act_in = act_out;
        part_in = part_out;
        float* opart = smem + G::OPART;
        floatx4 hF[MT][cmax(NFO, 1)], hS[MT][cmax(NSO, 1)];
        floatx4 hbF[cmax(NFO, 1)], hbS[cmax(NSO, 1)];
#pragma unroll
        for (int i = 0; i < NFO; ++i)
            hbF[i] = *reinterpret_cast<const floatx4*>(bo + ((wave * NFO + i) * 64 + lane) * 4);
#pragma unroll
        for (int s = 0; s < NSO; ++s)
            hbS[s] = *reinterpret_cast<const floatx4*>(bo + ((4 * NFO + s) * 64 + lane) * 4);

        // noise for the head tiles this wave finalises: full tiles wave*NFO+i, split tile s with s%4==wave
        constexpr int NOWN = NFO + (NSO > 0 ? 1 : 0);
        float nz[MT][cmax(NOWN, 1)][2];
        const bool own_split = NSO > 0 && wave < NSO;      // split tile index == wave (NSO <= 3)
        if (!a.deterministic) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int mi, nl, j;
                bool valid;
                row_info(mt, row16, mi, nl, j, valid);
                const size_t lr = ((size_t)mi * a.n_local + nl) * a.p + j;
                const uint32_t grow = (uint32_t)(((size_t)mi * a.n_global + a.cand_offset + nl) * a.p + j);
#pragma unroll
                for (int o = 0; o < NOWN; ++o) {
                    const int jt = o < NFO ? wave * NFO + o : 4 * NFO + wave;
                    const int d0 = 8 * jt + 2 * q;
                    nz[mt][o][0] = nz[mt][o][1] = 0.0f;
                    if ((o < NFO || own_split) && d0 < D) {
                        if (a.eps) {
                            const float* ep = a.eps + (((size_t)t * a.m * a.n_local * a.p) + lr) * D + d0;
                            nz[mt][o][0] = ep[0];
                            if (d0 + 1 < D) nz[mt][o][1] = ep[1];
                        } else {
                            uint32_t r4[4];
                            philox4x32_10(grow, (uint32_t)t, (uint32_t)(d0 >> 1),
                                          CADM_STREAM_EPS | ((uint32_t)a.it << 8), a.seed, a.call, r4);
                            box_muller(u01(r4[0]), u01(r4[1]), nz[mt][o][0], nz[mt][o][1]);
                        }
                    }
                }
            }
        }
        rebuild_split();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { zero_acc(hF[mt]); zero_acc(hS[mt]); }
        mfma_pass<G, NT, G::KSLH, NFO, NSO, NC0, NF, NS, NS>(ring, wo, w0, act_in, bsplit, hF, hS, wave, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < NSO; ++s) {
                floatx4 v = hS[mt][s];
                if (wave == 0) v += hbS[s];
                *reinterpret_cast<floatx4*>(opart + (((mt * NSO + s) * 4 + wave) * 64 + lane) * 4) = v;
            }
        __syncthreads();

        // ================= Gaussian head + state update (core/utils.py:348-365, 463-466) =================
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int mi, nl, j;
            bool valid;
            row_info(mt, row16, mi, nl, j, valid);
            const size_t lr = ((size_t)mi * a.n_local + nl) * a.p + j;
#pragma unroll
            for (int o = 0; o < NOWN; ++o) {
                const int jt = o < NFO ? wave * NFO + o : 4 * NFO + wave;
                if (o >= NFO && !own_split) continue;
                floatx4 v;
                if (o < NFO) {
                    v = hF[mt][o < NFO ? o : 0] + hbF[o < NFO ? o : 0];
                } else {
                    const float* pp = opart + ((mt * NSO + wave) * 4 * 64 + lane) * 4;
                    v = *reinterpret_cast<const floatx4*>(pp);
                    v += *reinterpret_cast<const floatx4*>(pp + 256);
                    v += *reinterpret_cast<const floatx4*>(pp + 512);
                    v += *reinterpret_cast<const floatx4*>(pp + 768);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = 8 * jt + 2 * q + h;
                    if (d < D) {
                        const float mu = v[h], lv0 = v[2 + h];
                        float delta = mu * smem[G::ST_DDEN + d] + smem[G::ST_DMEAN + d];   // denormalize, :349
                        if (!a.deterministic) {
                            const float mx = smem[G::ST_MAXLV + d], mn = smem[G::ST_MINLV + d];
                            float lv = mx - tf_softplus(mx - lv0);                        // :356
                            lv = mn + tf_softplus(lv - mn);                               // :357
                            const float sd = expf((lv + smem[G::ST_DL2S + d]) / 2.0f);    // :360-363
                            delta = delta + nz[mt][o][h] * sd;                            // :365
                        }
                        float* op = obs_s + (mt * 16 + row16) * DP + d;
                        float nxt;
                        if constexpr (ENV == CADM_ENV_HALFCHEETAH || ENV == CADM_ENV_ANT) {
                            nxt = (d == 0) ? delta : (*op + delta);   // obs_postproc: [pred0, obs1: + pred1:]
                        } else {
                            nxt = *op + delta;                        // obs + pred
                        }
                        *op = nxt;
                        if (a.traj && valid) a.traj[(((size_t)t * a.m * a.n_local * a.p) + lr) * D + d] = nxt;
                    }
                }
            }
        }
        __syncthreads();
    }

    if (tid < 16 * MT) {
        if constexpr (ENV == CADM_ENV_CARTPOLE) ret += reward_of<ENV>(obs_s + tid * DP, 0.0f);
        int mi, nl, j;
        bool valid;
        row_info(tid >> 4, tid & 15, mi, nl, j, valid);
        if (valid) a.returns_rows[((size_t)mi * a.n_local + nl) * a.p + j] = ret;
    }
}

template <class G>
int launch(cadm_ctx* ctx, const RolloutArgs& a, int rows_per_member, hipStream_t s) {
    RolloutArgs args = a;
    const int tiles = (rows_per_member + 15) / 16;
    args.wgs_per_member = (tiles + G::MT - 1) / G::MT;
    args.rows_per_member = rows_per_member;
    const size_t lds = ((size_t)G::CTRL_S + (size_t)G::MT * 16 * a.H) * sizeof(float);
    if (lds > 160 * 1024) {
        cadm_set_error("rollout: horizon %d needs %zu B of LDS (> 160 KiB)", a.H, lds);
        return CADM_EINVAL;
    }
    static bool attr_set = false;
    if (!attr_set) {
        CADM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_kernel<G>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(rollout_kernel<G>, dim3(args.wgs_per_member * ctx->E), dim3(256), lds, s, args);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

template <int ENV, int HID>
int dispatch_ctx(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) {
    if (ctx->C == 0) return launch<RC<ENV, 0, HID, 1>>(ctx, a, rpm, s);
    if (ctx->C == 10) return launch<RC<ENV, 10, HID, 1>>(ctx, a, rpm, s);
    cadm_set_error("rollout: context_out_dim %d not compiled in (supported: 0, 10)", ctx->C);
    return CADM_EINVAL;
}

}  // namespace

int cadm_launch_rollout(cadm_ctx* ctx, const float* obs, const float* obs_rows, const float* ctx_vec,
                        const float* actions, const float* eps, int norm_actions, uint32_t seed,
                        uint32_t call, int it, int cand_offset, int n_global, int m, int n_local,
                        float* returns_rows, float* traj_out, hipStream_t s) {
    RolloutArgs a{};
    a.wstream = ctx->wstream;
    a.bstream = ctx->bstream;
    a.wmember = ctx->wstream_member_floats;
    a.bmember = ctx->bstream_member_floats;
    a.w_l0 = ctx->g0.layer_floats();
    a.w_lh = ctx->gh.layer_floats();
    a.w_lo = ctx->go.layer_floats();
    a.b_l0 = ctx->g0.bias_floats();
    a.b_lh = ctx->gh.bias_floats();
    a.obs = obs; a.obs_rows = obs_rows; a.ctx_vec = ctx_vec; a.actions = actions; a.eps = eps;
    a.obs_mean = ctx->st.obs_mean; a.obs_std = ctx->st.obs_std;
    a.act_mean = ctx->st.act_mean; a.act_std = ctx->st.act_std;
    a.delta_mean = ctx->st.delta_mean; a.delta_std = ctx->st.delta_std;
    a.maxlv = ctx->ff_maxlv; a.minlv = ctx->ff_minlv;
    a.returns_rows = returns_rows; a.traj = traj_out;
    a.m = m; a.n_local = n_local; a.n_global = n_global; a.cand_offset = cand_offset;
    a.E = ctx->E; a.p = ctx->p; a.PE = ctx->p / ctx->E; a.H = ctx->H; a.NH = ctx->NH;
    a.it = it; a.quirks = ctx->cfg.reference_quirks; a.deterministic = ctx->cfg.deterministic;
    a.norm_actions = norm_actions; a.seed = seed; a.call = call;
    const int rpm = m * n_local * a.PE;
    if (ctx->HID != 200) {
        cadm_set_error("rollout: hidden width %d not compiled in (supported: 200)", ctx->HID);
        return CADM_EINVAL;
    }
    switch (ctx->cfg.env_kind) {
        case CADM_ENV_HALFCHEETAH: return dispatch_ctx<CADM_ENV_HALFCHEETAH, 200>(ctx, a, rpm, s);
        case CADM_ENV_ANT: return dispatch_ctx<CADM_ENV_ANT, 200>(ctx, a, rpm, s);
        case CADM_ENV_SLIM_HUMANOID: return dispatch_ctx<CADM_ENV_SLIM_HUMANOID, 200>(ctx, a, rpm, s);
        case CADM_ENV_CARTPOLE: return dispatch_ctx<CADM_ENV_CARTPOLE, 200>(ctx, a, rpm, s);
        case CADM_ENV_PENDULUM: return dispatch_ctx<CADM_ENV_PENDULUM, 200>(ctx, a, rpm, s);
    }
    cadm_set_error("rollout: unknown env kind %d", ctx->cfg.env_kind);
    return CADM_EINVAL;
}
