#pragma once
// Env closures and scalar math shared by the rollout kernels.
#include <type_traits>
#include <utility>

#include "common.h"

// Work item of a workgroup.  Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8) and every XCD has its own L2: with the
// plain item = linear id, every member's workgroups sit on all eight XCDs and each L2 fetches every member's weight stream (28 MB of fabric
// reads per cfg2 launch for 3.4 MB of weights).  XCD x takes the x-th CONTIGUOUS eighth of the member-major item list instead (as the training
// kernels do, train.hip: xcd_spread_item): an L2 then holds the streams of two or three members.  A bijection on [0, gridDim.x).
#ifndef CADM_ROLLOUT_XCD_AFFINE
#define CADM_ROLLOUT_XCD_AFFINE 1
#endif
__device__ __forceinline__ int rollout_item() {
#if CADM_ROLLOUT_XCD_AFFINE
    const int lin = (int)blockIdx.x, total = (int)gridDim.x, x = lin & 7;
    return x * (total >> 3) + (x < (total & 7) ? x : (total & 7)) + (lin >> 3);
#else
    return (int)blockIdx.x;
#endif
}
#ifdef CADM_PHASE_TIMING
#define NPH 24
#define TS_DECL unsigned long long ts_acc[NPH] = {}; unsigned long long ts_last = __builtin_amdgcn_s_memtime(); unsigned long long* ts_tr = nullptr;
#define TS(i) { const unsigned long long ts_now = __builtin_amdgcn_s_memtime(); ts_acc[i] += ts_now - ts_last; ts_last = ts_now; }
#define TS_DUMP if (a.tbuf && blockIdx.x == 0 && lane == 0) { for (int i = 0; i < NPH; ++i) a.tbuf[wave * NPH + i] = ts_acc[i]; }
#define TS_PARAMS , unsigned long long (&ts_acc)[NPH], unsigned long long& ts_last, unsigned long long* ts_tr
#define TS_ARGS , ts_acc, ts_last, ts_tr
// raw time stamps of ONE sweep (tools/sweep_trace.py): ts_tr points at this wave's 16 slots while that sweep runs
// (stamps stay in SGPRs until the sweep ends: reading one back costs an lgkmcnt(0), which would also drain the LDS operand loads in flight)
#define TR_DECL unsigned long long trv[12] = {};
#define TR(i) { trv[i] = __builtin_amdgcn_s_memtime(); }
#define TR_FLUSH { if (ts_tr) { _Pragma("unroll") for (int i_ = 0; i_ < 11; ++i_) ts_tr[i_] = trv[i_]; } }
#define TR_LATE(i) { if (ts_tr) ts_tr[i] = __builtin_amdgcn_s_memtime(); }
#define TR_ON(cond, w) { ts_tr = (a.tbuf && blockIdx.x == 0 && (cond)) ? a.tbuf + 8 * 24 + (w) * 16 : nullptr; }
#else
#define TS_DECL
#define TS(i)
#define TS_DUMP
#define TS_PARAMS
#define TS_ARGS
#define TR(i)
#define TR_DECL
#define TR_FLUSH
#define TR_LATE(i)
#define TR_ON(cond, w)
#endif

namespace {

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

template <int... Js, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Js...>, F&& f) {
    (f(std::integral_constant<int, Js>{}), ...);
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int rup(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------
// env closures, compile-time per env kind.
// The rollout state is tracked per DIM PAIR (2dp, 2dp+1): these tables say which input features
// a dim feeds and which part of the reward a pair contributes.
// ---------------------------------------------------------------------------------------------
// features computed from obs dim d: index f[] in the preprocessed vector and op[] (0 id, 1 sin, 2 cos)
template <int ENV> __device__ __forceinline__ int dim_feats(int d, int (&f)[2], int (&op)[2]) {
    f[0] = f[1] = 0; op[0] = op[1] = 0;
    if constexpr (ENV == CADM_ENV_HALFCHEETAH) {      // half_cheetah_env.py:46-50: [o1, sin o2, cos o2, o3:]
        if (d == 0) return 0;
        if (d == 1) { f[0] = 0; return 1; }
        if (d == 2) { f[0] = 1; op[0] = 1; f[1] = 2; op[1] = 2; return 2; }
        f[0] = d; return 1;
    } else if constexpr (ENV == CADM_ENV_ANT) {       // ant_env.py:52-53: o1:
        if (d == 0) return 0;
        f[0] = d - 1; return 1;
    } else {                                          // identity preproc
        f[0] = d; return 1;
    }
}

// obs_postproc (half_cheetah_env.py:52-56, ant_env.py:55-59: [pred0, obs1: + pred1:]; others obs + pred)
template <int ENV> __device__ __forceinline__ float postproc(int d, float o, float delta) {
    if constexpr (ENV == CADM_ENV_HALFCHEETAH || ENV == CADM_ENV_ANT) return d == 0 ? delta : o + delta;
    else return o + delta;
}

// action term of the reward, state independent: precomputed per (row, t)
template <int ENV> __device__ __forceinline__ float ctrl_term(const float* a, int A) {
    if constexpr (ENV == CADM_ENV_PENDULUM) {     // classic_control.py:214 (gym PendulumEnv.max_torque = 2)
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

// Contribution of dim pair dp = (o0, o1) to the step reward; a row's reward is the sum over its pairs
// (one non-zero contributor for halfcheetah / ant, so their summation order equals the reference's).
// Obs are the PRE-step state, except cartpole whose reward reads the NEXT state.
template <int ENV> __device__ __forceinline__ float reward_part(int dp, float o0, float o1, float ctrl) {
    if constexpr (ENV == CADM_ENV_HALFCHEETAH) {          // half_cheetah_env.py:82-88
        return dp == 0 ? o0 - 0.1f * ctrl : 0.0f;
    } else if constexpr (ENV == CADM_ENV_ANT) {           // ant_env.py:89-98
        return dp == 0 ? ((o0 + (-0.005f * ctrl)) + 0.0f) + 0.05f : 0.0f;
    } else if constexpr (ENV == CADM_ENV_SLIM_HUMANOID) { // slim_humanoid_env.py:95-111: dims 22 and 1
        if (dp == 11) return (16.666666666666668f * o0 - 0.1f * ctrl) - 0.0f;
        if (dp == 0) return (o1 > 1.0f && o1 < 2.0f) ? 5.0f : 0.0f;
        return 0.0f;
    } else if constexpr (ENV == CADM_ENV_CARTPOLE) {      // classic_control.py:154-166: dims 0 and 2
        const float th = 0.20943951023931953f;            // 12 * 2 * pi / 360
        if (dp == 0) return 1.0f - ((o0 > 2.4f ? 1.f : 0.f) + (o0 < -2.4f ? 1.f : 0.f)) * 1.0f;
        if (dp == 1) return -((o0 > th ? 1.f : 0.f) + (o0 < -th ? 1.f : 0.f)) * 1.0f;
        return 0.0f;
    } else {                                              // pendulum, classic_control.py:209-218
        if (dp == 0) {
            const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
            const float theta = atan2f(o1, o0);
            float m = fmodf(theta + PI_F, TWO_PI_F);
            if (m != 0.0f && m < 0.0f) m += TWO_PI_F;     // floormod
            const float tn = m - PI_F;
            return -(tn * tn + 0.001f * ctrl);
        }
        if (dp == 1) return -(0.1f * (o0 * o0));
        return 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// fast scalar math for the per-row head (absolute error ~1e-7 on logvar, see DESIGN.md numerics)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_cw(float x, float* sn, float* cs) {
    const float k = rintf(x * 0.63661977236758134f);                       // x / (pi/2)
    float r = fmaf(-k, 1.5707962513e+00f, x);                              // pi/2 split in three parts
    r = fmaf(-k, 7.5497894159e-08f, r);
    r = fmaf(-k, 5.3903029534e-15f, r);
    const float z = r * r;                                                 // cephes sinf / cosf minimax on [-pi/4, pi/4]
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    pc = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}

__device__ __forceinline__ float softplus_fast(float x) {     // tf.nn.softplus thresholds (+-13.94); the developer library's comparison kernel
    const float ex = __expf(fminf(x, 20.0f));
    const float mid = __builtin_amdgcn_logf(1.0f + ex) * 0.69314718055994531f;   // v_log_f32 is log2
    const float lo = x < -13.942385f ? ex : mid;
    return x > 13.942385f ? x : lo;
}

// Standard deviation of the Gaussian head (core/utils.py:356-363) with the double soft clamp collapsed algebraically:
//     lv1 = max - softplus(max - lv)     <=>  e^lv1 = 1 / (e^-max + e^-lv)
//     lv2 = min + softplus(lv1 - min)    <=>  e^lv2 = e^min + e^lv1
//     sd  = exp((lv2 + 2 log s) / 2)      =   s * sqrt(e^min + 1 / (e^-max + e^-lv))
// one v_exp, one v_rcp, one v_sqrt in a chain of 7 instructions instead of 3 exp + 2 log in a chain of ~30 (the state phase is on the
// critical path of every step of every flavour).  enmax = e^-max, emin = e^min, s = delta std are per-dim constants of the LDS table.
// Both ends saturate as the clamp does: e^-lv -> inf gives e^min, e^-lv -> 0 gives e^max + e^min.  tf.nn.softplus switches to
// x / e^x beyond |x| > 13.94; against this exact form that is < 9e-7 relative in the variance (tests/test_gpu_precision.py sweeps lv
// over +-30 and both thresholds against the oracle's tf_softplus chain).  Envelope: |max_logvar|, |min_logvar| < 80 (e^-max / e^min
// stay normal fp32 numbers; the reference initialises them to 0.5 / -10 and regularises them towards each other).
__device__ __forceinline__ float head_sd(float lv, float enmax, float emin, float s) {
    const float t = __builtin_amdgcn_exp2f(lv * -1.4426950408889634f);             // e^-lv
    return s * __builtin_amdgcn_sqrtf(emin + __builtin_amdgcn_rcpf(enmax + t));
}

// MODE.FP16_OVFL (bit 23 of the wave's MODE register): an f16 RESULT that overflows -- v_cvt_f16_f32, v_cvt_pk_f16_f32, v_fma_mix*_f16 --
// is clamped to +-65504 instead of becoming inf (true infinities stay).  The rollout kernels split every network input and activation
// in two f16 numbers for the matrix pipe; with this bit the split itself saturates: hi = sat(f16(v)), lo = sat(f16(v - hi)), i.e. a value
// beyond the f16 range goes in as at most +-131008 and stays finite (only diverged rows ever get there).  That replaces the explicit
// clamps in front of every conversion (4 v_min per hidden tile epilogue of 30 instructions; a v_med3 per input feature): the kernels
// are bound by their VALU instructions as much as by their MFMAs.  fp32 arithmetic and MFMAs are not affected.  Set once per wave, at
// kernel entry (the register is per wave and does not outlive it).
#ifndef CADM_FP16_SATURATE
#define CADM_FP16_SATURATE 1
#endif
__device__ __forceinline__ void fp16_saturate_on() {
#if CADM_FP16_SATURATE
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);      // hwreg(HW_REG_MODE, 23, 1) = 1
#endif
}

template <int R0, int R1>
__device__ __forceinline__ void philox_rounds(uint32_t (&c)[4], uint32_t (&k)[2]) {
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
}

// Gaussian-head noise of one (global row, step): ONE Philox4x32-10 call yields four words = two Box-Muller pairs = the noise of TWO dim
// pairs.  Pairs dp and dp + 4 share a call (the wave-tile kernel keeps pairs fg, fg + 4, fg + 8, .. of a row in one lane: two calls per
// lane and step instead of three at 18 dims, three instead of six at 45 -- the Philox rounds were a quarter of its state phase):
//   counter word 2 = eps_group(dp) = (dp & 3) | (dp >> 3) << 2,   the pair's words = (2 s, 2 s + 1) with s = eps_sub(dp) = (dp >> 2) & 1.
// oracle/philox.py: eps_normals restates it.
__device__ __forceinline__ uint32_t eps_group(int dp) { return (uint32_t)((dp & 3) | ((dp >> 3) << 2)); }
__device__ __forceinline__ int eps_sub(int dp) { return (dp >> 2) & 1; }

// NOISE: how the Gaussian head's eps is obtained -- compile-time so that no runtime branch (and no
// merged-register s_waitcnt vmcnt(0)) lands inside the software-pipelined MFMA sweeps.
#define CADM_NOISE_PHILOX 0   // drawn on device (production)
#define CADM_NOISE_INJECT 1   // read from the caller's eps tensor (parity tests)
#define CADM_NOISE_NONE 2     // deterministic model (dynamics.py:43): delta = denormalised mu

}  // namespace
