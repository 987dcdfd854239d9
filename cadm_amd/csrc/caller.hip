// Caller-side state of the planner kept on the device: the samplers' CEM warm start
// (reference cadm/samplers/sampler.py:118-120) and history ring buffer that produces cp_obs / cp_act
// (sampler.py:165-178, reset :196-200; samplers/utils.py:96-109), so a vectorised simulator can drive
// get_action without host round trips for this bookkeeping.
#include "common.h"

__global__ void warm_start_kernel(const float* __restrict__ plan, int m, int H, int A, float* __restrict__ prev_sol,
                                  float* __restrict__ action) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * H * A) return;
    const int a = i % A, t = (i / A) % H, mi = i / (H * A);
    prev_sol[i] = t + 1 < H ? plan[((size_t)mi * H + t + 1) * A + a] : 0.0f;   // shift left, zero the tail
    if (t == 0) action[mi * A + a] = plan[((size_t)mi * H) * A + a];            // act with the first step
}

// one block per env
__global__ void history_update_kernel(const float* __restrict__ obs, const float* __restrict__ next_obs,
                                      const float* __restrict__ act, const int32_t* __restrict__ done, int D, int A, int Hh,
                                      int H, int state_diff, int32_t* __restrict__ counts, float* __restrict__ hist_obs,
                                      float* __restrict__ hist_act, float* __restrict__ prev_sol) {
    const int mi = blockIdx.x;
    const int cnt = counts[mi];
    float* ho = hist_obs + (size_t)mi * D * Hh;
    float* ha = hist_act + (size_t)mi * A * Hh;
    const bool is_done = done && done[mi] != 0;
    __syncthreads();
    if (is_done) {                                               // sampler.py:193-200
        for (int i = threadIdx.x; i < D * Hh; i += blockDim.x) ho[i] = 0.0f;
        for (int i = threadIdx.x; i < A * Hh; i += blockDim.x) ha[i] = 0.0f;
        if (prev_sol) for (int i = threadIdx.x; i < H * A; i += blockDim.x) prev_sol[(size_t)mi * H * A + i] = 0.0f;   // reset_cem
        if (threadIdx.x == 0) counts[mi] = 0;
        return;
    }
    if (cnt >= Hh) {                                             // full: shift left by one entry (:172-178)
        // strided in-place shift: every thread moves its own columns of every entry, oldest first
        for (int c = threadIdx.x; c < D; c += blockDim.x)
            for (int s = 0; s + 1 < Hh; ++s) ho[s * D + c] = ho[(s + 1) * D + c];
        for (int c = threadIdx.x; c < A; c += blockDim.x)
            for (int s = 0; s + 1 < Hh; ++s) ha[s * A + c] = ha[(s + 1) * A + c];
    }
    const int slot = cnt < Hh ? cnt : Hh - 1;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float o = obs[(size_t)mi * D + c];
        ho[slot * D + c] = state_diff ? next_obs[(size_t)mi * D + c] - o : o;   // :166-169
    }
    for (int c = threadIdx.x; c < A; c += blockDim.x) ha[slot * A + c] = act[(size_t)mi * A + c];
    if (threadIdx.x == 0) counts[mi] = cnt + 1;                  // :202
}

extern "C" int cadm_warm_start_shift(cadm_ctx* ctx, const float* plan, int m, float* prev_sol_io, float* action_out,
                                     void* stream) {
    CADM_REQUIRE(ctx && plan && prev_sol_io && action_out && m > 0, "cadm_warm_start_shift: bad arguments");
    CADM_ON_DEVICE(ctx);
    const int total = m * ctx->H * ctx->A;
    hipLaunchKernelGGL(warm_start_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, plan, m, ctx->H, ctx->A,
                       prev_sol_io, action_out);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

extern "C" int cadm_history_update(cadm_ctx* ctx, const float* obs, const float* next_obs, const float* action,
                                   const int32_t* done, int m, int state_diff, int32_t* counts_io, float* hist_obs_io,
                                   float* hist_act_io, float* prev_sol_io, void* stream) {
    CADM_REQUIRE(ctx && obs && next_obs && action && counts_io && hist_obs_io && hist_act_io && m > 0,
                 "cadm_history_update: bad arguments");
    CADM_ON_DEVICE(ctx);
    CADM_REQUIRE(ctx->cfg.history_length > 0, "cadm_history_update: model has no history window");
    hipLaunchKernelGGL(history_update_kernel, dim3(m), dim3(64), 0, (hipStream_t)stream, obs, next_obs, action, done, ctx->D,
                       ctx->A, ctx->cfg.history_length, ctx->H, state_diff, counts_io, hist_obs_io, hist_act_io, prev_sol_io);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}
