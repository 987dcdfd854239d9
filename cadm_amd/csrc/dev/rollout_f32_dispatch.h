#pragma once
// Instantiations of the fp32-MFMA comparison kernel over the same (hidden width, context width) lists as the product.
#include "rollout_f32.h"

#ifndef CADM_CTX_LIST
#define CADM_CTX_LIST 0, 10
#endif
#ifndef CADM_HID_LIST
#define CADM_HID_LIST 200
#endif

namespace {

template <int ENV, int HID, int... CS>
int f32_dispatch_ctx(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) {
    int rc = CADM_EINVAL;
    bool hit = false;
    ((ctx->C == CS ? (hit = true, rc = launch<RC<ENV, CS, HID, 1>>(ctx, a, rpm, s), 0) : 0), ...);
    if (!hit) cadm_set_error("fp32 comparison rollout: context_out_dim %d not compiled in", ctx->C);
    return rc;
}

template <int ENV, int... HIDS>
int f32_dispatch_hid(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) {
    int rc = CADM_EINVAL;
    bool hit = false;
    ((ctx->HID == HIDS ? (hit = true, rc = f32_dispatch_ctx<ENV, HIDS, CADM_CTX_LIST>(ctx, a, rpm, s), 0) : 0), ...);
    if (!hit) cadm_set_error("fp32 comparison rollout: hidden width %d not compiled in", ctx->HID);
    return rc;
}

}  // namespace

#define CADM_ROLLOUT_F32_ENV(NAME, ENV) \
    int cadm_rollout_f32_env_##NAME(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) { return f32_dispatch_hid<ENV, CADM_HID_LIST>(ctx, a, rpm, s); }
