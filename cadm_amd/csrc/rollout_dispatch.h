#pragma once
// (env kind, context width, hidden width) -> instantiation of the production rollout kernel (rollout_xdl.h).
// The widths compiled into the library are the Makefile's HIDS x CTXS; any other (hidden width, context width) is
// built on demand as a small side module (cadm_amd/jit.py -> csrc/rollout_jit.hip) and registered on the ctx.
#include "rollout_xdl.h"

#ifndef CADM_CTX_LIST
#define CADM_CTX_LIST 0, 10            // 0 = vanilla PE-TS, 10 = the reference default --context_out_dim (run_cadm_pets.py:135)
#endif
#ifndef CADM_HID_LIST
#define CADM_HID_LIST 200              // the reference default --hidden_size (run_cadm_pets.py:129)
#endif
#define CADM_STR2(...) #__VA_ARGS__
#define CADM_STR(...) CADM_STR2(__VA_ARGS__)

namespace {

template <int ENV, int HID, int... CS>
int dispatch_ctx_list(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) {
    int rc = CADM_EINVAL;
    bool hit = false;
    ((ctx->C == CS ? (hit = true, rc = xdl_launch<ENV, CS, HID>(ctx, a, rpm, s), 0) : 0), ...);
    if (!hit) cadm_set_error("rollout: context_out_dim %d not compiled in (built with CTXS = " CADM_STR(CADM_CTX_LIST) ")", ctx->C);
    return rc;
}

template <int ENV, int... HIDS>
int dispatch_hid(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) {
    int rc = CADM_EINVAL;
    bool hit = false;
    ((ctx->HID == HIDS ? (hit = true, rc = dispatch_ctx_list<ENV, HIDS, CADM_CTX_LIST>(ctx, a, rpm, s), 0) : 0), ...);
    if (!hit) cadm_set_error("rollout: hidden width %d not compiled in (built with HIDS = " CADM_STR(CADM_HID_LIST) ")", ctx->HID);
    return rc;
}

}  // namespace

#define CADM_ROLLOUT_ENV(NAME, ENV) \
    int cadm_rollout_env_##NAME(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) { return dispatch_hid<ENV, CADM_HID_LIST>(ctx, a, rpm, s); }
