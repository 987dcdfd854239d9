#pragma once
// (env kind, context width, hidden width) -> instantiation of the production rollout kernel (rollout_xdl.h).
// Compiled into the library: the Makefile's HIDS x CTXS at the reference's defaults -- 4 hidden layers
// (`hidden_sizes=(200,)*4`, run_cadm_pets.py:122-123) and swish (`hidden_nonlinearity`, :31).  Any other (hidden width,
// context width, depth, nonlinearity) is built on demand as a side module from rollout_jit.hip (cadm_amd/jit.py) and
// registered on the ctx with cadm_register_rollout.
#include "rollout_xdl.h"

#define CADM_BUILTIN_NH 4
#define CADM_BUILTIN_ACT CADM_ACT_SWISH

namespace {

template <int ENV, int HID, int... CS>
int dispatch_ctx_list(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) {
    int rc = CADM_ENOTBUILT;
    ((ctx->C == CS ? (rc = xdl_launch<ENV, CS, HID, CADM_BUILTIN_NH, CADM_BUILTIN_ACT>(ctx, a, rpm, s), 0) : 0), ...);
    return rc;
}

template <int ENV, int... HIDS>
int dispatch_hid(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) {
    int rc = CADM_ENOTBUILT;
    if (ctx->NH == CADM_BUILTIN_NH && ctx->cfg.hidden_act == CADM_BUILTIN_ACT)
        ((ctx->xg.HID == HIDS ? (rc = dispatch_ctx_list<ENV, HIDS, CADM_CTX_LIST>(ctx, a, rpm, s), 0) : 0), ...);
    if (rc == CADM_ENOTBUILT)
        cadm_set_error("rollout: no kernel for hidden=%d x %d layers, context_out_dim=%d, nonlinearity %d in this build of the library; "
                       "cadm_amd.jit builds it on demand (cadm_register_rollout)", ctx->HID, ctx->NH, ctx->C, ctx->cfg.hidden_act);
    return rc;
}

}  // namespace

#define CADM_ROLLOUT_ENV(NAME, ENV) \
    int cadm_rollout_env_##NAME(cadm_ctx* ctx, const RolloutArgs& a, int rpm, hipStream_t s) { return dispatch_hid<ENV, CADM_HID_LIST>(ctx, a, rpm, s); }
