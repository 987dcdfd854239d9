// Host launcher of the fused rollout kernel: fills the argument block and dispatches to the
// per-env translation units (rollout_<env>.hip -> rollout_dispatch.h -> rollout_xdl.h).
#include "rollout_args.h"

int cadm_rollout_env_halfcheetah(cadm_ctx*, const RolloutArgs&, int, hipStream_t);
int cadm_rollout_env_ant(cadm_ctx*, const RolloutArgs&, int, hipStream_t);
int cadm_rollout_env_slim_humanoid(cadm_ctx*, const RolloutArgs&, int, hipStream_t);
int cadm_rollout_env_cartpole(cadm_ctx*, const RolloutArgs&, int, hipStream_t);
int cadm_rollout_env_pendulum(cadm_ctx*, const RolloutArgs&, int, hipStream_t);

int cadm_launch_rollout(cadm_ctx* ctx, const float* obs, const float* obs_rows, const float* ctx_vec,
                        const float* actions, const float* eps, int norm_actions, uint32_t seed,
                        uint32_t call, int it, int cand_offset, int n_global, int m, int n_local,
                        float* returns_rows, float* traj_out, hipStream_t s, int dry_run, int force_deterministic) {
    RolloutArgs a{};
    const size_t xbytes = (size_t)ctx->xg.member_frags() * CADM_XDL_FRAG_BYTES * ctx->E;
    if (xbytes >= (1ull << 31)) {
        cadm_set_error("rollout: weight stream of %zu bytes exceeds the 2 GiB buffer-descriptor range", xbytes);
        return CADM_EINVAL;
    }
    a.xw = ctx->xw;
    a.xw_bytes = (unsigned)xbytes;
    a.xw_member_b = (unsigned)((size_t)ctx->xg.member_frags() * CADM_XDL_FRAG_BYTES);
    for (int w = 0, off = 0; w < CADM_XDL_WAVES; ++w) { a.xw_wave_b[w] = (unsigned)off; off += ctx->xg.wave_frags(w) * CADM_XDL_FRAG_BYTES; }
    a.xb = ctx->xb;
    a.xw1 = ctx->xw1;
    a.xw1_member_b = (unsigned)((size_t)ctx->xg1.member_frags() * CADM_XDL_FRAG_BYTES);
    a.xb_member = (size_t)ctx->xg.bias_tiles() * 256;
    a.obs = obs; a.obs_rows = obs_rows; a.ctx_vec = ctx_vec; a.actions = actions; a.eps = eps;
    a.obs_mean = ctx->st.obs_mean; a.obs_std = ctx->st.obs_std;
    a.act_mean = ctx->st.act_mean; a.act_std = ctx->st.act_std;
    a.delta_mean = ctx->st.delta_mean; a.delta_std = ctx->st.delta_std;
    a.maxlv = ctx->ff_maxlv; a.minlv = ctx->ff_minlv;
    a.returns_rows = returns_rows; a.traj = traj_out;
    a.m = m; a.n_local = n_local; a.n_global = n_global; a.cand_offset = cand_offset;
    a.E = ctx->E; a.p = ctx->p; a.PE = ctx->p / ctx->E; a.H = ctx->H; a.NH = ctx->NH;
    a.it = it; a.quirks = ctx->cfg.reference_quirks; a.deterministic = force_deterministic >= 0 ? force_deterministic : ctx->cfg.deterministic;   // (>= 0: cadm_rollout_check asks about a mode)
    a.norm_actions = norm_actions; a.seed = seed; a.call = call;
    a.tbuf = ctx->tbuf;
    const long long rows_total = (long long)m * n_global * ctx->p;
    if (rows_total * ctx->H * ctx->A >= (1ll << 31) || rows_total * ctx->D * ctx->H >= (1ll << 31)) {
        cadm_set_error("rollout: problem too large for 32-bit row indexing (m*n*p = %lld)", rows_total);
        return CADM_EINVAL;
    }
    a.dry_run = dry_run;
    const int rpm = m * n_local * a.PE;
    if (ctx->dev_rollout) return ctx->dev_rollout(ctx, a, rpm, s);      // developer library only (common.h)
    const int mode = a.deterministic ? 2 : a.eps ? 1 : 0;               // CADM_NOISE_*
    if (ctx->jit_rollout[mode]) return ctx->jit_rollout[mode](ctx, &a, rpm, (void*)s);      // side module for this geometry
    switch (ctx->cfg.env_kind) {
        case CADM_ENV_HALFCHEETAH: return cadm_rollout_env_halfcheetah(ctx, a, rpm, s);
        case CADM_ENV_ANT: return cadm_rollout_env_ant(ctx, a, rpm, s);
        case CADM_ENV_SLIM_HUMANOID: return cadm_rollout_env_slim_humanoid(ctx, a, rpm, s);
        case CADM_ENV_CARTPOLE: return cadm_rollout_env_cartpole(ctx, a, rpm, s);
        case CADM_ENV_PENDULUM: return cadm_rollout_env_pendulum(ctx, a, rpm, s);
    }
    cadm_set_error("rollout: unknown env kind %d", ctx->cfg.env_kind);
    return CADM_EINVAL;
}

// 1 if the library carries a kernel for the ctx's geometry (rollout_dispatch.h: HIDS x CTXS, 4 hidden layers, swish)
int cadm_rollout_builtin_env(cadm_ctx* ctx) {
    static const int hids[] = {CADM_HID_LIST};
    static const int ctxs[] = {CADM_CTX_LIST};
    if (ctx->NH != 4 || ctx->cfg.hidden_act != CADM_ACT_SWISH) return 0;
    bool h = false, c = false;
    for (int v : hids) h = h || v == ctx->xg.HID;      // the kernel's width (narrow nets run zero-padded on 128: xdl_geo.h)
    for (int v : ctxs) c = c || v == ctx->C;
    return h && c ? 1 : 0;
}
