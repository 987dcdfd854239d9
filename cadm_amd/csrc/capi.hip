// extern "C" entry points of libcadm_hip.so (see include/cadm_hip.h for the contract).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "common.h"

int cadm_launch_clip(const float* in, float* out, int total, float lo, float hi, int do_clip, hipStream_t s);
int cadm_launch_refit(cadm_ctx* ctx, const float* cand_returns, const float* rows, int G, int n_local, const float* actions,
                      int m, const float* mean_in, const float* var_in, float* mean_out, float* var_out, int32_t* elites_out,
                      float* plan_out, hipStream_t stream);

int cadm_launch_refit_sample(cadm_ctx* ctx, const float* cand_returns, const float* rows, int G, int n_local, float* actions, int m,
                             const float* mean_in, const float* var_in, float* mean_out, float* var_out, uint32_t seed, uint32_t call,
                             int next_it, hipStream_t stream);
bool cadm_refit_sample_ok(const cadm_ctx* ctx, int n);

static thread_local char g_err[1024] = "";

void cadm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cadm_last_error(void) { return g_err; }
extern "C" int cadm_abi_version(void) { return CADM_ABI_VERSION; }

extern "C" int cadm_ctx_create(const cadm_config* cfg, cadm_ctx** out) {
    CADM_REQUIRE(cfg && out, "cadm_ctx_create: null argument");
    CADM_REQUIRE(cfg->abi_version == CADM_ABI_VERSION, "cadm_ctx_create: ABI version %d != %d", cfg->abi_version,
                 CADM_ABI_VERSION);
    CADM_REQUIRE(cfg->env_kind >= 0 && cfg->env_kind <= CADM_ENV_PENDULUM, "cadm_ctx_create: unknown env kind %d",
                 cfg->env_kind);
    CADM_REQUIRE(cfg->obs_dim == env_D(cfg->env_kind) && cfg->act_dim == env_A(cfg->env_kind) &&
                     cfg->proc_obs_dim == env_P(cfg->env_kind),
                 "cadm_ctx_create: dims (D=%d,A=%d,P=%d) do not match env kind %d (D=%d,A=%d,P=%d)", cfg->obs_dim,
                 cfg->act_dim, cfg->proc_obs_dim, cfg->env_kind, env_D(cfg->env_kind), env_A(cfg->env_kind),
                 env_P(cfg->env_kind));
    CADM_REQUIRE(cfg->ensemble_size >= 1 && cfg->n_particles >= 1 && cfg->n_particles % cfg->ensemble_size == 0,
                 "cadm_ctx_create: n_particles (%d) must be a positive multiple of ensemble_size (%d)", cfg->n_particles,
                 cfg->ensemble_size);
    CADM_REQUIRE(cfg->n_hidden >= 2 && cfg->n_hidden <= CADM_MAX_HIDDEN_LAYERS,
                 "cadm_ctx_create: n_hidden %d unsupported (2..%d)", cfg->n_hidden, CADM_MAX_HIDDEN_LAYERS);
    CADM_REQUIRE(cfg->hidden >= 16, "cadm_ctx_create: hidden width %d too small", cfg->hidden);
    CADM_REQUIRE(cfg->horizon >= 1, "cadm_ctx_create: horizon must be >= 1");
    CADM_REQUIRE(cfg->context_dim >= 0, "cadm_ctx_create: negative context_dim");
    CADM_REQUIRE(cfg->n_cp_hidden >= 0 && cfg->n_cp_hidden <= CADM_MAX_CP_LAYERS, "cadm_ctx_create: bad n_cp_hidden");
    CADM_REQUIRE(cfg->num_elites >= 1 && cfg->num_cem_iters >= 1, "cadm_ctx_create: bad CEM constants");
    CADM_REQUIRE(cfg->hidden_act >= CADM_ACT_SWISH && cfg->hidden_act <= CADM_ACT_NONE, "cadm_ctx_create: unknown hidden_act %d", cfg->hidden_act);

    cadm_ctx* c = new (std::nothrow) cadm_ctx();
    if (!c) { cadm_set_error("cadm_ctx_create: out of host memory"); return CADM_ENOMEM; }
    c->cfg = *cfg;
    c->set_error = &cadm_set_error;
    CADM_CHECK_HIP(hipGetDevice(&c->device));
    c->D = cfg->obs_dim; c->A = cfg->act_dim; c->P = cfg->proc_obs_dim; c->C = cfg->context_dim;
    c->E = cfg->ensemble_size; c->p = cfg->n_particles; c->H = cfg->horizon; c->HID = cfg->hidden;
    c->NH = cfg->n_hidden; c->K0 = c->P + c->A + c->C;
    c->ff.resize(c->NH + 2);
    c->back.resize(c->NH + 2);
    c->cp.resize(cfg->n_cp_hidden + 1);
    c->xg = make_xdl_geo(c->K0, c->HID, c->D, c->NH);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
    }
    hipError_t e4 = hipMalloc(&c->xw, (size_t)c->xg.member_frags() * CADM_XDL_FRAG_BYTES * c->E);
    hipError_t e5 = hipMalloc(&c->xb, (size_t)c->xg.bias_tiles() * 256 * sizeof(float) * c->E);
    hipError_t e6 = hipMalloc(&c->xflag, sizeof(int));
    const size_t nst = 2 * (size_t)c->P + 2 * c->A + 4 * (size_t)c->D + 2 * (size_t)(c->D + c->A) * cfg->history_length;
    hipError_t e3 = hipMalloc(&c->st.buf, nst * sizeof(float));
    if (e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess || e6 != hipSuccess) {
        const hipError_t bad = e3 != hipSuccess ? e3 : e4 != hipSuccess ? e4 : e5 != hipSuccess ? e5 : e6;
        cadm_set_error("cadm_ctx_create: hipMalloc failed (%s)", hipGetErrorString(bad));
        cadm_ctx_destroy(c);
        return CADM_ENOMEM;
    }
    float* q = c->st.buf;
    const int Hh = cfg->history_length;
    c->st.obs_mean = q; q += c->P;  c->st.obs_std = q; q += c->P;
    c->st.act_mean = q; q += c->A;  c->st.act_std = q; q += c->A;
    c->st.delta_mean = q; q += c->D; c->st.delta_std = q; q += c->D;
    c->st.cp_obs_mean = q; q += c->D * Hh; c->st.cp_obs_std = q; q += c->D * Hh;
    c->st.cp_act_mean = q; q += c->A * Hh; c->st.cp_act_std = q; q += c->A * Hh;
    c->st.back_delta_mean = q; q += c->D; c->st.back_delta_std = q; q += c->D;
    *out = c;
    return CADM_OK;
}

extern "C" int cadm_ctx_destroy(cadm_ctx* ctx) {
    if (!ctx) return CADM_OK;
    cadm_train_free(ctx);
    cadm_dist_destroy(ctx);
    if (ctx->dev_free) ctx->dev_free(ctx);
    if (ctx->xw) (void)hipFree(ctx->xw);
    if (ctx->xb) (void)hipFree(ctx->xb);
    if (ctx->xflag) (void)hipFree(ctx->xflag);
    if (ctx->st.buf) (void)hipFree(ctx->st.buf);
    if (ctx->cp_scratch) (void)hipFree(ctx->cp_scratch);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->prof_ag) (void)hipEventDestroy(e);
    delete ctx;
    return CADM_OK;
}

extern "C" int cadm_set_weights(cadm_ctx* ctx, int net, int layer, float* W, float* b) {
    CADM_REQUIRE(ctx && W && b, "cadm_set_weights: null argument");
    std::vector<DenseRef>* v = net == CADM_NET_FF ? &ctx->ff : net == CADM_NET_BACK ? &ctx->back : net == CADM_NET_CTX ? &ctx->cp : nullptr;
    CADM_REQUIRE(v, "cadm_set_weights: unknown net %d", net);
    CADM_REQUIRE(layer >= 0 && layer < (int)v->size(), "cadm_set_weights: layer %d out of range for net %d", layer, net);
    DenseRef& d = (*v)[layer];
    d.W = W; d.b = b;
    ctx->train_packs_stale = true;
    if (net == CADM_NET_CTX) {
        const int ncp = ctx->cfg.n_cp_hidden;
        d.din = layer == 0 ? (ctx->D + ctx->A) * ctx->cfg.history_length : ctx->cfg.cp_hidden[layer - 1];
        d.dout = layer < ncp ? ctx->cfg.cp_hidden[layer] : ctx->C;
    } else {
        d.din = layer == 0 ? ctx->K0 : ctx->HID;
        d.dout = layer < ctx->NH ? ctx->HID : ctx->D;
        if (net == CADM_NET_FF) ctx->packed = false;
    }
    return CADM_OK;
}

extern "C" int cadm_set_logvar_bounds(cadm_ctx* ctx, int net, float* max_logvar, float* min_logvar) {
    CADM_REQUIRE(ctx && max_logvar && min_logvar, "cadm_set_logvar_bounds: null argument");
    if (net == CADM_NET_FF) { ctx->ff_maxlv = max_logvar; ctx->ff_minlv = min_logvar; }
    else if (net == CADM_NET_BACK) { ctx->back_maxlv = max_logvar; ctx->back_minlv = min_logvar; }
    else { cadm_set_error("cadm_set_logvar_bounds: net %d has no logvar bounds", net); return CADM_EINVAL; }
    return CADM_OK;
}

// (Re)pack the planner's weight stream from the registered master weights.
int cadm_pack_streams(cadm_ctx* ctx, hipStream_t s) {
    for (int l = 0; l < ctx->NH + 2; ++l) {
        if (!ctx->ff[l].W || !ctx->ff[l].b) {
            cadm_set_error("cadm_repack: ff_model layer %d has no registered weights", l);
            return CADM_ESTATE;
        }
    }
    int rc = cadm_pack_xdl(ctx, s);
    if (rc) return rc;
    if (ctx->dev_pack && (rc = ctx->dev_pack(ctx, s))) return rc;       // developer library only (common.h)
    ctx->packed = true;
    return CADM_OK;
}

extern "C" int cadm_repack(cadm_ctx* ctx, void* stream) {
    CADM_REQUIRE(ctx, "cadm_repack: null ctx");
    CADM_ON_DEVICE(ctx);
    ctx->train_packs_stale = true;     // the caller announces new master weights: the training chains' copies follow at the next step
    return cadm_pack_streams(ctx, (hipStream_t)stream);
}

extern "C" int cadm_set_norm_stats(cadm_ctx* ctx, const float* const host_stats[12], void* stream) {
    CADM_REQUIRE(ctx && host_stats, "cadm_set_norm_stats: null argument");
    CADM_ON_DEVICE(ctx);
    const int Hh = ctx->cfg.history_length;
    float* dst[12] = {ctx->st.obs_mean, ctx->st.obs_std, ctx->st.act_mean, ctx->st.act_std, ctx->st.delta_mean,
                      ctx->st.delta_std, ctx->st.cp_obs_mean, ctx->st.cp_obs_std, ctx->st.cp_act_mean,
                      ctx->st.cp_act_std, ctx->st.back_delta_mean, ctx->st.back_delta_std};
    const int len[12] = {ctx->P, ctx->P, ctx->A, ctx->A, ctx->D, ctx->D, ctx->D * Hh, ctx->D * Hh,
                         ctx->A * Hh, ctx->A * Hh, ctx->D, ctx->D};
    for (int i = 0; i < 12; ++i) {
        CADM_REQUIRE(host_stats[i], "cadm_set_norm_stats: stat vector %d is null", i);
        // synchronous copy from pageable host memory: the caller's numpy buffers may die right after this call
        CADM_CHECK_HIP(hipMemcpy(dst[i], host_stats[i], (size_t)len[i] * sizeof(float), hipMemcpyHostToDevice));
    }
    (void)stream;
    ctx->st.set = true;
    return CADM_OK;
}

// next start/stop event pair of a profiling list (grown on demand)
static int prof_pair(std::vector<hipEvent_t>& ev, size_t& used, hipEvent_t* e0, hipEvent_t* e1) {
    if (used + 2 > ev.size()) {
        hipEvent_t a, b;
        CADM_CHECK_HIP(hipEventCreate(&a));
        CADM_CHECK_HIP(hipEventCreate(&b));
        ev.push_back(a);
        ev.push_back(b);
    }
    *e0 = ev[used];
    *e1 = ev[used + 1];
    used += 2;
    return CADM_OK;
}

static int prof_sum(std::vector<hipEvent_t>& ev, size_t& used, float* total_ms_out, int* launches_out) {
    float tot = 0.0f;
    for (size_t i = 0; i + 1 < used; i += 2) {
        CADM_CHECK_HIP(hipEventSynchronize(ev[i + 1]));
        float ms = 0.0f;
        CADM_CHECK_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        tot += ms;
    }
    *total_ms_out = tot;
    *launches_out = (int)(used / 2);
    used = 0;
    return CADM_OK;
}

static int require_ready(cadm_ctx* ctx, const char* who) {
    if (!ctx->st.set) { cadm_set_error("%s: normalisation stats not set (call cadm_set_norm_stats)", who); return CADM_ESTATE; }
    if (!ctx->ff_maxlv || !ctx->ff_minlv) { cadm_set_error("%s: logvar bounds not set", who); return CADM_ESTATE; }
    return CADM_OK;
}

extern "C" int cadm_context_forward(cadm_ctx* ctx, const float* cp_obs, const float* cp_act, int m, int bs,
                                    float* ctx_out, void* stream) {
    CADM_REQUIRE(ctx && cp_obs && cp_act && ctx_out && m > 0, "cadm_context_forward: bad arguments");
    CADM_ON_DEVICE(ctx);
    CADM_REQUIRE(ctx->C > 0, "cadm_context_forward: model has no context encoder");
    if (!ctx->st.set) { cadm_set_error("cadm_context_forward: normalisation stats not set"); return CADM_ESTATE; }
    return cadm_launch_context(ctx, cp_obs, cp_act, m, bs, ctx_out, (hipStream_t)stream);
}

extern "C" int cadm_rollout_returns(cadm_ctx* ctx, const float* obs, const float* obs_rows, const float* ctx_vec,
                                    const float* actions, const float* eps, int norm_actions, uint32_t seed,
                                    uint32_t call, int it, int cand_offset, int n_global, int m, int n_local,
                                    float* returns_rows, float* traj_out, void* stream) {
    CADM_REQUIRE(ctx && obs && actions && returns_rows, "cadm_rollout_returns: null argument");
    CADM_ON_DEVICE(ctx);
    CADM_REQUIRE(m > 0 && n_local > 0 && cand_offset >= 0 && cand_offset + n_local <= n_global,
                 "cadm_rollout_returns: bad candidate range [%d, %d) of %d", cand_offset, cand_offset + n_local, n_global);
    CADM_REQUIRE(ctx->C == 0 || ctx_vec, "cadm_rollout_returns: ctx_vec required for a context model");
    int rc = require_ready(ctx, "cadm_rollout_returns");
    if (rc) return rc;
    if (!ctx->packed) {
        rc = cadm_pack_streams(ctx, (hipStream_t)stream);
        if (rc) return rc;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->prof) {
        if ((rc = prof_pair(ctx->prof_ev, ctx->prof_used, &e0, &e1))) return rc;
        CADM_CHECK_HIP(hipEventRecord(e0, (hipStream_t)stream));
    }
    rc = cadm_launch_rollout(ctx, obs, obs_rows, ctx_vec, actions, eps, norm_actions, seed, call, it, cand_offset,
                             n_global, m, n_local, returns_rows, traj_out, (hipStream_t)stream);
    if (ctx->prof && rc == CADM_OK) CADM_CHECK_HIP(hipEventRecord(e1, (hipStream_t)stream));
    return rc;
}

int cadm_rollout_builtin_env(cadm_ctx* ctx);     // rollout.hip

extern "C" int cadm_rollout_builtin(cadm_ctx* ctx) {
    if (!ctx) return 0;
    return cadm_rollout_builtin_env(ctx);
}

extern "C" int cadm_register_rollout(cadm_ctx* ctx, int noise_mode, void* fn, const int d[8]) {
    CADM_REQUIRE(ctx && fn && d && noise_mode >= 0 && noise_mode <= 2, "cadm_register_rollout: bad arguments");
    CADM_REQUIRE(d[0] == CADM_CTX_LAYOUT_TAG && d[7] == (int)sizeof(cadm_ctx),
                 "cadm_register_rollout: module built against another library layout (tag %d / %d bytes, library %d / %d): rebuild it",
                 d[0], d[7], CADM_CTX_LAYOUT_TAG, (int)sizeof(cadm_ctx));
    CADM_REQUIRE(d[1] == ctx->cfg.env_kind && d[2] == ctx->C && d[3] == ctx->HID && d[4] == ctx->NH && d[5] == ctx->cfg.hidden_act && d[6] == noise_mode,
                 "cadm_register_rollout: module is for env %d C %d hidden %d x %d act %d noise %d, the ctx needs env %d C %d hidden %d x %d act %d noise %d",
                 d[1], d[2], d[3], d[4], d[5], d[6], ctx->cfg.env_kind, ctx->C, ctx->HID, ctx->NH, ctx->cfg.hidden_act, noise_mode);
    ctx->jit_rollout[noise_mode] = (int (*)(cadm_ctx*, const RolloutArgs*, int, void*))fn;
    return CADM_OK;
}

extern "C" int cadm_rollout_check(cadm_ctx* ctx, int noise_mode, int m, int n_local) {
    CADM_REQUIRE(ctx && noise_mode >= 0 && noise_mode <= 2 && m > 0 && n_local > 0, "cadm_rollout_check: bad arguments");
    CADM_ON_DEVICE(ctx);
    // dummy non-null pointers select the noise mode; nothing is dereferenced or launched
    const float* eps = noise_mode == 1 ? (const float*)ctx->xb : nullptr;
    const int det = ctx->cfg.deterministic;
    ctx->cfg.deterministic = noise_mode == 2;
    const int rc = cadm_launch_rollout(ctx, (const float*)ctx->xb, nullptr, (const float*)ctx->xb, (const float*)ctx->xb, eps, 1, 0, 0, 0, 0, n_local, m,
                                       n_local, (float*)ctx->xb, nullptr, nullptr, 1);
    ctx->cfg.deterministic = det;
    return rc;
}

extern "C" int cadm_profile_enable(cadm_ctx* ctx, int enable) {
    CADM_REQUIRE(ctx, "cadm_profile_enable: null ctx");
    ctx->prof = enable != 0;
    ctx->prof_used = 0;
    ctx->prof_ag_used = 0;
    return CADM_OK;
}

extern "C" int cadm_profile_read(cadm_ctx* ctx, float* total_ms_out, int* launches_out) {
    CADM_REQUIRE(ctx && total_ms_out && launches_out, "cadm_profile_read: null argument");
    return prof_sum(ctx->prof_ev, ctx->prof_used, total_ms_out, launches_out);
}

extern "C" int cadm_profile_read_collective(cadm_ctx* ctx, float* total_ms_out, int* calls_out) {
    CADM_REQUIRE(ctx && total_ms_out && calls_out, "cadm_profile_read_collective: null argument");
    return prof_sum(ctx->prof_ag, ctx->prof_ag_used, total_ms_out, calls_out);
}

// the path's one collective, bracketed by hipEvents when profiling is on
static int allgather_timed(cadm_ctx* ctx, const float* send, float* recv, size_t count, hipStream_t s) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc;
    if (ctx->prof) {
        if ((rc = prof_pair(ctx->prof_ag, ctx->prof_ag_used, &e0, &e1))) return rc;
        CADM_CHECK_HIP(hipEventRecord(e0, s));
    }
    rc = cadm_dist_allgather(ctx, send, recv, count, s);
    if (ctx->prof && rc == CADM_OK) CADM_CHECK_HIP(hipEventRecord(e1, s));
    return rc;
}

// ---------------------------------------------------------------------------------------------
// fused single-GPU planners
// ---------------------------------------------------------------------------------------------
struct PlanWs {
    float *ctxv, *actions, *rows, *cand, *gath, *mean, *var;
    int32_t* raw;
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t carve(cadm_ctx* ctx, int m, int n, char* base, PlanWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    float* ctxv = (float*)take((size_t)ctx->E * m * (ctx->C > 0 ? ctx->C : 1) * 4);
    float* actions = (float*)take((size_t)m * n * ctx->H * ctx->A * 4);
    float* rows = (float*)take((size_t)m * n * ctx->p * 4);
    float* cand = (float*)take((size_t)m * n * 4);
    float* gath = (float*)take((size_t)m * n * 4);
    float* mean = (float*)take((size_t)m * ctx->H * ctx->A * 4);
    float* var = (float*)take((size_t)m * ctx->H * ctx->A * 4);
    int32_t* raw = (int32_t*)take((size_t)m * n * ctx->H * 4);
    if (w) { w->ctxv = ctxv; w->actions = actions; w->rows = rows; w->cand = cand; w->gath = gath; w->mean = mean; w->var = var; w->raw = raw; }
    return off;
}

extern "C" size_t cadm_plan_workspace_bytes(cadm_ctx* ctx, int m, int n) {
    if (!ctx || m <= 0 || n <= 0) return 0;
    return carve(ctx, m, n, nullptr, nullptr);
}

extern "C" int cadm_cem_plan(cadm_ctx* ctx, const float* obs, const float* cp_obs, const float* cp_act,
                             const float* init_mean, const float* init_var, int m, int n, uint32_t seed,
                             uint32_t call, void* workspace, float* plan_out, void* stream) {
    CADM_REQUIRE(ctx && obs && init_mean && init_var && workspace && plan_out && m > 0 && n > 0,
                 "cadm_cem_plan: bad arguments");
    CADM_ON_DEVICE(ctx);
    CADM_REQUIRE(ctx->C == 0 || (cp_obs && cp_act), "cadm_cem_plan: cp_obs/cp_act required for a context model");
    hipStream_t s = (hipStream_t)stream;
    PlanWs w;
    carve(ctx, m, n, (char*)workspace, &w);
    int rc;
    if (ctx->C > 0 && (rc = cadm_context_forward(ctx, cp_obs, cp_act, m, 0, w.ctxv, stream))) return rc;
    const int G = ctx->comm ? ctx->nranks : 1;
    CADM_REQUIRE(n % G == 0, "cadm_cem_plan: n_candidates %d not divisible by %d ranks", n, G);
    const int nl = n / G, off = (ctx->comm ? ctx->rank : 0) * nl;
    const int iters = ctx->cfg.num_cem_iters;
    // Small candidate sets (rank-by-counting regime): the refit of iteration it and the sampling of iteration it + 1 are ONE
    // launch parallel over the plan's (t, a) elements (cem_refit_sample_kernel) -- sample(0), then rollout + fused step per
    // iteration, the last refit alone (it writes the clipped plan).  Larger sets: sample / rollout / refit per iteration.
    const bool fuse = cadm_refit_sample_ok(ctx, n);
    for (int it = 0; it < iters; ++it) {
        const bool first = it == 0, last = it + 1 == iters;
        // iteration 0 reads the caller's mean / var directly; the last refit also writes the clipped plan (dynamics.py:365-366)
        const float* mean_in = first ? init_mean : w.mean;
        const float* var_in = first ? init_var : w.var;
        float* plan = last ? plan_out : nullptr;
        // every rank draws ALL n candidates (counter-based RNG keyed by global candidate id): elites need no exchange
        if (first || !fuse) {
            if ((rc = cadm_sample_actions(ctx, mean_in, var_in, nullptr, seed, call, it, m, n, w.actions, stream))) return rc;
        }
        if ((rc = cadm_rollout_returns(ctx, obs, nullptr, ctx->C > 0 ? w.ctxv : nullptr, w.actions, nullptr, 1, seed,
                                       call, it, off, n, m, nl, w.rows, nullptr, stream))) return rc;
        const float* cand = nullptr;
        const float* rows = w.rows;
        if (G > 1) {   // the one collective of the path: [m, n/G] per rank -> [G, m, n/G] everywhere
            if ((rc = cadm_particle_mean(ctx, w.rows, m, nl, w.cand, stream))) return rc;
            if ((rc = allgather_timed(ctx, w.cand, w.gath, (size_t)m * nl, s))) return rc;
            cand = w.gath;
            rows = nullptr;
        }              // (single rank: the particle mean is taken inside the refit kernels)
        if (fuse && !last) {
            if ((rc = cadm_launch_refit_sample(ctx, cand, rows, G, nl, w.actions, m, mean_in, var_in, w.mean, w.var, seed, call, it + 1, s))) return rc;
        } else {
            if ((rc = cadm_launch_refit(ctx, cand, rows, G, nl, w.actions, m, mean_in, var_in, w.mean, w.var, nullptr, plan, s))) return rc;
        }
    }
    return CADM_OK;
}

// Per-call inputs of a small planner call travel as KERNEL ARGUMENTS: the runtime writes them into the kernarg segment with
// the launch packet, a one-workgroup kernel unpacks them into the device block.  Same queue as the planner kernels: no copy
// engine, no cross-engine dependency (hipMemcpyAsync of 2.5 KB cost ~10 us more per call; a kernel READING the pinned block
// over PCIe was 10x slower still).
#define CADM_INGEST_MAX 960
struct IngestBlock { float v[CADM_INGEST_MAX]; };
__global__ void ingest_kernel(const IngestBlock blk, float* __restrict__ dev_block, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dev_block[i] = blk.v[i];
}

// The class API's call (dynamics.py:344-367: numpy in -> numpy out) as ONE library call: the caller has packed the five
// per-call inputs into a pinned host block; one async H2D copy of that block, the whole planner, the plan written straight
// into the caller's pinned host buffer by the last refit kernel, and (optionally) the stream synchronisation.
extern "C" int cadm_cem_plan_staged(cadm_ctx* ctx, const float* host_block, float* dev_block, const int32_t off[5], int nfloats,
                                    int m, int n, uint32_t seed, uint32_t call, void* workspace, float* plan_out_host, int sync,
                                    void* stream) {
    CADM_REQUIRE(ctx && host_block && dev_block && off && nfloats > 0 && plan_out_host, "cadm_cem_plan_staged: bad arguments");
    CADM_ON_DEVICE(ctx);
    hipStream_t s = (hipStream_t)stream;
    if (nfloats <= CADM_INGEST_MAX) {
        IngestBlock blk;
        memcpy(blk.v, host_block, (size_t)nfloats * sizeof(float));
        hipLaunchKernelGGL(ingest_kernel, dim3(1), dim3(256), 0, s, blk, dev_block, nfloats);
        CADM_CHECK_HIP(hipGetLastError());
    } else {
        CADM_CHECK_HIP(hipMemcpyAsync(dev_block, host_block, (size_t)nfloats * sizeof(float), hipMemcpyHostToDevice, s));
    }
    auto at = [&](int i) -> const float* { return off[i] < 0 ? nullptr : dev_block + off[i]; };      // obs, cp_obs, cp_act, mean, var
    // completion: [m] flag words behind the plan in the caller's pinned buffer, released by the last refit kernel with this
    // call's id; the host polls them (a sleeping hipStreamSynchronize wakes up ~10 us late on a 1 ms call)
    unsigned* flags = reinterpret_cast<unsigned*>(plan_out_host + (size_t)m * ctx->H * ctx->A);
    const unsigned val = call ^ 0x5ca1ab1eu;
    if (sync) { for (int i = 0; i < m; ++i) flags[i] = ~val; }
    ctx->plan_done = sync ? flags : nullptr;
    ctx->plan_done_val = val;
    const int rc = cadm_cem_plan(ctx, at(0), at(1), at(2), at(3), at(4), m, n, seed, call, workspace, plan_out_host, stream);
    ctx->plan_done = nullptr;
    if (rc) return rc;
    if (sync) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        volatile unsigned* vf = flags;
        bool ok = false;
        for (unsigned long spins = 0; !ok; ++spins) {
            ok = true;
            for (int i = 0; i < m; ++i) ok = ok && vf[i] == val;
            if (ok) break;
            __builtin_ia32_pause();
            if ((spins & 4095) == 4095) {           // a kernel that faulted never raises the flag: fall back to the runtime's verdict
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > 0.25) break;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (!ok) CADM_CHECK_HIP(hipStreamSynchronize(s));
    }
    return CADM_OK;
}

__global__ void gather_raw_first_kernel(const int32_t* raw, const int32_t* best, int m, int n, int H, int32_t* out) {
    const int mi = blockIdx.x * blockDim.x + threadIdx.x;
    if (mi < m) out[mi] = raw[((size_t)mi * n + best[mi]) * H];
}

extern "C" int cadm_rs_plan(cadm_ctx* ctx, const float* obs, const float* cp_obs, const float* cp_act, int m, int n,
                            uint32_t seed, uint32_t call, void* workspace, float* action_out, int32_t* raw_best_out,
                            void* stream) {
    CADM_REQUIRE(ctx && obs && workspace && action_out && m > 0 && n > 0, "cadm_rs_plan: bad arguments");
    CADM_ON_DEVICE(ctx);
    CADM_REQUIRE(ctx->C == 0 || (cp_obs && cp_act), "cadm_rs_plan: cp_obs/cp_act required for a context model");
    CADM_REQUIRE(!ctx->cfg.discrete || raw_best_out, "cadm_rs_plan: raw_best_out required for discrete actions");
    hipStream_t s = (hipStream_t)stream;
    PlanWs w;
    carve(ctx, m, n, (char*)workspace, &w);
    int rc;
    if (ctx->C > 0 && (rc = cadm_context_forward(ctx, cp_obs, cp_act, m, 0, w.ctxv, stream))) return rc;
    if ((rc = cadm_sample_uniform(ctx, seed, call, m, n, w.actions, w.raw, stream))) return rc;
    // it = 0: the RS graph transposes the context tensor once (core/utils.py:513) -> the even-iteration layout
    const int G = ctx->comm ? ctx->nranks : 1;
    CADM_REQUIRE(n % G == 0, "cadm_rs_plan: n_candidates %d not divisible by %d ranks", n, G);
    const int nl = n / G, off = (ctx->comm ? ctx->rank : 0) * nl;
    if ((rc = cadm_rollout_returns(ctx, obs, nullptr, ctx->C > 0 ? w.ctxv : nullptr, w.actions, nullptr,
                                   ctx->cfg.discrete ? 0 : 1, seed, call, 0, off, n, m, nl, w.rows, nullptr, stream))) return rc;
    if ((rc = cadm_particle_mean(ctx, w.rows, m, nl, w.cand, stream))) return rc;
    const float* cand = w.cand;
    if (G > 1) {
        if ((rc = allgather_timed(ctx, w.cand, w.gath, (size_t)m * nl, s))) return rc;
        cand = w.gath;
    }
    int32_t* best = (int32_t*)w.mean;  // scratch reuse: m ints
    if ((rc = cadm_rs_select(ctx, cand, G, nl, w.actions, m, action_out, best, stream))) return rc;
    if (ctx->cfg.discrete) {
        hipLaunchKernelGGL(gather_raw_first_kernel, dim3((m + 63) / 64), dim3(64), 0, s, w.raw, best, m, n, ctx->H, raw_best_out);
        CADM_CHECK_HIP(hipGetLastError());
    } else {
        if ((rc = cadm_launch_clip(action_out, action_out, m * ctx->A, ctx->cfg.lower_bound, ctx->cfg.upper_bound, 1, s))) return rc;
    }
    return CADM_OK;
}
