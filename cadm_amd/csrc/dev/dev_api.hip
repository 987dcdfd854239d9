// Developer entry points, linked ONLY into libcadm_hip_dev.so (never into the product libcadm_hip.so):
//   * the fp32-MFMA rollout kernel of round 1 (dev/rollout_f32.h) as a comparison kernel for the production split-f16 kernel
//     (tests/test_gpu_precision.py, tools/xdl_vs_f32.py, tools/fuzz_rollout.py),
//   * a forced row-tile flavour of the production kernel (tests/test_gpu_rowtiles.py),
//   * the phase-timing buffer of the CADM_PHASE_TIMING build (tools/phase_timing.py).
// Selection is per ctx and explicit (these calls); nothing reads the environment.
#include "../rollout_args.h"
#include "dev_api.h"

int cadm_dev_pack_f32(cadm_ctx* ctx, hipStream_t s);
void cadm_dev_free_f32(cadm_ctx* ctx);
int cadm_rollout_f32_env_halfcheetah(cadm_ctx*, const RolloutArgs&, int, hipStream_t);
int cadm_rollout_f32_env_ant(cadm_ctx*, const RolloutArgs&, int, hipStream_t);
int cadm_rollout_f32_env_slim_humanoid(cadm_ctx*, const RolloutArgs&, int, hipStream_t);
int cadm_rollout_f32_env_cartpole(cadm_ctx*, const RolloutArgs&, int, hipStream_t);
int cadm_rollout_f32_env_pendulum(cadm_ctx*, const RolloutArgs&, int, hipStream_t);

static int rollout_f32(cadm_ctx* ctx, const RolloutArgs& a0, int rpm, hipStream_t s) {
    if (a0.dry_run) return CADM_OK;       // cadm_rollout_check: nothing to validate for the comparison kernel
    RolloutArgs a = a0;
    a.wstream = ctx->wstream;
    a.bstream = ctx->bstream;
    const size_t wbytes = ctx->wstream_member_floats * ctx->E * sizeof(float);
    if (wbytes >= (1ull << 31)) {
        cadm_set_error("fp32 comparison rollout: weight stream of %zu bytes exceeds the 2 GiB buffer-descriptor range", wbytes);
        return CADM_EINVAL;
    }
    a.wbytes = (unsigned)wbytes;
    a.wmember_b = (unsigned)(ctx->wstream_member_floats * sizeof(float));
    a.w_l0_b = (unsigned)(ctx->g0.layer_floats() * sizeof(float));
    a.w_lh_b = (unsigned)(ctx->gh.layer_floats() * sizeof(float));
    a.w_lo_b = (unsigned)(ctx->go.layer_floats() * sizeof(float));
    a.bmember = ctx->bstream_member_floats;
    a.b_l0 = ctx->g0.bias_floats();
    a.b_lh = ctx->gh.bias_floats();
    switch (ctx->cfg.env_kind) {
        case CADM_ENV_HALFCHEETAH: return cadm_rollout_f32_env_halfcheetah(ctx, a, rpm, s);
        case CADM_ENV_ANT: return cadm_rollout_f32_env_ant(ctx, a, rpm, s);
        case CADM_ENV_SLIM_HUMANOID: return cadm_rollout_f32_env_slim_humanoid(ctx, a, rpm, s);
        case CADM_ENV_CARTPOLE: return cadm_rollout_f32_env_cartpole(ctx, a, rpm, s);
        case CADM_ENV_PENDULUM: return cadm_rollout_f32_env_pendulum(ctx, a, rpm, s);
    }
    return CADM_EINVAL;
}

extern "C" int cadm_dev_set_rollout(cadm_ctx* ctx, int kind, int row_tiles) {
    CADM_REQUIRE(ctx, "cadm_dev_set_rollout: null ctx");
    CADM_REQUIRE(kind == CADM_DEV_ROLLOUT_XDL || kind == CADM_DEV_ROLLOUT_F32, "cadm_dev_set_rollout: unknown kind %d", kind);
    CADM_REQUIRE(row_tiles >= 0 && row_tiles <= 4, "cadm_dev_set_rollout: row_tiles must be 0 (launcher's choice), 1 or 2 (cooperative kernel), 3 / 4 (wave-tile kernel, 8 / 4 tiles per workgroup)");
    ctx->dev_force_mt = row_tiles;
    if (kind == CADM_DEV_ROLLOUT_F32) {
        ctx->dev_rollout = rollout_f32;
        ctx->dev_pack = cadm_dev_pack_f32;
        ctx->dev_free = cadm_dev_free_f32;
        ctx->packed = false;            // the fp32 stream is packed with the next repack / rollout
    } else {
        ctx->dev_rollout = nullptr;
        ctx->dev_pack = nullptr;
    }
    return CADM_OK;
}

extern "C" int cadm_dev_rollout_plan(int units, int two_tile_ok, int wave_tile_ok, int* count_out) {
    CADM_REQUIRE(units >= 0 && count_out, "cadm_dev_rollout_plan: bad argument");
    int count[4];
    xdl_plan_units(units, two_tile_ok != 0, wave_tile_ok != 0, count);
    for (int o = 0; o < 4; ++o) count_out[o] = count[o];
    return CADM_OK;
}

extern "C" int cadm_dev_rollout_plan_for(int units, int two_tile_ok, int wave_tile_ok, int env_kind, int hid, int* count_out, float* costs_out) {
    CADM_REQUIRE(units >= 0 && count_out, "cadm_dev_rollout_plan_for: bad argument");
    int count[4];
    const XdlCosts c = xdl_costs(env_kind, hid);
    xdl_plan_units(units, two_tile_ok != 0, wave_tile_ok != 0, count, c);
    for (int o = 0; o < 4; ++o) { count_out[o] = count[o]; if (costs_out) costs_out[o] = c.c[o]; }
    if (costs_out) for (int k = 0; k < 3; ++k) costs_out[4 + k] = c.wt8p[k];      // (a wave-tile-8 round covering 5 / 6 / 7 units)
    return CADM_OK;
}

extern "C" int cadm_dev_set_timing_buffer(cadm_ctx* ctx, void* dev_u64_buf) {
    CADM_REQUIRE(ctx, "cadm_dev_set_timing_buffer: null ctx");
    ctx->tbuf = (unsigned long long*)dev_u64_buf;
    return CADM_OK;
}

extern "C" int cadm_dev_read_adam_moment(cadm_ctx* ctx, int net, int layer, int is_bias, int second, float* dst, long n_floats,
                                         void* stream) {
    CADM_REQUIRE(ctx && dst, "cadm_dev_read_adam_moment: null argument");
    CADM_ON_DEVICE(ctx);
    float *m = nullptr, *v = nullptr;
    size_t n = 0;
    const int rc = cadm_train_adam_slot(ctx, net, layer, is_bias, &m, &v, &n);
    CADM_REQUIRE(rc == CADM_OK, "cadm_dev_read_adam_moment: no Adam slot for net %d layer %d (%s) -- cadm_train_configure first", net, layer,
                 is_bias ? "bias" : "weight");
    CADM_REQUIRE((long)n == n_floats, "cadm_dev_read_adam_moment: tensor has %zu elements, caller expects %ld", n, n_floats);
    CADM_CHECK_HIP(hipMemcpyAsync(dst, second ? v : m, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CADM_OK;
}

int cadm_launch_refit(cadm_ctx* ctx, const float* cand_returns, const float* rows, int G, int n_local, const float* actions,
                      int m, const float* mean_in, const float* var_in, float* mean_out, float* var_out, int32_t* elites_out,
                      float* plan_out, hipStream_t stream, const RefitRegen* regen);

extern "C" int cadm_dev_refit_sharded(cadm_ctx* ctx, const float* payload, int G, int n_local, int m, int my_rank, float* mean_io, float* var_io,
                                      uint32_t seed, uint32_t call, int it, float* plan_out, void* stream) {
    CADM_REQUIRE(ctx && payload && mean_io && var_io && G > 0 && n_local > 0 && m > 0 && my_rank >= 0 && my_rank < G, "cadm_dev_refit_sharded: bad arguments");
    CADM_ON_DEVICE(ctx);
    RefitRegen rg{};
    rg.on = 1; rg.seed = seed; rg.call = call; rg.it = it; rg.gstride = m * n_local + 1; rg.my_rank = my_rank;
    return cadm_launch_refit(ctx, payload, nullptr, G, n_local, nullptr, m, mean_io, var_io, mean_io, var_io, nullptr, plan_out, (hipStream_t)stream, &rg);
}

extern "C" int cadm_dev_input_checksum(cadm_ctx* ctx, const float* obs, const float* cp_obs, const float* cp_act, const float* mean,
                                       const float* var, int m, float* word_out, void* stream) {
    CADM_REQUIRE(ctx && obs && mean && var && word_out && m > 0, "cadm_dev_input_checksum: bad arguments");
    CADM_ON_DEVICE(ctx);
    return cadm_launch_input_checksum(ctx, obs, cp_obs, cp_act, mean, var, m, reinterpret_cast<unsigned*>(word_out), (hipStream_t)stream);
}

extern "C" int cadm_dev_set_train_flavour(cadm_ctx* ctx, int flavour) {
    // + 16: large-batch forward path (work items spread over all XCDs, one launch per net), + 32: member-affine, joint launch;
    // + 64: one-pass context backward, + 128: one pass per dynamics net (0: the launcher's rules)
    const int waves = flavour & 15, map = (flavour >> 4) & 3, merge = (flavour >> 6) & 3;
    CADM_REQUIRE(ctx && flavour >= 0 && flavour < 256 && (waves == 0 || waves == 4 || waves == 8) && map <= 2 && merge <= 2,
                 "cadm_dev_set_train_flavour: 0, 4 or 8 waves (+ 16 / + 32: forward path, + 64 / + 128: context backward)");
    ctx->train_force_nw = waves;
    ctx->train_force_spread = map;
    ctx->train_force_merge = merge;
    return CADM_OK;
}
