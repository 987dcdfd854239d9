/* Developer C ABI of libcadm_hip_dev.so (= the product objects + dev/): NOT part of the product interface
 * (include/cadm_hip.h) and absent from libcadm_hip.so. */
#ifndef CADM_DEV_API_H
#define CADM_DEV_API_H
#include "../../../include/cadm_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#define CADM_DEV_ROLLOUT_XDL 0   /* the production split-f16 kernel (rollout_xdl.h) */
#define CADM_DEV_ROLLOUT_F32 1   /* round 1's fp32-MFMA kernel (dev/rollout_f32.h): comparison only */
/* kind: which kernel cadm_rollout_returns / the planners launch on this ctx; row_tiles: 0 = the launcher's plan,
 * 1 / 2 = ONE launch of the cooperative kernel with that many 16-row tiles per workgroup, 3 / 4 = of the wave-tile kernel
 * (8 / 4 tiles per workgroup). */
int cadm_dev_set_rollout(cadm_ctx* ctx, int kind, int row_tiles);
/* device buffer of 8*24 uint64 that the CADM_PHASE_TIMING build (make timing) fills with per-phase s_memtime sums of workgroup 0 */
int cadm_dev_set_timing_buffer(cadm_ctx* ctx, void* dev_u64_buf);
/* Copy one Adam moment buffer (second == 0: first moment m, 1: second moment v) of a trained tensor into dst (device, n_floats =
 * the tensor's element count [E, in, out] / [E, out] / [D]).  layer as in cadm_set_weights; layer -1 / -2 = max / min_logvar
 * (CADM_NET_FF).  With beta1 = 0 the first moment after a step IS that step's gradient (m = 0 m + 1 g, exact): the tests read
 * dL/dW and dL/db element by element this way (tests/test_gpu_train.py) instead of differencing weights. */
/* The launcher's plan for a member of `units` CU shares of row tiles (csrc/xdl_geo.h: xdl_plan_units): count_out[4] = launches
 * (rounds) of {cooperative one tile, cooperative two tiles, wave-tile 4, wave-tile 8}.  Host logic only: no ctx, no device. */
int cadm_dev_rollout_plan(int units, int two_tile_ok, int wave_tile_ok, int* count_out);
/* the same with the cost table of one instantiation (xdl_geo.h: xdl_costs(env_kind, hid)); costs_out[7] (optional) receives that table:
 * one-tile, two-tile, wave-tile 4, wave-tile 8 (full round), then a wave-tile-8 round that covers only 5 / 6 / 7 units */
int cadm_dev_rollout_plan_for(int units, int two_tile_ok, int wave_tile_ok, int env_kind, int hid, int* count_out, float* costs_out);
/* The sharded planner's refit on a FABRICATED all-gather result (one GPU plays every rank): payload [G][m * n_local + 1] floats -- every
 * rank's candidate means followed by the checksum word of the inputs it was fed (cadm_dev_input_checksum) --, this rank = my_rank; the
 * elites are regenerated from (seed, call, it) as in a sharded cadm_cem_plan; plan_out [m,H,A] optional.  A checksum that differs from
 * my_rank's makes mean / var / plan NaN. */
int cadm_dev_refit_sharded(cadm_ctx* ctx, const float* payload, int G, int n_local, int m, int my_rank, float* mean_io, float* var_io,
                           uint32_t seed, uint32_t call, int it, float* plan_out, void* stream);
/* the checksum word (as stored in a payload) of a call's replicated inputs; cp_obs / cp_act may be NULL */
int cadm_dev_input_checksum(cadm_ctx* ctx, const float* obs, const float* cp_obs, const float* cp_act, const float* mean, const float* var,
                            int m, float* word_out, void* stream);
/* waves per workgroup of the training chain kernel: 8 (latency flavour) / 4 (throughput flavour: three workgroups per CU) / 0 = by the
 * launch's number of work items (the product's rule) */
int cadm_dev_set_train_flavour(cadm_ctx* ctx, int flavour);   /* 0 / 4 / 8 waves (+ 16: large-batch forward path, + 32: member-affine joint launch; + 64 / + 128: one-pass context backward on / off) */
int cadm_dev_read_adam_moment(cadm_ctx* ctx, int net, int layer, int is_bias, int second, float* dst, long n_floats, void* stream);
#ifdef __cplusplus
}
#endif
#endif
