/*
 * cadm_hip.h -- C ABI of libcadm_hip.so: the MI355X (gfx950) implementation of
 * CaDM's CEM-planning / ensemble-training hot path.
 *
 * The reference (younggyoseo/CaDM) has NO FFI layer: its seam is the Python class
 * MLPEnsembleCEMDynamicsModel (cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:12)
 * whose methods call sess.run on one TensorFlow graph (cadm/utils/tensor_utils.py:6-11).
 * Every entry point below replaces one piece of that graph; the reference lines it
 * replaces are cited per function (paths relative to /root/reference).  The Python
 * mirror of the class lives in cadm_amd/dynamics/ and binds these with ctypes
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - every pointer argument is a DEVICE pointer to row-major float32 unless noted;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously;
 *   - every function returns 0 on success or a negative CADM_E* code and never throws;
 *     cadm_last_error() returns a thread-local message for the last failure;
 *   - no global state beyond the opaque cadm_ctx;
 *   - a cadm_ctx is bound to the HIP device that was current at cadm_ctx_create.
 */
#ifndef CADM_HIP_H
#define CADM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CADM_ABI_VERSION 2

#define CADM_OK 0
#define CADM_EINVAL (-1)      /* bad argument / unsupported configuration */
#define CADM_ENOMEM (-2)      /* device allocation failed */
#define CADM_EHIP (-3)        /* HIP runtime error (see cadm_last_error) */
#define CADM_ESTATE (-4)      /* weights / stats not set */
#define CADM_ENOTBUILT (-5)   /* no rollout kernel for this geometry in this build (cadm_amd.jit builds one: cadm_register_rollout) */

/* hidden nonlinearity (the reference's `_activations` table, dynamics.py:17-24; ctor default relu :30, the scripts pass swish) */
#define CADM_ACT_SWISH 0
#define CADM_ACT_RELU 1
#define CADM_ACT_TANH 2
#define CADM_ACT_SIGMOID 3
#define CADM_ACT_NONE 4

/* env kinds: the closures obs_preproc / obs_postproc / tf_reward_fn that the reference
 * compiles into its graph (cadm/envs/<env>.py) are compiled into the kernels */
#define CADM_ENV_HALFCHEETAH 0   /* half_cheetah_env.py:46-59,82-88; cripple: half_cheetah_cripple_env.py:60-73,90-96 */
#define CADM_ENV_ANT 1           /* ant_env.py:52-62,89-98 */
#define CADM_ENV_SLIM_HUMANOID 2 /* slim_humanoid_env.py:39-46,95-111 */
#define CADM_ENV_CARTPOLE 3      /* classic_control.py:94-101,154-166 */
#define CADM_ENV_PENDULUM 4      /* classic_control.py:209-218,284-291 */

#define CADM_NET_FF 0    /* variable scope 'ff_model'        (dynamics.py:159) */
#define CADM_NET_BACK 1  /* variable scope 'backward_model'  (dynamics.py:213) */
#define CADM_NET_CTX 2   /* variable scope 'context_model'   (dynamics.py:140) */

#define CADM_MAX_HIDDEN_LAYERS 8
#define CADM_MAX_CP_LAYERS 8

typedef struct cadm_ctx cadm_ctx;

/* Mirrors the ctor kwargs of MLPEnsembleCEMDynamicsModel (dynamics.py:26-54) that shape the graph. */
typedef struct cadm_config {
    int32_t abi_version;       /* CADM_ABI_VERSION */
    int32_t env_kind;          /* CADM_ENV_* */
    int32_t ensemble_size;     /* E */
    int32_t n_particles;       /* p, p % E == 0 (core/utils.py:447) */
    int32_t obs_dim;           /* D  (must match the env kind) */
    int32_t act_dim;           /* A */
    int32_t proc_obs_dim;      /* P */
    int32_t context_dim;       /* C = context_out_dim, 0 for the vanilla PE-TS model */
    int32_t n_hidden;          /* number of hidden layers (4; 1 .. CADM_MAX_HIDDEN_LAYERS) */
    int32_t hidden;            /* hidden width, all layers equal (200) */
    int32_t horizon;           /* n_forwards H */
    int32_t deterministic;     /* dynamics.py:43 */
    int32_t discrete;          /* discrete action space (cartpole) */
    int32_t reference_quirks;  /* 1: reproduce context-layout quirks Q1/Q2 (core/utils.py:434-435) */
    int32_t history_length;    /* Hh */
    int32_t n_cp_hidden;       /* context encoder hidden layers (3) */
    int32_t cp_hidden[CADM_MAX_CP_LAYERS]; /* (256,128,64) */
    int32_t num_elites;        /* 50  (core/utils.py:391) */
    int32_t num_cem_iters;     /* 5   (core/utils.py:392) */
    float alpha;               /* 0.1 (core/utils.py:393) */
    float lower_bound;         /* -1  (core/utils.py:395) */
    float upper_bound;         /* +1  (core/utils.py:396) */
    int32_t back_model;        /* 1: backward model present (back_coeff > 0, dynamics.py:212) */
    int32_t hidden_act;        /* CADM_ACT_*: hidden_nonlinearity of both dynamics nets (dynamics.py:104; 0 = swish) */
    int32_t reserved[6];
} cadm_config;

const char* cadm_last_error(void);
int cadm_abi_version(void);
/* 16 hex digits: sha256 over the kernel sources (csrc/ *.hip, *.h, include/cadm_hip.h) this binary was compiled from.  Profiling
 * artefacts (profiles/ *pmc*.json) record it, and bench.py only quotes counters whose build id equals the loaded library's. */
const char* cadm_build_id(void);

/* Build / free the per-model device state (replaces graph construction, dynamics.py:107-342). */
int cadm_ctx_create(const cadm_config* cfg, cadm_ctx** out);
int cadm_ctx_destroy(cadm_ctx* ctx);

/* Register one dense layer's master weights (raw TF layout W [E,in,out], b [E,1,out];
 * create_dense_layer, core/utils.py:635-641).  The pointers stay owned by the caller and must
 * outlive the ctx; the planner's MFMA-fragment weight streams are (re)packed from them.
 * layer index: CADM_NET_FF/BACK: 0..n_hidden-1 hidden, n_hidden = output_mu, n_hidden+1 = output_logvar;
 *              CADM_NET_CTX:     0..n_cp_hidden-1 hidden, n_cp_hidden = cp_output. */
int cadm_set_weights(cadm_ctx* ctx, int net, int layer, float* W, float* b);
/* max_logvar / min_logvar [1,D] (core/utils.py:338-339). */
int cadm_set_logvar_bounds(cadm_ctx* ctx, int net, float* max_logvar, float* min_logvar);
/* The caller has written master weights in place (load(), a manual assign; ANY net): re-pack the planner's weight streams
 * now and mark the training chains' packed operand copies stale (rebuilt at the next cadm_train_step / cadm_predict;
 * training steps themselves keep them current).  Writing registered weights without this call leaves both on old values.
 * Fails with CADM_EINVAL if a dynamics weight cannot be split for the f16 matrix pipe: non-finite, or |w| > 65000 AS PACKED -- swish
 * nets carry log2(e) in layer 0 (limit 45052 in the model's units) and 1 / log2(e) in the heads (93780); biases: any finite value.
 * The planner's other range limits (inputs clamped at +-65000, hidden pre-activations at 60000 = 41589 for swish nets in the model's
 * units) are listed in INTEGRATION.md, "Numeric envelope of the planner". */
int cadm_repack(cadm_ctx* ctx, void* stream);

/* The 12 normalisation vectors fed per call in the reference (dynamics.py:344-347,604-645), order:
 * obs_mean[P] obs_std[P] act_mean[A] act_std[A] delta_mean[D] delta_std[D] cp_obs_mean[D*Hh]
 * cp_obs_std[D*Hh] cp_act_mean[A*Hh] cp_act_std[A*Hh] back_delta_mean[D] back_delta_std[D].
 * HOST pointers (float32); copied. */
int cadm_set_norm_stats(cadm_ctx* ctx, const float* const host_stats[12], void* stream);

/* Context encoder inference (core/utils.py:401-406,614-617): cp_obs [m,D*Hh], cp_act [m,A*Hh]
 * -> ctx_out [E,m,C].  With bs != 0 the inputs are already [E,m,.] (get_context_pred,
 * dynamics.py:369-380 / training graph :619-622). */
int cadm_context_forward(cadm_ctx* ctx, const float* cp_obs, const float* cp_act, int m, int bs,
                         float* ctx_out, void* stream);

/* Candidate action sequences (core/utils.py:425-429):
 * actions[m,n_global,H,A] = mean + sqrt(min(min(((mean-lb)/2)^2,((ub-mean)/2)^2),var)) * z.
 * z = injected truncated-normal draws [m,n_global,H,A], or NULL to draw them on device from
 * Philox4x32-10 keyed (seed, call) with counters (element, attempt, STREAM_ACT | it<<8). */
int cadm_sample_actions(cadm_ctx* ctx, const float* mean, const float* var, const float* z,
                        uint32_t seed, uint32_t call, int it, int m, int n_global,
                        float* actions_out, void* stream);
/* The same for candidates [cand_offset, cand_offset + n_local) only, written at their GLOBAL positions of actions_out
 * [m,n_global,H,A]: the draws are keyed by the global element index, so a rank of a candidate-sharded planner that draws only its own
 * shard writes exactly what any other rank would have written there (the other positions are left untouched). */
int cadm_sample_actions_shard(cadm_ctx* ctx, const float* mean, const float* var, const float* z,
                              uint32_t seed, uint32_t call, int it, int m, int n_global, int cand_offset, int n_local,
                              float* actions_out, void* stream);
/* Random-shooting draws (core/utils.py:498-503): U[-1,1) actions (continuous) or one-hot rows of
 * uniform integer actions (discrete; raw ints also written to raw_out [m,n,H] if non-NULL). */
int cadm_sample_uniform(cadm_ctx* ctx, uint32_t seed, uint32_t call, int m, int n_global,
                        float* actions_out, int32_t* raw_out, void* stream);

/* THE hot kernel: trajectory-sampling rollout of candidates [cand_offset, cand_offset+n_local)
 * through the ensemble over the whole horizon (core/utils.py:431-472, one CEM iteration's inner
 * loop; vanilla :138-170).  One launch = H steps x (m*n_local*p) rows.
 *   obs          [m,D]            start state (tiled over n,p as :432)
 *   obs_rows     [m,n_local,p,D]  optional per-row start state (teacher forcing), else NULL
 *   ctx_vec      [E,m,C]          context encoder output (NULL when C == 0)
 *   actions      [m,n_global,H,A] candidate actions; normalised in-kernel (:443) unless
 *                                 norm_actions == 0 (discrete RS feeds one-hot rows, :500)
 *   eps          [H,m,n_local,p,D] injected N(0,1) for the Gaussian head (:365), or NULL to draw
 *                                 from Philox keyed (seed, call), counters (global row, t, d/2,
 *                                 STREAM_EPS | it<<8) -- independent of the candidate sharding
 *   it           CEM iteration (selects the context layout of quirk Q2, :434)
 *   returns_rows [m,n_local,p]    out: per-row returns before the particle mean (:471)
 *   traj_out     [H,m,n_local,p,D] optional out: next observation after every step, else NULL */
int cadm_rollout_returns(cadm_ctx* ctx, const float* obs, const float* obs_rows, const float* ctx_vec,
                         const float* actions, const float* eps, int norm_actions,
                         uint32_t seed, uint32_t call, int it,
                         int cand_offset, int n_global, int m, int n_local,
                         float* returns_rows, float* traj_out, void* stream);

/* Rollout kernels are compile-time specialised per (env kind, hidden width, number of hidden layers, context width,
 * nonlinearity).  The library carries the reference's defaults (4 x {128,200,256,512} swish, context 0 / 10); for anything else
 * (`--hidden_size`, `--context_out_dim`, run_cadm_pets.py:122-135) the caller builds cadm_amd/csrc/rollout_jit.hip with hipcc
 * (cadm_amd/jit.py does) and registers the module's entry point for a noise mode (0 device Philox, 1 injected eps, 2 deterministic).
 *   cadm_rollout_builtin   1 if the ctx's geometry is compiled in, else 0
 *   cadm_register_rollout  fn = the module's `cadm_jit_rollout`; describe = its `cadm_jit_describe` output (checked against the ctx)
 *   cadm_rollout_check     validates that a rollout of m x n_local candidates can be launched (kernel present, LDS fits the
 *                          horizon) WITHOUT launching -- construction-time instead of first-get_action failure */
int cadm_rollout_builtin(cadm_ctx* ctx);
int cadm_register_rollout(cadm_ctx* ctx, int noise_mode, void* fn, const int describe[8]);
int cadm_rollout_check(cadm_ctx* ctx, int noise_mode, int m, int n_local);

/* Mean over particles (core/utils.py:474): returns_rows [m,n_local,p] -> cand_returns [m,n_local]. */
int cadm_particle_mean(cadm_ctx* ctx, const float* returns_rows, int m, int n_local,
                       float* cand_returns, void* stream);

/* Elite refit (core/utils.py:475-486): top-k (sorted, ties -> lower index), gather, mean /
 * biased variance over elites, EMA with alpha.
 *   cand_returns  [G,m,n_local]  the all-gathered per-candidate returns of G ranks (G = 1: [m,n]);
 *                                global candidate ni lives at [ni / n_local][mi][ni % n_local]
 *   actions       [m,G*n_local,H,A]
 *   mean_io/var_io [m,H,A]       updated in place
 *   elites_out    [m,num_elites] optional int32 out (global candidate ids), else NULL */
int cadm_cem_refit(cadm_ctx* ctx, const float* cand_returns, int G, int n_local, const float* actions,
                   int m, float* mean_io, float* var_io, int32_t* elites_out, void* stream);

/* cadm_cem_refit without the candidates' action sequences: the <= num_elites elites' sequences are DRAWN AGAIN from the device RNG by
 * global candidate id -- (seed, call, it) and mean_io / var_io (before the update) are the distribution iteration `it` was sampled
 * from -- so a rank that sampled only its own shard (cadm_sample_actions_shard) refits over the global set without any exchange of
 * actions.  Bit-identical to cadm_cem_refit on the fully sampled buffer. */
int cadm_cem_refit_regen(cadm_ctx* ctx, const float* cand_returns, int G, int n_local, int m, float* mean_io, float* var_io,
                         uint32_t seed, uint32_t call, int it, int32_t* elites_out, void* stream);

/* Random-shooting selection (core/utils.py:554-561): argmax over candidates (first maximum),
 * out[m,A] = actions[mi, best, 0, :].  best_out [m] optional. */
int cadm_rs_select(cadm_ctx* ctx, const float* cand_returns, int G, int n_local, const float* actions,
                   int m, float* first_action_out, int32_t* best_out, void* stream);

/* Whole single-GPU CEM planner = one `sess.run(optimal_action_var)` (dynamics.py:349-356,
 * core/utils.py:398-488): context encoder once, then num_cem_iters x (sample, rollout, refit);
 * the result [m,H,A] is clipped to [lb,ub] as get_action does (dynamics.py:365-366) unless discrete.
 * workspace: device scratch of cadm_plan_workspace_bytes(ctx, m, n) bytes. */
size_t cadm_plan_workspace_bytes(cadm_ctx* ctx, int m, int n);
int cadm_cem_plan(cadm_ctx* ctx, const float* obs, const float* cp_obs, const float* cp_act,
                  const float* init_mean, const float* init_var, int m, int n,
                  uint32_t seed, uint32_t call, void* workspace, float* plan_out, void* stream);
/* The same planner fed from the HOST, as the class's get_action is (numpy in -> numpy out, dynamics.py:344-367): `host_block`
 * is a pinned host buffer of `nfloats` floats holding obs / cp_obs / cp_act / init_mean / init_var at float offsets off[0..4]
 * (-1 = absent); it is copied to `dev_block` (device, >= nfloats floats) with one async copy on `stream`, the planner runs, and
 * the last refit kernel writes the plan into `plan_out_host` (pinned, device-visible host memory: [m,H,A] floats followed by m
 * 32-bit completion flags).  sync != 0: returns when plan_out_host is readable (the flags are polled; no sleep in the runtime). */
int cadm_cem_plan_staged(cadm_ctx* ctx, const float* host_block, float* dev_block, const int32_t off[5], int nfloats,
                         int m, int n, uint32_t seed, uint32_t call, void* workspace, float* plan_out_host, int sync,
                         void* stream);
/* Random-shooting planner (core/utils.py:490-561): out [m,A] (continuous) or raw_best_out [m] ints. */
int cadm_rs_plan(cadm_ctx* ctx, const float* obs, const float* cp_obs, const float* cp_act,
                 int m, int n, uint32_t seed, uint32_t call, void* workspace,
                 float* action_out, int32_t* raw_best_out, void* stream);

/* One training step = sess.run([mse_loss, back_mse_loss, recon_loss, train_op]) (dynamics.py:505-507):
 * forward of context / forward / backward nets on the [E,B,.] bootstrap batch, losses
 * (dynamics.py:269-314), gradients, TF1-semantics Adam (dynamics.py:316-317) applied IN PLACE to the
 * registered master weights.  train != 0 applies the update; train == 0 only evaluates the losses
 * (validation, dynamics.py:531-533).  losses_out: device float[3] = mse, back_mse, recon. */
typedef struct cadm_train_hparams {
    float learning_rate;          /* 1e-3 */
    float beta1, beta2, epsilon;  /* 0.9, 0.999, 1e-8 */
    float back_coeff;             /* dynamics.py:53 */
    float weight_decay_coeff;     /* dynamics.py:45 */
    float weight_decays[CADM_MAX_HIDDEN_LAYERS + 1];      /* hidden_i ..., last = both heads (core/utils.py:319,328,334) */
    float context_weight_decays[CADM_MAX_CP_LAYERS + 1];  /* cp_hidden_i ..., last = cp_output (core/utils.py:601,610) */
} cadm_train_hparams;
int cadm_train_configure(cadm_ctx* ctx, const cadm_train_hparams* hp, int max_batch);
int cadm_train_step(cadm_ctx* ctx, const float* obs, const float* act, const float* delta,
                    const float* obs_next, const float* back_delta, const float* cp_obs,
                    const float* cp_act, int B, int train, float* losses_out, void* stream);
/* The same step on rows of a windowed dataset that stays on the device -- what `fit` (dynamics.py:382-569) feeds after
 * `_preprocess_inputs` (:676-696) without materialising the exploded rows: per-step tensors are [N, F, .] (obs, act,
 * delta, obs_next, back_delta), history tensors [N, .] (cp_obs, cp_act); training row rid is window row_w[rid] at future
 * offset row_f[rid]; the batch is idx[e * idx_ld + b], b < B (row ids, int64): a column slice of the [E, n_train] bootstrap
 * matrix needs no copy. */
int cadm_train_step_rows(cadm_ctx* ctx, const float* ds_obs, const float* ds_act, const float* ds_delta,
                         const float* ds_obs_next, const float* ds_back_delta, const float* ds_cp_obs,
                         const float* ds_cp_act, int F, const long long* row_w, const long long* row_f,
                         const long long* idx, long long idx_ld, int B, int train, float* losses_out, void* stream);
/* One-step prediction heads of every member on an [E,B,.] batch: normalised mean mu [E,B,D] and (optional,
 * probabilistic models) soft-clamped log-variance [E,B,D] -- the vanilla reference's `_get_pred`
 * (mlp_ensemble_cem_dynamics.py:185-189 -> [mlp.mu, mlp.logvar]); backs the public predict(). */
int cadm_predict(cadm_ctx* ctx, const float* obs, const float* act, const float* cp_obs, const float* cp_act,
                 int B, float* mu_out, float* logvar_out, void* stream);
/* Reset Adam moments / step count (a fresh tf.global_variables_initializer()). */
int cadm_train_reset(cadm_ctx* ctx, void* stream);

/* Caller-side planner state on the device (no host round trips between env steps):
 * cadm_warm_start_shift: the samplers' CEM warm start (cadm/samplers/sampler.py:118-120):
 *   prev_sol[:, :-1] = plan[:, 1:]; prev_sol[:, -1] = 0; action = plan[:, 0]     (plan/prev_sol [m,H,A], action [m,A])
 * cadm_history_update: the history ring buffer behind cp_obs / cp_act (sampler.py:165-178,193-202):
 *   per env: entry = (state_diff ? next_obs - obs : obs, action) written at slot count (< Hh) or, once full,
 *   after shifting the window left by one entry; count += 1; done[mi] != 0 instead zeroes the window, the count
 *   and (if given) that env's prev_sol (reset_cem).  done may be NULL.  action is the [m,A] vector fed to the
 *   model (one-hot for discrete envs, sampler.py:148-149). */
int cadm_warm_start_shift(cadm_ctx* ctx, const float* plan, int m, float* prev_sol_io, float* action_out, void* stream);
int cadm_history_update(cadm_ctx* ctx, const float* obs, const float* next_obs, const float* action,
                        const int32_t* done, int m, int state_diff, int32_t* counts_io, float* hist_obs_io,
                        float* hist_act_io, float* prev_sol_io, void* stream);

/* Training-set builder of the context model: device twin of the `context` branch of
 * ModelSampleProcessor.process_samples (cadm/samplers/model_sample_processor.py:58-100).  All paths' steps are concatenated:
 * obs [T,D], act [T,A], cp_obs [T,Dh], cp_act [T,Ah] of 4- or 8-byte elements (elem_bytes; float32 or the reference's
 * float64 -- the kernel only moves words, results are bit-exact); path_off [P+1] = first step of every path (int32);
 * a path of len steps yields n = max(len, F+1) - 1 rows (short paths are zero-padded, :62-68); row_path / row_step [N] name
 * the path and step of every output row (N = sum of n).  Writes concat_obs / concat_next_obs [N,F*D], concat_act [N,F*A],
 * concat_bool [N,F] (1.0 / 0.0 of the element type, including the reference's "row 0 of every path is masked" quirk,
 * :85-86) and the rows' history windows cp_obs_out [N,Dh], cp_act_out [N,Ah].  Needs no ctx; asynchronous on `stream`. */
int cadm_build_windows(const void* obs, const void* act, const void* cp_obs, const void* cp_act, int elem_bytes, int D, int A,
                       int Dh, int Ah, const int32_t* path_off, const int32_t* row_path, const int32_t* row_step, int N, int F,
                       void* concat_obs, void* concat_act, void* concat_next_obs, void* concat_bool, void* cp_obs_out,
                       void* cp_act_out, void* stream);

/* Multi-GPU planning: candidates shard contiguously over the ranks of an RCCL communicator owned by the
 * ctx (one process per GPU).  The reference is single-device (cadm/trainers/mb_trainer.py:103-107); this
 * adds exactly one collective per CEM iteration -- ncclAllGather of the per-candidate returns
 * ([m, n/G] floats per rank) -- between core/utils.py:474 and :475.  RCCL is dlopen'ed on first use.
 *   cadm_dist_unique_id  rank 0: fill a 128-byte ncclUniqueId, to be broadcast by the host to every rank
 *   cadm_dist_init       every rank: ncclCommInitRank on the ctx's device
 * Once initialised, cadm_cem_plan / cadm_rs_plan take n = GLOBAL candidate count (divisible by nranks),
 * roll out candidates [rank*n/G, (rank+1)*n/G) and return the identical plan on every rank.
 * cadm_cem_plan: a rank draws only its own candidates (cadm_sample_actions_shard; the draws are keyed by the global
 * element index) and the refit regenerates the elite sequences (cadm_cem_refit_regen) instead of reading them.  Every
 * rank's payload carries one more float: a checksum word of its replicated inputs (obs, cp_obs, cp_act, mean, var).
 * The refit compares the ranks' words; on a mismatch -- the ranks were fed different inputs -- the plan is NaN on EVERY
 * rank and a flag is raised (cadm_dist_mismatch; a staged call also writes it into m words behind its completion flags).  The
 * call still returns CADM_OK; the Python class raises RuntimeError on the flag. */
int cadm_dist_unique_id(char out_id[128]);
int cadm_dist_init(cadm_ctx* ctx, const char id[128], int nranks, int rank);
int cadm_dist_destroy(cadm_ctx* ctx);
/* The same sharded planner over a collective the HOST supplies (any torch.distributed backend, MPI, ...): every rank registers an
 * all-gather `fn(user, send, recv, count, stream)` -- [count] floats at device pointer `send` of every rank -> [nranks * count] floats at
 * device pointer `recv`, rank-major, valid for work enqueued on `stream` after fn returns; 0 = success.  cadm_cem_plan / cadm_rs_plan
 * then run EXACTLY the loop of the RCCL path (same shard arithmetic, same payload with the trailing checksum word, same regenerating
 * refit) and call fn where they would call ncclAllGather: there is one sharded implementation, the collective is a plug.  fn is called
 * on the thread that called the planner, between kernel enqueues (it may synchronise `stream`; both pointers lie inside the `workspace`
 * argument of the planner call). */
typedef int (*cadm_allgather_fn)(void* user, const void* send, void* recv, size_t count, void* stream);
int cadm_dist_init_external(cadm_ctx* ctx, int nranks, int rank, cadm_allgather_fn fn, void* user);
/* Did the ranks of the last sharded cadm_cem_plan feed different replicated inputs (obs, history, warm start)?  *mismatch_out = 1 / 0;
 * reading resets the flag.  Synchronises `stream`.  On a mismatch the plan of that call is NaN on every rank; a NaN plan WITHOUT this
 * flag is what a single-rank call returns for the same inputs (a non-finite observation, a diverged model).  Inputs are compared by
 * value: -0.0 equals +0.0 and every NaN equals every NaN. */
int cadm_dist_mismatch(cadm_ctx* ctx, int* mismatch_out, void* stream);
/* (nranks, rank) as RCCL reports them for the ctx's communicator (ncclCommCount / ncclCommUserRank); (1, 0) without one. */
int cadm_dist_info(cadm_ctx* ctx, int* nranks_out, int* rank_out);

/* In-library timing of the dominant kernel (the rollout): when enabled, every
 * cadm_rollout_returns launch is bracketed by hipEvents on the launch stream.
 * cadm_profile_read synchronises, returns the summed elapsed milliseconds and launch count
 * since the last read, and resets both. */
int cadm_profile_enable(cadm_ctx* ctx, int enable);
int cadm_profile_read(cadm_ctx* ctx, float* total_ms_out, int* launches_out);
/* Same for the sharded planner's collective: every ncclAllGather issued by cadm_cem_plan / cadm_rs_plan while profiling is
 * enabled is bracketed too; returns the summed milliseconds and the number of all-gathers since the last read. */
int cadm_profile_read_collective(cadm_ctx* ctx, float* total_ms_out, int* calls_out);
/* (Developer hooks -- comparison kernel, forced launch flavours, phase timing -- are NOT part of this interface and not
 * in libcadm_hip.so: cadm_amd/csrc/dev/dev_api.h, libcadm_hip_dev.so.  The product reads no environment variable.) */

#ifdef __cplusplus
}
#endif
#endif /* CADM_HIP_H */
