// Training step (placeholder until the fwd/bwd/Adam kernels land; keeps the ABI complete).
#include "common.h"

void cadm_train_free(cadm_ctx* ctx) { (void)ctx; }

extern "C" int cadm_train_configure(cadm_ctx* ctx, const cadm_train_hparams* hp, int max_batch) {
    (void)ctx; (void)hp; (void)max_batch;
    cadm_set_error("cadm_train_configure: training kernels not built yet");
    return CADM_EINVAL;
}
extern "C" int cadm_train_step(cadm_ctx* ctx, const float* obs, const float* act, const float* delta,
                               const float* obs_next, const float* back_delta, const float* cp_obs,
                               const float* cp_act, int B, int train, float* losses_out, void* stream) {
    (void)ctx; (void)obs; (void)act; (void)delta; (void)obs_next; (void)back_delta; (void)cp_obs; (void)cp_act;
    (void)B; (void)train; (void)losses_out; (void)stream;
    cadm_set_error("cadm_train_step: training kernels not built yet");
    return CADM_EINVAL;
}
extern "C" int cadm_train_reset(cadm_ctx* ctx, void* stream) {
    (void)ctx; (void)stream;
    cadm_set_error("cadm_train_reset: training kernels not built yet");
    return CADM_EINVAL;
}
