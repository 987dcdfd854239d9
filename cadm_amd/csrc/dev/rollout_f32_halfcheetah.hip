// fp32-MFMA comparison kernel (developer library only) for env kind halfcheetah
#include "rollout_f32_dispatch.h"
CADM_ROLLOUT_F32_ENV(halfcheetah, CADM_ENV_HALFCHEETAH)
