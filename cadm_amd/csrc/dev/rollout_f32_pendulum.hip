// fp32-MFMA comparison kernel (developer library only) for env kind pendulum
#include "rollout_f32_dispatch.h"
CADM_ROLLOUT_F32_ENV(pendulum, CADM_ENV_PENDULUM)
