// fp32-MFMA comparison kernel (developer library only) for env kind cartpole
#include "rollout_f32_dispatch.h"
CADM_ROLLOUT_F32_ENV(cartpole, CADM_ENV_CARTPOLE)
