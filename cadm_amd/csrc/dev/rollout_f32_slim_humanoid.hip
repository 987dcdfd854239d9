// fp32-MFMA comparison kernel (developer library only) for env kind slim_humanoid
#include "rollout_f32_dispatch.h"
CADM_ROLLOUT_F32_ENV(slim_humanoid, CADM_ENV_SLIM_HUMANOID)
