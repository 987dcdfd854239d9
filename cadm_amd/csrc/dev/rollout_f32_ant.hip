// fp32-MFMA comparison kernel (developer library only) for env kind ant
#include "rollout_f32_dispatch.h"
CADM_ROLLOUT_F32_ENV(ant, CADM_ENV_ANT)
