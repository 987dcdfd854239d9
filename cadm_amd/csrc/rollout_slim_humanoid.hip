// production rollout kernel instantiations for env kind slim_humanoid (one translation unit per env: parallel builds)
#include "rollout_dispatch.h"
CADM_ROLLOUT_ENV(slim_humanoid, CADM_ENV_SLIM_HUMANOID)
