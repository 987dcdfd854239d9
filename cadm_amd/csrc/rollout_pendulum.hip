// production rollout kernel instantiations for env kind pendulum (one translation unit per env: parallel builds)
#include "rollout_dispatch.h"
CADM_ROLLOUT_ENV(pendulum, CADM_ENV_PENDULUM)
