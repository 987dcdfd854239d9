// production rollout kernel instantiations for env kind halfcheetah (one translation unit per env: parallel builds)
#include "rollout_dispatch.h"
CADM_ROLLOUT_ENV(halfcheetah, CADM_ENV_HALFCHEETAH)
