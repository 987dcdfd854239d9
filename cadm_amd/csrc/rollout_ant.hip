// production rollout kernel instantiations for env kind ant (one translation unit per env: parallel builds)
#include "rollout_dispatch.h"
CADM_ROLLOUT_ENV(ant, CADM_ENV_ANT)
