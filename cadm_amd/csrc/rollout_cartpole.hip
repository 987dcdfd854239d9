// production rollout kernel instantiations for env kind cartpole (one translation unit per env: parallel builds)
#include "rollout_dispatch.h"
CADM_ROLLOUT_ENV(cartpole, CADM_ENV_CARTPOLE)
