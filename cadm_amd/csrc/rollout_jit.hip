// ONE instantiation of the production rollout kernel as a side module, built on demand by cadm_amd/jit.py:
//   hipcc ... -DCADM_JIT_MODULE -DCADM_JIT_ENV=e -DCADM_JIT_C=c -DCADM_JIT_HID=h -DCADM_JIT_NH=n -DCADM_JIT_ACT=a -DCADM_JIT_NOISE=k
// for geometries the library does not carry (`--hidden_size` / `--context_out_dim` of run_cadm_pets.py:122-135 outside the
// compiled lists, other depths, the other nonlinearities of dynamics.py:17-24).  One module per noise mode (device Philox /
// injected / deterministic): the planner needs one, parity tests another, and they build in parallel.
// The module calls nothing in libcadm_hip.so: errors go through the function pointer the ctx carries.
#include "rollout_xdl.h"

void (*cadm_jit_set_error)(const char*, ...) = nullptr;

extern "C" int cadm_jit_rollout(cadm_ctx* ctx, const RolloutArgs* a, int rows_per_member, void* stream) {
    cadm_jit_set_error = ctx->set_error;
    return xdl_launch<CADM_JIT_ENV, CADM_JIT_C, CADM_JIT_HID, CADM_JIT_NH, CADM_JIT_ACT, CADM_JIT_NOISE>(ctx, *a, rows_per_member, (hipStream_t)stream);
}
// what the module was built for: checked against the ctx by cadm_register_rollout
extern "C" void cadm_jit_describe(int out[8]) {
    out[0] = CADM_CTX_LAYOUT_TAG; out[1] = CADM_JIT_ENV; out[2] = CADM_JIT_C; out[3] = CADM_JIT_HID; out[4] = CADM_JIT_NH;
    out[5] = CADM_JIT_ACT; out[6] = CADM_JIT_NOISE; out[7] = (int)sizeof(cadm_ctx);
}
