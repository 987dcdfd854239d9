// Future-window builder of the context model's training set: the device twin of
// ModelSampleProcessor.process_samples' `context` branch (/root/reference/cadm/samplers/model_sample_processor.py:58-100).
// Pure data movement, so it is written over 4- or 8-byte WORDS: float32 and the reference's float64 arrays both come out
// bit for bit.  Per path p with len_p recorded steps (zero-padded to L = max(len_p, F + 1), :62-68) and n = L - 1 rows:
//     concat_obs[row s, future i]      = step(s + i)          concat_next_obs[s, i] = step(s + 1 + i)     (:73-80)
//     concat_act[s, i]                 = action(s + i)                                                   (:74,82-83)
//     concat_bool[s, j] = 0 if s == 0 (the reference's `concat_bool[-0]` quirk, :85-86 at i = 0)
//                         0 if 1 <= n - s <= F - 1 and j >= max(n - s - remainder, 0)   (:85-86), else 1
//     cp_obs / cp_act rows = the history windows of the first n steps (:99-100)
// where step(t) / action(t) are zero for t >= len_p.  HBM-bound: every output word is written once, inputs are re-read from L2.
#include "common.h"

namespace {

template <class W>
struct WinArgs {
    const W *obs, *act, *cp_obs, *cp_act;
    const int32_t *path_off, *row_path, *row_step;
    W *concat_obs, *concat_act, *concat_next_obs, *concat_bool, *cp_obs_out, *cp_act_out;
    W one;
    int D, A, Dh, Ah, N, F;
};

template <class W>
__global__ __launch_bounds__(256) void build_windows_kernel(const WinArgs<W> a) {
    const int D = a.D, A = a.A, F = a.F;
    const long long per_row = (long long)F * (2 * D + A + 1) + a.Dh + a.Ah;      // output words per row
    const long long total = per_row * a.N;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / per_row);
        int w = (int)(idx % per_row);
        const int p = a.row_path[r], s = a.row_step[r];
        const int base = a.path_off[p], len = a.path_off[p + 1] - base;
        const int L = len > F + 1 ? len : F + 1, n = L - 1, rem = L - len;
        if (w < F * D) {                                        // concat_obs
            const int i = w / D, d = w % D, t = s + i;
            a.concat_obs[(size_t)r * F * D + w] = t < len ? a.obs[(size_t)(base + t) * D + d] : (W)0;
        } else if ((w -= F * D) < F * A) {                      // concat_act
            const int i = w / A, d = w % A, t = s + i;
            a.concat_act[(size_t)r * F * A + w] = t < len ? a.act[(size_t)(base + t) * A + d] : (W)0;
        } else if ((w -= F * A) < F * D) {                      // concat_next_obs
            const int i = w / D, d = w % D, t = s + 1 + i;
            a.concat_next_obs[(size_t)r * F * D + w] = t < len ? a.obs[(size_t)(base + t) * D + d] : (W)0;
        } else if ((w -= F * D) < F) {                          // concat_bool
            const int i = n - s;                                // this row is row -i of its path
            const int cut = i - rem > 0 ? i - rem : 0;
            const bool zero = s == 0 || (i >= 1 && i <= F - 1 && w >= cut);
            a.concat_bool[(size_t)r * F + w] = zero ? (W)0 : a.one;
        } else if ((w -= F) < a.Dh) {                           // history windows of the row's step
            a.cp_obs_out[(size_t)r * a.Dh + w] = s < len ? a.cp_obs[(size_t)(base + s) * a.Dh + w] : (W)0;
        } else {
            w -= a.Dh;
            a.cp_act_out[(size_t)r * a.Ah + w] = s < len ? a.cp_act[(size_t)(base + s) * a.Ah + w] : (W)0;
        }
    }
}

template <class W>
int launch(WinArgs<W> a, hipStream_t s) {
    const long long total = ((long long)a.F * (2 * a.D + a.A + 1) + a.Dh + a.Ah) * a.N;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(build_windows_kernel<W>, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, a);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

}  // namespace

extern "C" int cadm_build_windows(const void* obs, const void* act, const void* cp_obs, const void* cp_act, int elem_bytes,
                                  int D, int A, int Dh, int Ah, const int32_t* path_off, const int32_t* row_path,
                                  const int32_t* row_step, int N, int F, void* concat_obs, void* concat_act,
                                  void* concat_next_obs, void* concat_bool, void* cp_obs_out, void* cp_act_out, void* stream) {
    CADM_REQUIRE(obs && act && path_off && row_path && row_step && concat_obs && concat_act && concat_next_obs && concat_bool,
                 "cadm_build_windows: null argument");
    CADM_REQUIRE((Dh == 0 || (cp_obs && cp_obs_out)) && (Ah == 0 || (cp_act && cp_act_out)), "cadm_build_windows: history arrays missing");
    CADM_REQUIRE(D > 0 && A > 0 && Dh >= 0 && Ah >= 0 && N >= 0 && F >= 1, "cadm_build_windows: bad dimensions");
    CADM_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "cadm_build_windows: elements must be 4 or 8 bytes, got %d", elem_bytes);
    if (N == 0) return CADM_OK;
    if (elem_bytes == 4) {
        WinArgs<uint32_t> a{(const uint32_t*)obs, (const uint32_t*)act, (const uint32_t*)cp_obs, (const uint32_t*)cp_act, path_off,
                            row_path, row_step, (uint32_t*)concat_obs, (uint32_t*)concat_act, (uint32_t*)concat_next_obs,
                            (uint32_t*)concat_bool, (uint32_t*)cp_obs_out, (uint32_t*)cp_act_out, 0x3f800000u, D, A, Dh, Ah, N, F};
        return launch(a, (hipStream_t)stream);
    }
    WinArgs<uint64_t> a{(const uint64_t*)obs, (const uint64_t*)act, (const uint64_t*)cp_obs, (const uint64_t*)cp_act, path_off,
                        row_path, row_step, (uint64_t*)concat_obs, (uint64_t*)concat_act, (uint64_t*)concat_next_obs,
                        (uint64_t*)concat_bool, (uint64_t*)cp_obs_out, (uint64_t*)cp_act_out, 0x3ff0000000000000ull, D, A, Dh, Ah, N, F};
    return launch(a, (hipStream_t)stream);
}
