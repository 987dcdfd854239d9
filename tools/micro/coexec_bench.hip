// Micro-benchmark: can the two waves of a SIMD overlap one wave's f16 MFMAs with the other's VALU work (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coexec_bench.hip -o tools/micro/coexec_bench && tools/micro/coexec_bench
// 512-thread workgroups (two waves per SIMD), one per CU.  Waves 0-3 = "first" wave of each SIMD, 4-7 = their partners.
// modes: M/M both MFMA, V/V both VALU, M/V first MFMA + partner VALU, M/- and -/V alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__device__ __forceinline__ void mfma_loop(int iters, float* out) {
    floatx4 acc[NACC];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    for (int k = 0; k < NACC; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the hidden-layer epilogue's instruction mix: fma, min, mul, exp2, add, rcp, mul, cvt f16, cvt back, fma, cvt  (8 independent chains)
template <bool TRANS>
__device__ __forceinline__ void valu_loop(int iters, float* out) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float pre[8], s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { pre[i] = fminf(fmaf(v[i], 4.8828125e-4f, 0.25f), 60000.0f); s[i] = pre[i] * -1.4426950408889634f; }
            if (TRANS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = __builtin_amdgcn_rcpf(1.0f + s[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = fmaf(s[i], 0.5f, 1.0f);
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = fmaf(s[i], s[i], 0.125f) + 1.0f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float h = pre[i] * s[i];
                const _Float16 h1 = (_Float16)h;
                const _Float16 h2 = (_Float16)fmaf((float)h1, -2048.0f, h * 2048.0f);
                v[i] = (float)h1 + (float)h2;
            }
        }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// first: what waves 0-3 do, second: what waves 4-7 do (0 idle, 1 mfma, 2 valu with transcendentals, 3 valu without)
template <int FIRST, int SECOND, int NACC>
__global__ __launch_bounds__(512) void bench(int iters_m, int iters_v, float* out) {
    extern __shared__ float pad[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int what = wave < 4 ? FIRST : SECOND;
    if (what == 1) mfma_loop<NACC>(iters_m, out);
    else if (what == 2) valu_loop<true>(iters_v, out);
    else if (what == 3) valu_loop<false>(iters_v, out);
    if (threadIdx.x == 9999) pad[0] = 1.f;
}

template <int F, int S, int NACC>
float run(int im, int iv, float* out) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bench<F, S, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((bench<F, S, NACC>), dim3(256), dim3(512), 100 * 1024, 0, im, iv, out);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((bench<F, S, NACC>), dim3(256), dim3(512), 100 * 1024, 0, im, iv, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int im = 2000, iv = 600;      // mfma: im * 4 * NACC MFMAs per wave; valu: iv * 2 * 8 chains x ~12 ops
    printf("us per launch (256 workgroups x 8 waves; MFMA loop = %d x 4 x NACC v_mfma_f32_16x16x32_f16, VALU loop = %d x 16 epilogue values)\n", im, iv);
#define ROW(NACC) \
    printf("NACC=%d  M/-: %7.1f   -/M: %7.1f   M/M: %7.1f   V/-: %7.1f   V/V: %7.1f   M/V: %7.1f   V/M: %7.1f   | no-trans VALU  v/-: %7.1f  v/v: %7.1f  M/v: %7.1f\n", NACC, \
           run<1, 0, NACC>(im, iv, out), run<0, 1, NACC>(im, iv, out), run<1, 1, NACC>(im, iv, out), run<2, 0, NACC>(im, iv, out), run<2, 2, NACC>(im, iv, out), \
           run<1, 2, NACC>(im, iv, out), run<2, 1, NACC>(im, iv, out), run<3, 0, NACC>(im, iv, out), run<3, 3, NACC>(im, iv, out), run<1, 3, NACC>(im, iv, out));
    ROW(4) ROW(6)
    return 0;
}
