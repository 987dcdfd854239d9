// Micro-benchmark: do VALU instructions overlap with f16 MFMAs on gfx950, by MFMA shape?
//   (a) two waves per SIMD, one issuing only MFMAs, its partner only VALU work (as coexec_bench.hip), for v_mfma_f32_16x16x32_f16
//       (16 cycles in the pipe) and v_mfma_f32_32x32x16_f16 (32 cycles, the same FLOP rate);
//   (b) ONE instruction stream: every MFMA followed by K independent plain VALU instructions, both waves of a SIMD running it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coexec32_bench.hip -o /tmp/coexec32 && /tmp/coexec32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int K>      // SHAPE 16: 4 accumulators of 16x16x32; 32: 2 accumulators of 32x32x16 (same registers, same FLOPs per round)
__device__ __forceinline__ void mfma_loop(int iters, float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        floatx4 acc[4];
        for (int k = 0; k < 4; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < K; ++u) v[(k * K + u) & 7] = fmaf(v[(k * K + u) & 7], 1.0001f, 0.25f);
                }
        }
        for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    } else {
        floatx16 acc[2];
        for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2 * K; ++u) v[(k * 2 * K + u) & 7] = fmaf(v[(k * 2 * K + u) & 7], 1.0001f, 0.25f);      // (K per 16-cycle equivalent)
                }
        }
        for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    }
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

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

// what: 0 idle, 1 mfma (SHAPE, K fillers), 2 valu with transcendentals, 3 valu without
template <int FIRST, int SECOND, int SHAPE, int K>
__global__ __launch_bounds__(512) void bench(int iters_m, int iters_v, float* out) {
    extern __shared__ float pad[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int what = wave < 4 ? FIRST : SECOND;
    if (what == 1) mfma_loop<SHAPE, K>(iters_m, out);
    else if (what == 2) valu_loop<true>(iters_v, out);
    else if (what == 3) valu_loop<false>(iters_v, out);
    if (threadIdx.x == 9999) pad[0] = 1.f;
}

template <int F, int S, int SHAPE, int K>
float run(int im, int iv, float* out) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bench<F, S, SHAPE, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((bench<F, S, SHAPE, K>), dim3(256), dim3(512), 100 * 1024, 0, im, iv, out);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((bench<F, S, SHAPE, K>), dim3(256), dim3(512), 100 * 1024, 0, im, iv, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int im = 2000, iv = 600;
    printf("us per launch, 256 workgroups x 8 waves (2 per SIMD).  MFMA loop: %d x 16 x (16x16x32) or %d x 8 x (32x32x16) f16 MFMAs per wave (same FLOPs);\n"
           "VALU loop: %d x 16 epilogue values (V: with exp/rcp, v: plain).  X/Y = waves 0-3 run X, their SIMD partners (waves 4-7) run Y\n", im, im, iv);
#define PAIRS(SH) \
    printf("shape %2d   M/-: %6.1f  M/M: %6.1f  v/-: %6.1f  V/-: %6.1f  M/v: %6.1f  M/V: %6.1f\n", SH, run<1, 0, SH, 0>(im, iv, out), run<1, 1, SH, 0>(im, iv, out), \
           run<3, 0, SH, 0>(im, iv, out), run<2, 0, SH, 0>(im, iv, out), run<1, 3, SH, 0>(im, iv, out), run<1, 2, SH, 0>(im, iv, out));
    PAIRS(16) PAIRS(32)
#define FILL(SH) \
    printf("shape %2d   one stream, K plain VALU per 16 MFMA-cycles; one wave per SIMD  K=0: %6.1f  1: %6.1f  2: %6.1f  3: %6.1f  4: %6.1f  6: %6.1f | two waves per SIMD  K=0: %6.1f  1: %6.1f  2: %6.1f  3: %6.1f  4: %6.1f\n", SH, \
           run<1, 0, SH, 0>(im, iv, out), run<1, 0, SH, 1>(im, iv, out), run<1, 0, SH, 2>(im, iv, out), run<1, 0, SH, 3>(im, iv, out), run<1, 0, SH, 4>(im, iv, out), run<1, 0, SH, 6>(im, iv, out), \
           run<1, 1, SH, 0>(im, iv, out), run<1, 1, SH, 1>(im, iv, out), run<1, 1, SH, 2>(im, iv, out), run<1, 1, SH, 3>(im, iv, out), run<1, 1, SH, 4>(im, iv, out));
    FILL(16) FILL(32)
    return 0;
}
