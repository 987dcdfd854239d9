// Issue rate of v_mfma_f32_16x16x16_f16 against v_mfma_f32_16x16x32_f16 on gfx950 (one wave per SIMD, independent accumulators):
// does the K = 16 form run in half the time of the K = 32 form ("K-tail of every 200-wide layer on 16x16x16")?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_k16_rate.hip -o /tmp/mfma_k16_rate && /tmp/mfma_k16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    floatx4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    const f16x4 a4 = __builtin_shufflevector(a, a, 0, 1, 2, 3), b4 = __builtin_shufflevector(b, b, 0, 1, 2, 3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (KIND == 32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else if constexpr (KIND == 16) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
            else {      // mixed: 6 x K32 + 1 x K16 per accumulator pair, the shape of a 7-chunk sweep with a K16 tail
                if (i == 7) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int kind : {32, 16, 0, 32, 16, 0}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (kind == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        else if (kind == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-22s %8.3f ms  %6.2f ns per MFMA per SIMD   (s_memtime ticks per MFMA %.2f)\n",
               kind == 32 ? "16x16x32_f16" : kind == 16 ? "16x16x16_f16" : "7 x K32 + 1 x K16", ms, ms * 1e6 / (iters * 8.0), (double)c / (iters * 8.0));
    }
    return 0;
}
