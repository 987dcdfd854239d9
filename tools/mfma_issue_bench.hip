// Micro-benchmark behind DESIGN.md 4.1 "what one wave can overlap": cycles per k-step of NT back-to-back
// v_mfma_f32_16x16x4_f32 with independent VALU / LDS / VMEM work added to the same wave.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_issue_bench.hip -o /tmp/mfma_issue_bench && /tmp/mfma_issue_bench
// Result on MI355X (profiles/r1_mfma_issue_microbench.md): 32.6 cycles per MFMA alone; every independent VALU
// instruction of the same wave adds ~3.5 cycles -- a wave does not overlap anything with its own MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
// VAR: 0 pure MFMA; 2: + NV independent VALU fma per step (clustered after the MFMAs);
//      3: + NV independent VALU, interleaved one group after each MFMA (sched_group_barrier);
//      4: + 1 independent ds_read per step; 5: + 1 independent global load (same line) per step
template <int NT, int VAR, int NV>
__global__ __launch_bounds__(256) void k(float* out, const float* in, long long* cyc, int iters) {
    __shared__ float sh[4096];
    sh[threadIdx.x] = threadIdx.x; sh[threadIdx.x + 256] = 1.0f;
    __syncthreads();
    floatx4 acc[NT];
    for (int j = 0; j < NT; ++j) acc[j] = floatx4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b[NT];
    for (int j = 0; j < NT; ++j) b[j] = j + 1.0f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
    float ld = 0.0f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[j], 0, 0, 0);
            if (VAR == 2 || VAR == 3) {
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
            }
            if (VAR == 4) ld += sh[(threadIdx.x + u * 64 + i) & 4095];
            if (VAR == 5) ld += in[(threadIdx.x + u * 64) & 1023];
            if (VAR == 3) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (NV + NT - 1) / NT, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = ld;
    for (int j = 0; j < NT; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NT, int VAR, int NV>
void run(float* out, float* in, long long* cyc) {
    const int iters = 1000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NT, VAR, NV>), dim3(256), dim3(256), 0, 0, out, in, cyc, iters);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("NT=%d VAR=%d NV=%2d: %.1f cycles per step (MFMA alone %d)\n", NT, VAR, NV, (double)h / (iters * 8.0), 32 * NT);
}
int main() {
    float *out, *in; long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&in, 4096 * 4); hipMemset(in, 0, 4096 * 4); hipMalloc(&cyc, 8);
    run<4, 0, 0>(out, in, cyc);
    run<4, 2, 4>(out, in, cyc); run<4, 2, 8>(out, in, cyc); run<4, 2, 16>(out, in, cyc); run<4, 2, 32>(out, in, cyc);
    run<4, 3, 4>(out, in, cyc); run<4, 3, 8>(out, in, cyc); run<4, 3, 16>(out, in, cyc); run<4, 3, 32>(out, in, cyc);
    run<4, 4, 0>(out, in, cyc); run<4, 5, 0>(out, in, cyc);
    run<1, 2, 4>(out, in, cyc); run<1, 3, 4>(out, in, cyc); run<1, 3, 8>(out, in, cyc);
    return 0;
}
