#!/usr/bin/env python3
"""Generates tools/micro/wtshape_bench.hip: what bounds a wave-tile BLOCK (rollout_wt.h) -- LDS fragment reads, MFMAs, epilogue VALU --
for candidate shapes of the kernel, as hand-placed asm streams (no compiler scheduling):

  one block = NCH chunks; per chunk the wave reads the pair's 4 fragment parts (4 x ds_read_b128, 1 KB each, the SAME bytes for every wave
  of the workgroup), one chunk ahead, and issues 6 MFMAs per row tile on them; then the pair's epilogue (30 VALU per tile and row tile).

  shape   threads  row tiles/wave  epilogue
  a1      512      1               behind the block's MFMAs (today's kernel)
  a1i     512      1               previous block's epilogue interleaved with this block's MFMAs (software pipeline)
  b2      256      2               behind the MFMAs
  b2i     256      2               interleaved
  c2i     512      2               interleaved (two waves per SIMD x two row tiles: what 512 registers per SIMD cannot hold -- reference point)
"""
import os
HERE = os.path.dirname(os.path.abspath(__file__))
NCH = 7
EPI = ["fma", "min", "exp", "add", "rcp", "mul", "cvt", "mix"]     # per value; 4 values per tile -> cvt/mix are per pair in the kernel, close enough

def valu(kind, j):
    r = "%%[f%d]" % (j % 12)
    r2 = "%%[f%d]" % ((j + 5) % 12)
    return {"fma": "v_fma_f32 %s, %s, %%[c], %%[c]" % (r, r), "min": "v_min_f32 %s, %s, %%[c]" % (r, r), "exp": "v_exp_f32 %s, %s" % (r, r),
            "add": "v_add_f32 %s, %s, %%[c]" % (r, r), "rcp": "v_rcp_f32 %s, %s" % (r, r), "mul": "v_mul_f32 %s, %s, %%[c]" % (r, r),
            "cvt": "v_cvt_f16_f32 %s, %s" % (r, r2), "mix": "v_fma_mixlo_f16 %s, %s, %%[c], %%[c]" % (r, r2)}[kind]

def stream(mt, interleave, nvalu_per_tile=30, dsr=True, mfma=True, transport=0, nwaves=8, spread=False):
    L = []
    pieces = []
    if transport:
        if transport & 1:
            L.append("s_waitcnt vmcnt(0)")
            L.append("s_barrier")
        # this wave's share of the block's 28 pieces (1 KB each): piece i -> LDS slot offset, global offset = lane * 16 + piece * 1024
        n = (28 + nwaves - 1) // nwaves
        for i in range(n if transport & 2 else 0):
            pieces.append(["s_add_u32 m0, %%[m0b], %d" % (i * nwaves * 1024), "s_nop 0", "buffer_load_dwordx4 %%[vo], %%[rs], 0 offen offset:%d lds" % ((i * 1024) % 4096)])
        if not spread:
            for pc in pieces: L += pc
            pieces = []
    nv = 2 * mt * nvalu_per_tile          # epilogue instructions of a pair block
    nm = NCH * 6 * mt
    vq = [valu(EPI[(j // 8) % len(EPI)], j) for j in range(nv)]      # stages of 8 independent values
    def reads(c):
        if not dsr: return []
        s = c & 1
        return ["ds_read_b128 %%[q%d], %%[la] offset:%d" % (s * 4 + i, (c * 4 + i) * 1024) for i in range(4)]
    L += reads(0)
    vi = 0
    mi = 0
    for c in range(NCH):
        if c + 1 < NCH: L += reads(c + 1)
        if dsr: L.append("s_waitcnt lgkmcnt(%d)" % (4 if c + 1 < NCH else 0))
        s = c & 1
        for prod in range(3):
            for k in range(2):
                for h in range(mt):
                    acc = (k * 2 + (1 if prod == 1 else 0)) * mt + h
                    if mfma: L.append("v_mfma_f32_16x16x32_f16 %%[a%d], %%[q%d], %%[B%d], %%[a%d]" % (acc, s * 4 + k * 2 + (1 if prod == 1 else 0), h, acc))
                    mi += 1
                    if pieces and mi % 6 == 0:
                        L += pieces.pop(0)
                    if interleave:
                        want = (mi * nv) // nm
                        while vi < want:
                            L.append(vq[vi]); vi += 1
    while vi < nv:
        L.append(vq[vi]); vi += 1
    for pc in pieces: L += pc
    return L

def kernel(name, threads, mt, interleave, **kw):
    body = "\\n\"\n            \"".join(stream(mt, interleave, **kw))
    nacc = 4 * mt
    outs = ["[a%d] \"+v\"(acc[%d])" % (i, i) for i in range(nacc)] + ["[f%d] \"+v\"(fr[%d])" % (i, i) for i in range(12)] + ["[q%d] \"+v\"(qv[%d])" % (i, i) for i in range(8)]
    ins = ["[B0] \"v\"(b0)", "[B1] \"v\"(b1)", "[c] \"v\"(cv)", "[la] \"v\"(lds_addr)", "[vo] \"v\"(voff)", "[rs] \"s\"(rsrc)", "[m0b] \"s\"(m0base)"]
    return """
__global__ __launch_bounds__(%(threads)d) void k_%(name)s(float* out, long long* cyc, int iters, const float* wsrc) {
    extern __shared__ float sh[];
    for (int i = threadIdx.x; i < 7168; i += blockDim.x) sh[i] = 0.001f * i;
    __syncthreads();
    floatx4 acc[%(nacc)d];
    for (int i = 0; i < %(nacc)d; ++i) acc[i] = floatx4{0, 0, 0, 0};
    f16x8 b0, b1;
    for (int i = 0; i < 8; ++i) { b0[i] = (_Float16)(0.5f + 0.01f * i); b1[i] = (_Float16)(0.25f + 0.01f * i); }
    float fr[12]; for (int i = 0; i < 12; ++i) fr[i] = threadIdx.x * 0.5f + i;
    floatx4 qv[8]; for (int i = 0; i < 8; ++i) qv[i] = floatx4{0, 0, 0, 0};
    const float cv = 1.0001f;
    const unsigned lds_addr = (threadIdx.x & 63) * 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff = (threadIdx.x & 63) * 16 + wave * 1024;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 1 << 20, 0x00020000);
    const unsigned m0base = 28672 + wave * 1024;      // the ring's other slot
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        asm volatile("%(body)s\\n" : %(outs)s : %(ins)s : "memory", "m0");
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int i = 0; i < %(nacc)d; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += fr[i];
    for (int i = 0; i < 8; ++i) s += qv[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
""" % dict(name=name, threads=threads, nacc=nacc, body=body, outs=", ".join(outs), ins=", ".join(ins))

V = [("a1t", 512, 1, False, dict(transport=3)), ("a1it", 512, 1, True, dict(transport=3)), ("a1ts", 512, 1, False, dict(transport=3, spread=True)),
     ("a1its", 512, 1, True, dict(transport=3, spread=True)),
     ("b2t", 256, 2, False, dict(transport=3, nwaves=4)), ("b2it", 256, 2, True, dict(transport=3, nwaves=4)), ("b2its", 256, 2, True, dict(transport=3, nwaves=4, spread=True)),
     ("a1b", 512, 1, False, dict(transport=1)), ("a1d", 512, 1, False, dict(transport=2)), ("a1ib", 512, 1, True, dict(transport=1)), ("a1id", 512, 1, True, dict(transport=2)), ("a1ids", 512, 1, True, dict(transport=2, spread=True)), ("a1", 512, 1, False, {}), ("a1i", 512, 1, True, {}), ("b2", 256, 2, False, {}), ("b2i", 256, 2, True, {}), ("c2i", 512, 2, True, {}),
     ("a1_nodsr", 512, 1, False, dict(dsr=False)), ("a1_nomfma", 512, 1, False, dict(mfma=False)), ("a1_noepi", 512, 1, False, dict(nvalu_per_tile=0)),
     ("a1i_epi20", 512, 1, True, dict(nvalu_per_tile=20)), ("b2i_epi20", 256, 2, True, dict(nvalu_per_tile=20)),
     ("b2_nodsr", 256, 2, False, dict(dsr=False)), ("b2i_nodsr", 256, 2, True, dict(dsr=False)), ("b2_noepi", 256, 2, False, dict(nvalu_per_tile=0)),
     ("a1_solo", 256, 1, False, {}), ("a1i_solo", 256, 1, True, {})]

def main():
    src = ["// GENERATED by tools/micro/gen_wtshape_bench.py -- do not edit.", "#include <hip/hip_runtime.h>", "#include <stdio.h>",
           "typedef float floatx4 __attribute__((ext_vector_type(4)));", "typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));"]
    for (n, t, mt, il, kw) in V:
        src.append(kernel(n, t, mt, il, **kw))
    src.append("""
typedef void (*kfn)(float*, long long*, int, const float*);
struct V { const char* name; kfn f; int threads; int mt; };
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64); float* wsrc; (void)hipMalloc(&wsrc, 1 << 20); (void)hipMemset(wsrc, 0, 1 << 20);
    const int iters = 3000;
    V vs[] = {""")
    for (n, t, mt, il, kw) in V:
        src.append("        {\"%s\", k_%s, %d, %d}," % (n, n, t, mt))
    src.append("""    };
    printf("%-14s %8s %10s %14s %18s\\n", "shape", "threads", "us", "cyc/block(wave)", "ns per 16-row block per CU");
    for (auto& v : vs) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(v.f, dim3(256), dim3(v.threads), 2 * 28672, 0, out, cyc, iters, wsrc);
        (void)hipEventRecord(e0);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(v.f, dim3(256), dim3(v.threads), 2 * 28672, 0, out, cyc, iters, wsrc);
        (void)hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: FAILED\\n", v.name); return 1; }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        const double us = ms * 1e3 / 3;
        const int tiles = (v.threads / 64) * v.mt;
        printf("%-14s %8d %10.1f %14.0f %18.2f\\n", v.name, v.threads, us, (double)h[0] / iters, us * 1e3 / iters / tiles);
    }
    return 0;
}""")
    open(os.path.join(HERE, "wtshape_bench.hip"), "w").write("\n".join(src) + "\n")
    print("wrote wtshape_bench.hip (%d shapes)" % len(V))

if __name__ == "__main__":
    main()
