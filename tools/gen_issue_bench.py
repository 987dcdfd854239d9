#!/usr/bin/env python3
"""Generates tools/issue_bench.hip: hand-placed instruction streams (inline asm, fixed order) that measure how many
filler instructions a wave can hide in the shadow of its own MFMAs on gfx950, per MFMA flavour.

    python tools/gen_issue_bench.py && hipcc --offload-arch=gfx950 -O3 tools/issue_bench.hip -o /tmp/issue_bench && /tmp/issue_bench

Every variant is one asm block of NM MFMAs on NM different accumulators with F fillers placed after each MFMA,
looped; cycles from s_memtime on wave 0 of block 0; 256 blocks so that every CU is busy (DVFS as in the real kernel).
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MFMA = {   # name: (mnemonic, acc regs, a/b regs, nominal cycles)
    "f32_16x16x4": ("v_mfma_f32_16x16x4_f32", 4, 1, 32),
    "f32_32x32x2": ("v_mfma_f32_32x32x2_f32", 16, 1, 64),
    "bf16_16x16x32": ("v_mfma_f32_16x16x32_bf16", 4, 4, 16),
    "bf16_32x32x16": ("v_mfma_f32_32x32x16_bf16", 16, 4, 32),
}
NM = 8          # MFMAs (= accumulators) per block
NFR = 8         # filler registers


def filler(kind, j):
    r = "%%[f%d]" % (j % NFR)
    if kind == "fma":
        return "v_fma_f32 %s, %s, %%[c], %%[c]" % (r, r)
    if kind == "mov":
        return "v_mov_b32 %s, %%[c]" % r
    if kind == "exp":
        return "v_exp_f32 %s, %s" % (r, r)
    if kind == "pkfma":
        return "v_pk_fma_f32 %%[p%d], %%[p%d], %%[pc], %%[pc]" % (j % 4, j % 4)
    if kind == "dsr":
        return "ds_read_b128 %%[q%d], %%[la]" % (j % 4)
    if kind == "dsw":
        return "ds_write_b128 %%[la], %%[q%d]" % (j % 4)
    if kind == "vld":
        return "buffer_load_dwordx4 %%[q%d], %%[vo], %%[rs], 0 offen" % (j % 4)
    if kind == "salu":
        return "s_add_u32 %[s0], %[s0], 1"
    if kind == "nop":
        return "s_nop 0"
    raise KeyError(kind)


def variant(name, mf, kind, F, threads=256, role=None):
    mn, nacc, nab, nominal = MFMA[mf]
    acc_t = "floatx4" if nacc == 4 else "floatx16"
    ab_t = "float" if nab == 1 else "bf16x8"
    lines = []
    j = 0
    for i in range(NM):
        lines.append("%s %%[a%d], %%[A], %%[B], %%[a%d]" % (mn, i, i))
        for _ in range(F):
            lines.append(filler(kind, j))
            j += 1
    if kind in ("dsr", "vld"):
        lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if kind == "dsw":
        lines.append("s_waitcnt lgkmcnt(0)")
    body = "\\n\"\n            \"".join(lines)
    outs = ["[a%d] \"+v\"(acc[%d])" % (i, i) for i in range(NM)]
    outs += ["[f%d] \"+v\"(fr[%d])" % (i, i) for i in range(NFR)]
    outs += ["[p%d] \"+v\"(pk[%d])" % (i, i) for i in range(4)]
    outs += ["[q%d] \"+v\"(qv[%d])" % (i, i) for i in range(4)]
    outs += ["[s0] \"+s\"(sc)"]
    ins = ["[A] \"v\"(av)", "[B] \"v\"(bv)", "[c] \"v\"(cv)", "[pc] \"v\"(pcv)", "[la] \"v\"(lds_addr)", "[vo] \"v\"(voff)",
           "[rs] \"s\"(rsrc)"]
    return """
__global__ __launch_bounds__(%(threads)d) void k_%(name)s(float* out, const float* in, long long* cyc, int iters) {
    __shared__ float sh[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) sh[i] = 0.001f * i;
    __syncthreads();
    %(acc_t)s acc[%(NM)d];
    for (int i = 0; i < %(NM)d; ++i) for (int r = 0; r < %(nacc)d; ++r) acc[i][r] = 0.0f;
    %(ab_t)s av, bv;
    init_ab(av, threadIdx.x * 0.001f); init_ab(bv, 1.0f + threadIdx.x * 0.002f);
    float fr[%(NFR)d]; for (int i = 0; i < %(NFR)d; ++i) fr[i] = threadIdx.x * 0.5f + i;
    floatx2 pk[4]; for (int i = 0; i < 4; ++i) pk[i] = floatx2{1.0f * i, 2.0f};
    floatx4 qv[4]; for (int i = 0; i < 4; ++i) qv[i] = floatx4{0, 0, 0, 0};
    const float cv = 1.0001f; const floatx2 pcv = floatx2{1.0001f, 0.9999f};
    unsigned sc = 0;
    const unsigned lds_addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    const unsigned voff = (threadIdx.x & 63) * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 4096 * 4, 0x00020000);
    const bool mfma_role = %(role_expr)s;
    long long t0 = __builtin_readcyclecounter();
    if (mfma_role) {
        for (int it = 0; it < iters; ++it) {
            asm volatile("%(body)s\\n"
                         : %(outs)s
                         : %(ins)s
                         : "memory");
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            asm volatile(%(valu_body)s
                         : %(outs)s
                         : %(ins)s
                         : "memory");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = (float)sc;
    for (int i = 0; i < %(NM)d; ++i) for (int r = 0; r < %(nacc)d; ++r) s += acc[i][r];
    for (int i = 0; i < %(NFR)d; ++i) s += fr[i];
    for (int i = 0; i < 4; ++i) s += pk[i][0] + pk[i][1] + qv[i][0] + qv[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
""" % dict(name=name, threads=threads, acc_t=acc_t, ab_t=ab_t, NM=NM, nacc=nacc, NFR=NFR, body=body,
           outs=", ".join(outs), ins=", ".join(ins),
           role_expr="true" if role is None else "(threadIdx.x >> 6) < 4",
           valu_body="\"" + "\\n\"\n            \"".join(filler(role or "fma", q) for q in range(32)) + "\\n\"")


def main():
    V = []   # (name, mf, kind, F, threads, role)
    for F in (0, 1, 2, 3, 4, 5, 6, 8):
        V.append(("f32_16_fma%d" % F, "f32_16x16x4", "fma", F, 256, None))
    for kind in ("mov", "exp", "pkfma", "dsr", "dsw", "vld", "salu", "nop"):
        for F in (1, 2, 4):
            V.append(("f32_16_%s%d" % (kind, F), "f32_16x16x4", kind, F, 256, None))
    for F in (0, 2, 4, 8, 12):
        V.append(("f32_32_fma%d" % F, "f32_32x32x2", "fma", F, 256, None))
    for F in (0, 1, 2, 3, 4):
        V.append(("bf16_16_fma%d" % F, "bf16_16x16x32", "fma", F, 256, None))
    for kind in ("dsr", "vld", "exp"):
        for F in (1, 2):
            V.append(("bf16_16_%s%d" % (kind, F), "bf16_16x16x32", kind, F, 256, None))
    for F in (0, 2, 4, 5, 6, 8):
        V.append(("bf16_32_fma%d" % F, "bf16_32x32x16", "fma", F, 256, None))
    # two waves per SIMD: waves 0-3 pure MFMA stream, waves 4-7 pure filler stream (32 fillers per iteration)
    for mf, tag in (("f32_16x16x4", "f32_16"), ("bf16_16x16x32", "bf16_16")):
        for kind in ("fma", "exp", "dsr"):
            V.append(("pair_%s_%s" % (tag, kind), mf, "fma", 0, 512, kind))
    # two waves per SIMD, both running the same MFMA + fillers stream
    for F in (0, 2, 4):
        V.append(("both_f32_16_fma%d" % F, "f32_16x16x4", "fma", F, 512, None))
    src = ["// GENERATED by tools/gen_issue_bench.py -- do not edit.",
           "#include <hip/hip_runtime.h>", "#include <stdio.h>",
           "typedef float floatx2 __attribute__((ext_vector_type(2)));",
           "typedef float floatx4 __attribute__((ext_vector_type(4)));",
           "typedef float floatx16 __attribute__((ext_vector_type(16)));",
           "typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));",
           "__device__ inline void init_ab(float& x, float v) { x = v; }",
           "__device__ inline void init_ab(bf16x8& x, float v) { for (int i = 0; i < 8; ++i) x[i] = (__bf16)(v + i); }"]
    for v in V:
        src.append(variant(*v))
    src.append("""
typedef void (*kfn)(float*, const float*, long long*, int);
struct V { const char* name; kfn f; int threads; int nm; int F; int nominal; int paired; };
int main() {
    float *out, *in; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMemset(in, 0, 4096 * 4); (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    V vs[] = {""")
    for (name, mf, kind, F, threads, role) in V:
        src.append("        {\"%s\", k_%s, %d, %d, %d, %d, %d}," % (name, name, threads, NM, F, MFMA[mf][3], 1 if role else 0))
    src.append("""    };
    printf("%-22s %8s %10s %10s %10s\\n", "variant", "fill/mf", "cyc/mfma", "nominal", "w4 cyc/it");
    for (auto& v : vs) {
        (void)hipMemset(cyc, 0, 64);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(v.f, dim3(256), dim3(v.threads), 0, 0, out, in, cyc, iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: FAILED\\n", v.name); return 1; }
        long long h[8];
        (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("%-22s %8d %10.1f %10d %10.1f\\n", v.name, v.F, (double)h[0] / (iters * (double)v.nm), v.nominal,
               v.threads == 512 ? (double)h[4] / iters : 0.0);
    }
    return 0;
}""")
    path = os.path.join(ROOT, "tools", "issue_bench.hip")
    open(path, "w").write("\n".join(src) + "\n")
    print("wrote", path, "(%d variants)" % len(V))


if __name__ == "__main__":
    main()
