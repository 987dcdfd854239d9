#!/usr/bin/env python3
"""Generates tools/micro/coissue_bench.hip: the CLEAN MFMA / VALU co-issue experiment VERDICT r4 asked for.

    python tools/micro/gen_coissue_bench.py
    hipcc --offload-arch=gfx950 -O3 tools/micro/coissue_bench.hip -o tools/micro/coissue_bench && tools/micro/coissue_bench

Every stream is ONE `asm volatile` block -- the instruction order is the one written here, hipcc neither packs (no v_pk_*_f32 unless the
variant asks for it), clusters nor re-spaces anything (tests/test_isa_hygiene.py::test_coissue_bench_streams_are_what_they_claim
disassembles the object and checks spacing and the absence of packed fp32 in the scalar variants):

    NM f16 MFMAs on NM different accumulators, F independent filler instructions behind EACH of them (evenly spaced), looped.

Questions, each one a row group of the table the program prints (cycles from s_memtime, per MFMA):
  solo   one wave per SIMD runs the stream                         -> what a wave hides in its own MFMA shadows
  both   two waves per SIMD run the same stream                    -> what the SIMD hides when both of its waves mix MFMAs and fillers
                                                                      (cycles per MFMA OF THE SIMD = wave cycles / (2 NM iters))
  pair   waves 0-3 run MFMAs only, their SIMD partners fillers only -> cross-wave overlap (each also measured alone: mode 1 / 2)
Fillers: fma (v_fma_f32), pkfma (v_pk_fma_f32), pkmul, exp, rcp, cvt (v_cvt_f16_f32), mix (v_fma_mixlo_f16), mov, dsr (ds_read_b128),
epi / epipk (the hidden-layer epilogue's instruction mix, scalar / packed).
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

MFMA = {   # name: (mnemonic, acc regs, nominal cycles in the pipe)
    "16": ("v_mfma_f32_16x16x32_f16", 4, 16),
    "32": ("v_mfma_f32_32x32x16_f16", 16, 32),
}
NFR = 12        # filler registers (a filler's result is next read 12 fillers later)

EPI = ["fma", "fma", "exp", "fma", "rcp", "fma", "cvt", "mix", "fma", "fma", "exp", "fma"]          # roughly the swish epilogue's mix
EPIPK = ["pkfma", "exp", "exp", "pkfma", "rcp", "rcp", "pkmul", "cvt", "mix", "pkfma", "exp", "mix"]


def filler(kind, j):
    r = "%%[f%d]" % (j % NFR)
    r2 = "%%[f%d]" % ((j + 5) % NFR)
    p = "%%[p%d]" % (j % 6)
    if kind == "fma":
        return "v_fma_f32 %s, %s, %%[c], %%[c]" % (r, r)
    if kind == "mov":
        return "v_mov_b32 %s, %%[c]" % r
    if kind == "exp":
        return "v_exp_f32 %s, %s" % (r, r)
    if kind == "rcp":
        return "v_rcp_f32 %s, %s" % (r, r)
    if kind == "cvt":
        return "v_cvt_f16_f32 %s, %s" % (r, r2)
    if kind == "mix":
        return "v_fma_mixlo_f16 %s, %s, %%[c], %%[c]" % (r, r2)
    if kind == "pkfma":
        return "v_pk_fma_f32 %s, %s, %%[pc], %%[pc]" % (p, p)
    if kind == "pkmul":
        return "v_pk_mul_f32 %s, %s, %%[pc]" % (p, p)
    if kind == "dsr":
        return "ds_read_b128 %%[q%d], %%[la]" % (j % 4)
    if kind == "epi":
        return filler(EPI[j % len(EPI)], j)
    if kind == "epipk":
        return filler(EPIPK[j % len(EPIPK)], j)
    raise KeyError(kind)


def stream(mf, nm, kind, F):
    mn = MFMA[mf][0]
    lines = []
    j = 0
    for i in range(nm):
        lines.append("%s %%[a%d], %%[A], %%[B], %%[a%d]" % (mn, i, i))
        for _ in range(F):
            lines.append(filler(kind, j))
            j += 1
    if kind == "dsr" and F:
        lines.append("s_waitcnt lgkmcnt(0)")
    return lines


def variant(name, mf, nm, kind, F, threads, pair):
    mn, nacc, nominal = MFMA[mf]
    acc_t = "floatx4" if nacc == 4 else "floatx16"
    body_a = stream(mf, nm, kind, 0 if pair else F)
    body_b = [filler(kind, q) for q in range(48)] + (["s_waitcnt lgkmcnt(0)"] if kind == "dsr" else [])
    def join(ls):
        return "\\n\"\n            \"".join(ls)
    outs = ["[a%d] \"+v\"(acc[%d])" % (i, i) for i in range(nm)]
    outs += ["[f%d] \"+v\"(fr[%d])" % (i, i) for i in range(NFR)]
    outs += ["[p%d] \"+v\"(pk[%d])" % (i, i) for i in range(6)]
    outs += ["[q%d] \"+v\"(qv[%d])" % (i, i) for i in range(4)]
    ins = ["[A] \"v\"(av)", "[B] \"v\"(bv)", "[c] \"v\"(cv)", "[pc] \"v\"(pcv)", "[la] \"v\"(lds_addr)"]
    return """
// %(name)s: %(mn)s x %(nm)d, %(F)d x %(kind)s behind each%(pairnote)s
__global__ __launch_bounds__(%(threads)d) void k_%(name)s(float* out, long long* cyc, int iters, int iters_b, int mode) {
    __shared__ float sh[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) sh[i] = 0.001f * i;
    __syncthreads();
    %(acc_t)s acc[%(nm)d];
    for (int i = 0; i < %(nm)d; ++i) for (int r = 0; r < %(nacc)d; ++r) acc[i][r] = 0.0f;
    f16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(threadIdx.x * 0.001f + i); bv[i] = (_Float16)(0.5f + 0.01f * i); }
    float fr[%(NFR)d]; for (int i = 0; i < %(NFR)d; ++i) fr[i] = threadIdx.x * 0.5f + i;
    floatx2 pk[6]; for (int i = 0; i < 6; ++i) pk[i] = floatx2{1.0f * i, 2.0f};
    floatx4 qv[4]; for (int i = 0; i < 4; ++i) qv[i] = floatx4{0, 0, 0, 0};
    const float cv = 1.0001f; const floatx2 pcv = floatx2{1.0001f, 0.9999f};
    const unsigned lds_addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool role_a = %(pair)d ? wave < 4 : true;
    const bool on = %(pair)d ? (mode == 0 || (mode == 1 && role_a) || (mode == 2 && !role_a)) : true;
    long long t0 = __builtin_readcyclecounter();
    if (on && role_a) {
        for (int it = 0; it < iters; ++it) {
            asm volatile("%(body_a)s\\n"
                         : %(outs)s
                         : %(ins)s
                         : "memory");
        }
    } else if (on) {
        for (int it = 0; it < iters_b; ++it) {
            asm volatile("%(body_b)s\\n"
                         : %(outs)s
                         : %(ins)s
                         : "memory");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int i = 0; i < %(nm)d; ++i) for (int r = 0; r < %(nacc)d; ++r) s += acc[i][r];
    for (int i = 0; i < %(NFR)d; ++i) s += fr[i];
    for (int i = 0; i < 6; ++i) s += pk[i][0] + pk[i][1];
    for (int i = 0; i < 4; ++i) s += qv[i][0] + qv[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
""" % dict(name=name, mn=mn, nm=nm, F=F, kind=kind, threads=threads, acc_t=acc_t, nacc=nacc, NFR=NFR, pair=1 if pair else 0,
           pairnote=" (pair: waves 0-3 MFMAs only, waves 4-7 48 fillers per iteration)" if pair else "",
           body_a=join(body_a), body_b=join(body_b), outs=", ".join(outs), ins=", ".join(ins))


def variants():
    V = []   # (name, mf, nm, kind, F, threads, pair)
    for mf, nm in (("16", 4), ("16", 8), ("32", 2), ("32", 4)):
        for threads, tag in ((256, "solo"), (512, "both")):
            fs = (0, 1, 2, 3, 4, 5, 6, 8) if mf == "16" else (0, 2, 4, 5, 6, 8, 10, 12)
            for F in fs:
                V.append(("%s_m%s_n%d_fma%d" % (tag, mf, nm, F), mf, nm, "fma", F, threads, False))
    for mf, nm in (("16", 4), ("32", 2)):
        for threads, tag in ((256, "solo"), (512, "both")):
            for kind in ("pkfma", "pkmul", "exp", "rcp", "cvt", "mix", "mov", "dsr", "epi", "epipk"):
                for F in ((1, 2, 3, 4) if mf == "16" else (2, 4, 6, 8)):
                    V.append(("%s_m%s_n%d_%s%d" % (tag, mf, nm, kind, F), mf, nm, kind, F, threads, False))
    for mf, nm in (("16", 4), ("32", 2)):
        for kind in ("fma", "pkfma", "exp", "epi", "epipk", "dsr"):
            V.append(("pair_m%s_n%d_%s" % (mf, nm, kind), mf, nm, kind, 0, 512, True))
    return V


def main():
    V = variants()
    src = ["// GENERATED by tools/micro/gen_coissue_bench.py -- do not edit.",
           "#include <hip/hip_runtime.h>", "#include <stdio.h>", "#include <string.h>",
           "typedef float floatx2 __attribute__((ext_vector_type(2)));",
           "typedef float floatx4 __attribute__((ext_vector_type(4)));",
           "typedef float floatx16 __attribute__((ext_vector_type(16)));",
           "typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));"]
    for v in V:
        src.append(variant(*v))
    src.append("""
typedef void (*kfn)(float*, long long*, int, int, int);
struct V { const char* name; kfn f; int threads; int nm; int F; int nominal; int pair; };
int main(int argc, char** argv) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
    const int iters = 4000;
    V vs[] = {""")
    for (name, mf, nm, kind, F, threads, pair) in V:
        src.append("        {\"%s\", k_%s, %d, %d, %d, %d, %d}," % (name, name, threads, nm, F, MFMA[mf][2], 1 if pair else 0))
    src.append("""    };
    auto run = [&](const V& v, int it_a, int it_b, int mode, long long* h, float* us) {
        (void)hipMemset(cyc, 0, 64);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(v.f, dim3(256), dim3(v.threads), 0, 0, out, cyc, it_a, it_b, mode);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(v.f, dim3(256), dim3(v.threads), 0, 0, out, cyc, it_a, it_b, mode);
        (void)hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: FAILED\\n", v.name); exit(1); }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); *us = ms * 1e3f;
        (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    };
    printf("%-26s %5s %5s %12s %14s %10s\\n", "variant", "fill", "nomin", "cyc/mfma(wave)", "cyc/mfma(SIMD)", "us");
    for (auto& v : vs) {
        if (argc > 1 && !strstr(v.name, argv[1])) continue;
        long long h[8]; float us;
        if (!v.pair) {
            run(v, iters, 0, 0, h, &us);
            const double per = (double)h[0] / (iters * (double)v.nm);
            printf("%-26s %5d %5d %12.1f %14.1f %10.1f\\n", v.name, v.F, v.nominal, per, v.threads == 512 ? per / 2 : per, us);
        } else {
            // filler iterations chosen so that the filler wave ALONE takes about as long as the MFMA wave alone
            long long ha[8], hb[8]; float ua, ub;
            run(v, iters, 0, 1, ha, &ua);
            run(v, 0, 1000, 2, hb, &ub);
            const double cyc_a = (double)ha[0], per_b = (double)hb[4] / 1000.0;
            const int it_b = (int)(cyc_a / per_b + 0.5);
            run(v, 0, it_b, 2, hb, &ub);
            run(v, iters, it_b, 0, h, &us);
            printf("%-26s  A alone %9.0f cyc (%.1f / mfma)   B alone %9.0f cyc (%.2f / filler)   together A %9.0f  B %9.0f   us %.1f / %.1f / %.1f   overlap %.2f\\n",
                   v.name, cyc_a, cyc_a / (iters * (double)v.nm), (double)hb[4], (double)hb[4] / (it_b * 48.0), (double)h[0], (double)h[4], ua, ub, us,
                   (ua + ub - us) / (ua < ub ? ua : ub));
        }
    }
    return 0;
}""")
    path = os.path.join(HERE, "coissue_bench.hip")
    open(path, "w").write("\n".join(src) + "\n")
    print("wrote", path, "(%d variants)" % len(V))


if __name__ == "__main__":
    main()
