// Packs the raw TF-layout weights W[E,in,out] (reference core/utils.py:636-641) into the split-f16 fragment stream
// of the xdl rollout kernel (layout: xdl_geo.h) and the fp32 D-layout bias tiles.
#include "common.h"
#include "xdl_geo.h"

namespace {

struct XdlPackArgs {
    const float* W[CADM_MAX_HIDDEN_LAYERS + 2];   // hidden 0..NH-1, output_mu, output_logvar   [E, in, out]
    const float* b[CADM_MAX_HIDDEN_LAYERS + 2];   // [E, 1, out]
    unsigned short* dst;                          // [E][member_frags][2][64][8] halfs
    float* bias;                                  // [E][bias_tiles][64][4]
    int* overflow;                                // set to 1 if a weight does not fit the f16 range
    XdlGeo g;
    int E;
    int swish_fold;                               // hidden nonlinearity is swish: log2(e) folded into the packed weights (xdl_geo.h)
};

__device__ __forceinline__ unsigned short f16_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }

// one thread per half of the stream
__global__ void pack_xdl_kernel(const XdlPackArgs a) {
    const XdlGeo& g = a.g;
    const size_t per_member = (size_t)g.member_frags() * 1024;
    const size_t total = per_member * a.E;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx / per_member;
        size_t rem = idx % per_member;
        int w = 0;
        while (rem >= (size_t)g.wave_frags(w) * 1024) { rem -= (size_t)g.wave_frags(w) * 1024; ++w; }
        int fj = rem / 1024;                  // fragment of this wave's per-step sequence
        const int part = (rem % 1024) / 512;
        const int lane = (rem % 512) / 8;
        const int i = rem % 8;
        const int ntw = g.ntw(w);
        // which layer
        int layer, nchl;
        const int nc0s = g.NC0 - g.INV;       // layer-0 chunks in the per-step part of the stream (xdl_geo.h: invariant last chunk)
        const int inv0 = ntw * nc0s + (g.NH - 1) * ntw * g.NCH + g.nhead(w) * g.NCH;      // first fragment of the invariant chunk's block (INV)
        bool invf = false;
        if (g.INV && fj >= inv0) { invf = true; layer = 0; nchl = 1; fj -= inv0; }
        else if (fj < ntw * nc0s) { layer = 0; nchl = nc0s; }
        else {
            fj -= ntw * nc0s;
            layer = 1 + fj / (ntw * g.NCH);
            nchl = g.NCH;
            if (layer < g.NH) fj -= (layer - 1) * ntw * g.NCH;
            else { fj -= (g.NH - 1) * ntw * g.NCH; layer = g.NH; }
        }
        const int grp = lane >> 4, m = lane & 15;
        int tile, c;
        if (invf) { tile = g.tstart(w) + fj; c = g.NC0 - 1; }      // the invariant chunk: one fragment per tile, in tile order
        else if (layer < g.NH) {              // hidden-type outputs: tiles in groups of xdl_group(w), chunk-major inside a group
            const int gsz = xdl_group(w);
            const int gi = fj / (gsz * nchl), jj = fj - gsz * gi * nchl;
            const int gs = (ntw - gsz * gi) < gsz ? (ntw - gsz * gi) : gsz;
            c = jj / gs;
            tile = g.tstart(w) + gsz * gi + jj % gs;
        } else {                              // head: valid slots in order, chunk-major per slot
            int sv = fj / g.NCH;
            c = fj % g.NCH;
            tile = -1;
            for (int s = 0; s < g.NTOW; ++s) {
                if (g.head_tile(w, s) < g.NTO) { if (sv == 0) { tile = g.head_tile(w, s); break; } --sv; }
            }
        }
        // input feature of (chunk c, lane group grp, element i)
        int kin;
        if (layer == 0) { kin = 32 * c + 8 * grp + i; if (kin >= g.K0) kin = -1; }
        else { kin = (2 * c + (i >> 2)) * 16 + 4 * grp + (i & 3); if (kin >= g.HIDR) kin = -1; }
        const int K = layer == 0 ? g.K0 : g.HIDR;     // (master weights: the MODEL's width; units HIDR .. HID - 1 are zero padding)
        float v = 0.0f;
        if (kin >= 0 && tile >= 0) {
            if (layer < g.NH) {
                const int u = 16 * tile + m;
                if (u < g.HIDR) v = a.W[layer][((size_t)e * K + kin) * g.HIDR + u];
            } else {
                const int q = m >> 2, r = m & 3;
                const int d = 8 * tile + 2 * q + (r & 1);
                if (d < g.D) v = a.W[g.NH + (r >> 1)][((size_t)e * K + kin) * g.D + d];
            }
        }
        // swish fold: layer 0 maps true inputs to scaled pre-activations (W * c), the hidden layers map scaled activations to scaled
        // pre-activations (W unchanged), the head maps scaled activations to true outputs (W / c)
        if (a.swish_fold) v *= layer == 0 ? CADM_XDL_LOG2E : layer == g.NH ? (1.0f / CADM_XDL_LOG2E) : 1.0f;
        if (!(fabsf(v) <= 65000.0f)) { *a.overflow = 1; v = 0.0f; }
        const _Float16 v1 = (_Float16)v;
        const _Float16 v2 = (_Float16)((v - (float)v1) * 2048.0f);
        a.dst[idx] = f16_bits(part == 0 ? v1 : v2);
    }
}

// D-layout bias tiles: [e][layer tiles..][lane][4]; hidden: lane (grp,row) reg r <-> unit 16*tile + 4*grp + r;
// head: (mu[d0], mu[d0+1], lv[d0], lv[d0+1]) with d0 = 8*tile + 2*grp
__global__ void pack_xdl_bias_kernel(const XdlPackArgs a) {
    const XdlGeo& g = a.g;
    const size_t per = (size_t)g.bias_tiles() * 256;
    const size_t total = per * a.E;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx / per;
        const size_t rem = idx % per;
        int tile = rem / 256;
        const int lane = (rem % 256) / 4, r = rem % 4, grp = lane >> 4;
        float v = 0.0f;
        if (tile < g.NH * g.NT) {
            const int layer = tile / g.NT;
            tile %= g.NT;
            const int u = 16 * tile + 4 * grp + r;
            if (u < g.HIDR) v = a.b[layer][(size_t)e * g.HIDR + u];
            if (a.swish_fold) v *= CADM_XDL_LOG2E;          // every hidden pre-activation is scaled
        } else {
            tile -= g.NH * g.NT;
            const int d = 8 * tile + 2 * grp + (r & 1);
            if (d < g.D) v = a.b[g.NH + (r >> 1)][(size_t)e * g.D + d];
        }
        if (!(fabsf(v) <= 3.0e38f)) { *a.overflow = 1; v = 0.0f; }      // biases are fp32 accumulator inits: any FINITE value is representable
        a.bias[idx] = v;
    }
}

}  // namespace

int cadm_pack_xdl(cadm_ctx* ctx, hipStream_t s) {
    XdlPackArgs a{};
    for (int l = 0; l < ctx->NH + 2; ++l) { a.W[l] = ctx->ff[l].W; a.b[l] = ctx->ff[l].b; }
    a.dst = ctx->xw;
    a.bias = ctx->xb;
    a.overflow = ctx->xflag;
    a.g = ctx->xg;
    a.E = ctx->E;
    a.swish_fold = ctx->cfg.hidden_act == CADM_ACT_SWISH ? 1 : 0;
    CADM_CHECK_HIP(hipMemsetAsync(ctx->xflag, 0, sizeof(int), s));
    const size_t total = (size_t)ctx->xg.member_frags() * 1024 * ctx->E;
    hipLaunchKernelGGL(pack_xdl_kernel, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, s, a);
    {      // the same weights in the wave-tile kernel's order (one wave owns every tile: rollout_wt.h)
        XdlPackArgs a1 = a;
        a1.dst = ctx->xw1;
        a1.g = ctx->xg1;
        const size_t total1 = (size_t)ctx->xg1.member_frags() * 1024 * ctx->E;
        hipLaunchKernelGGL(pack_xdl_kernel, dim3((unsigned)((total1 + 255) / 256 < 8192 ? (total1 + 255) / 256 : 8192)), dim3(256), 0, s, a1);
    }
    const size_t btotal = (size_t)ctx->xg.bias_tiles() * 256 * ctx->E;
    hipLaunchKernelGGL(pack_xdl_bias_kernel, dim3((unsigned)((btotal + 255) / 256)), dim3(256), 0, s, a);
    CADM_CHECK_HIP(hipGetLastError());
    // a weight outside the f16 range cannot be split: fail loudly rather than plan with a truncated model
    int flag = 0;
    CADM_CHECK_HIP(hipMemcpyAsync(&flag, ctx->xflag, sizeof(int), hipMemcpyDeviceToHost, s));
    CADM_CHECK_HIP(hipStreamSynchronize(s));
    if (flag) {
        cadm_set_error("cadm_repack: a dynamics weight is non-finite or exceeds the split-f16 range (|w| > 65000 as packed: swish nets carry log2(e) "
                       "in layer 0, i.e. |w| > 45052 there), or a bias is non-finite");
        return CADM_EINVAL;
    }
    return CADM_OK;
}
