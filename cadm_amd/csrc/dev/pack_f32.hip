// Packs the raw TF-layout weights W[E,in,out] (core/utils.py:636-641 of the reference) into the
// planner's MFMA-fragment "weight streams" (layout in common.h / DESIGN.md).
#include "../common.h"

// output unit of A-operand row i (= lane & 15) of output tile `tile`.
//   hidden layers: D-layout lane (q,row) register r holds unit 16*tile + 4*r + q, and the MFMA puts
//                  A row i = 4*q + r there  ->  unit(i) = 16*tile + 4*(i&3) + (i>>2)
//   head tiles:    lane (q,row) holds mu[d0], mu[d0+1], lv[d0], lv[d0+1] with d0 = 8*tile + 2*q
// returns -1 for padding; for head tiles *is_lv tells which raw tensor the unit comes from.
__device__ __forceinline__ int out_unit(int head, int tile, int i, int nout, int* is_lv) {
    const int q = i >> 2, r = i & 3;
    *is_lv = 0;
    if (!head) {
        const int u = 16 * tile + 4 * r + q;
        return u < nout ? u : -1;
    }
    const int d = 8 * tile + 2 * q + (r & 1);
    *is_lv = r >> 1;
    return d < nout ? d : -1;
}

// one thread per stream float of one layer (all members)
__global__ void pack_layer_kernel(const float* __restrict__ W, const float* __restrict__ W2,
                                  float* __restrict__ dst, size_t dst_member_stride, int E, int K,
                                  int nout, int nch, int nfo, int nso, int head, int csplit) {
    const int nslots = csplit ? (nch + 3) / 4 : nch;
    const size_t slot = csplit ? (size_t)nso * 256 : (size_t)(nfo * 4 + nso) * 64;
    const size_t wavef = slot * nslots;
    const size_t layerf = wavef * 4;
    const size_t total = layerf * E;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx / layerf;
        size_t rem = idx % layerf;
        const int w = rem / wavef;
        rem %= wavef;
        int c = rem / slot;
        rem %= slot;
        int tile, lane, r;
        if (csplit) {                      // slot j of wave w = chunk w + 4j, float4 blocks per tile
            c = w + 4 * c;
            tile = rem / 256;
            lane = (rem % 256) / 4;
            r = rem % 4;
        } else if (rem < (size_t)nfo * 256) {
            const int blk = rem / 256;
            lane = (rem % 256) / 4;
            r = rem % 4;
            tile = w * nfo + blk;
        } else {
            rem -= (size_t)nfo * 256;
            const int s = rem / 64;
            lane = rem % 64;
            r = w;  // split tiles: wave w owns k-step w of every chunk
            tile = 4 * nfo + s;
        }
        const int fin = 16 * c + 4 * r + (lane >> 4);
        int is_lv;
        const int u = out_unit(head, tile, lane & 15, nout, &is_lv);
        float v = 0.0f;
        if (c < nch && fin < K && u >= 0) {
            const float* src = is_lv ? W2 : W;
            v = src[((size_t)e * K + fin) * nout + u];
        }
        dst[(size_t)e * dst_member_stride + (idx % layerf)] = v;
    }
}

// D-layout bias tiles: dst[e][tile][lane][r]
__global__ void pack_bias_kernel(const float* __restrict__ b, const float* __restrict__ b2,
                                 float* __restrict__ dst, size_t dst_member_stride, int E, int nout,
                                 int ntiles, int head) {
    const size_t per = (size_t)ntiles * 256;
    const size_t total = per * E;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx / per;
        const size_t rem = idx % per;
        const int tile = rem / 256;
        const int lane = (rem % 256) / 4;
        const int r = rem % 4;
        const int q = lane >> 4;
        float v = 0.0f;
        if (!head) {
            const int u = 16 * tile + 4 * r + q;
            if (u < nout) v = b[(size_t)e * nout + u];
        } else {
            const int d = 8 * tile + 2 * q + (r & 1);
            if (d < nout) v = ((r >> 1) ? b2 : b)[(size_t)e * nout + d];
        }
        dst[(size_t)e * dst_member_stride + rem] = v;
    }
}

// fp32 fragment stream of the comparison kernel: allocated on first use, (re)packed after the production stream
// (installed as ctx->dev_pack by cadm_dev_set_rollout)
static LayerGeo make_geo(int K, int ntiles, int head, int nout) {
    LayerGeo g;
    g.K = K;
    g.nch = (K + 15) / 16;
    g.ntiles = ntiles;
    g.nfo = ntiles / 4;
    g.nso = ntiles % 4;
    g.head = head;
    g.nout = nout;
    // head layers whose tiles are all K-split use the chunk-split stream (rollout_f32.h CSPLIT) when the
    // producer layer has exactly one K-split tile as its last chunk (HID % 64 in (0,16], e.g. 200)
    g.csplit = (head && g.nfo == 0 && g.nso > 0 && (g.nch % 4) == 1) ? 1 : 0;
    return g;
}

void cadm_dev_free_f32(cadm_ctx* ctx) {
    if (ctx->wstream) (void)hipFree(ctx->wstream);
    if (ctx->bstream) (void)hipFree(ctx->bstream);
    ctx->wstream = ctx->bstream = nullptr;
}

int cadm_dev_pack_f32(cadm_ctx* ctx, hipStream_t s) {
    const int NH = ctx->NH, E = ctx->E;
    if (!ctx->wstream) {
        const int NT = (ctx->HID + 15) / 16;
        ctx->g0 = make_geo(ctx->K0, NT, 0, ctx->HID);
        ctx->gh = make_geo(ctx->HID, NT, 0, ctx->HID);
        ctx->go = make_geo(ctx->HID, (ctx->D + 7) / 8, 1, ctx->D);
        ctx->wstream_member_floats = ctx->g0.layer_floats() + (size_t)(NH - 1) * ctx->gh.layer_floats() + ctx->go.layer_floats();
        ctx->bstream_member_floats = ctx->g0.bias_floats() + (size_t)(NH - 1) * ctx->gh.bias_floats() + ctx->go.bias_floats();
        CADM_CHECK_HIP(hipMalloc(&ctx->wstream, ctx->wstream_member_floats * E * sizeof(float)));
        CADM_CHECK_HIP(hipMalloc(&ctx->bstream, ctx->bstream_member_floats * E * sizeof(float)));
    }
    size_t woff = 0, boff = 0;

    auto pack = [&](const LayerGeo& g, const DenseRef& a, const DenseRef* a2) {
        const size_t total = g.layer_floats() * E;
        const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(pack_layer_kernel, dim3(grid), dim3(256), 0, s, a.W, a2 ? a2->W : nullptr,
                           ctx->wstream + woff, ctx->wstream_member_floats, E, g.K, g.nout, g.nch,
                           g.nfo, g.nso, g.head, g.csplit);
        const size_t btotal = g.bias_floats() * E;
        const int bgrid = (int)((btotal + 255) / 256);
        hipLaunchKernelGGL(pack_bias_kernel, dim3(bgrid), dim3(256), 0, s, a.b, a2 ? a2->b : nullptr,
                           ctx->bstream + boff, ctx->bstream_member_floats, E, g.nout, g.ntiles, g.head);
        woff += g.layer_floats();
        boff += g.bias_floats();
    };
    pack(ctx->g0, ctx->ff[0], nullptr);
    for (int l = 1; l < NH; ++l) pack(ctx->gh, ctx->ff[l], nullptr);
    pack(ctx->go, ctx->ff[NH], &ctx->ff[NH + 1]);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}
