// CEM / random-shooting bookkeeping kernels: action sampling (core/utils.py:425-429, 498-503),
// particle mean (:474), elite refit (:475-486), RS argmax (:554-561), final clip (dynamics.py:365-366).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// action sampling
// ---------------------------------------------------------------------------------------------
// (sample_action: common.h -- also used by the fused head of the staged planner call, context.hip)
// candidates [c0, c0 + nc) of every env, at their GLOBAL positions in out [m, n, H, A] (a rank of a sharded planner draws only its own
// shard: the draws are keyed by the global element index, so what it writes is what every other rank would have written there)
__global__ void sample_actions_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                      const float* __restrict__ z, uint32_t seed, uint32_t call, int it,
                                      int m, int n, int H, int A, float lb, float ub,
                                      float* __restrict__ out, int c0, int nc) {
    const int HA = H * A;
    const size_t per = (size_t)nc * HA, total = (size_t)m * per;
    for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
        const int mi = (int)(q / per);
        const size_t r = q - (size_t)mi * per;
        const size_t L = ((size_t)mi * n + c0) * HA + r;
        const int ta = (int)(r % HA);
        out[L] = sample_action(mean[(size_t)mi * HA + ta], var[(size_t)mi * HA + ta], z, L, seed, call, it, lb, ub);
    }
}

__global__ void sample_uniform_kernel(uint32_t seed, uint32_t call, int m, int n, int H, int A, int discrete,
                                      float* __restrict__ out, int32_t* __restrict__ raw) {
    if (!discrete) {
        const size_t total = (size_t)m * n * H * A;
        for (size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x; L < total; L += (size_t)gridDim.x * blockDim.x) {
            uint32_t r[4];
            philox4x32_10((uint32_t)(L & 0xFFFFFFFFull), 0u, (uint32_t)(L >> 32), CADM_STREAM_UNI, seed, call, r);
            out[L] = 2.0f * u01(r[0]) - 1.0f;                                      // :502
        }
    } else {
        const size_t total = (size_t)m * n * H;
        for (size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x; L < total; L += (size_t)gridDim.x * blockDim.x) {
            uint32_t r[4];
            philox4x32_10((uint32_t)(L & 0xFFFFFFFFull), 0u, (uint32_t)(L >> 32), CADM_STREAM_UNI, seed, call, r);
            int k = (int)(u01(r[0]) * (float)A);                                   // :499
            if (k >= A) k = A - 1;
            for (int a = 0; a < A; ++a) out[L * A + a] = (a == k) ? 1.0f : 0.0f;   // :500 one_hot
            if (raw) raw[L] = k;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// particle mean (:474)
// ---------------------------------------------------------------------------------------------
// tail: (sharded planner) one more word behind the means -- the checksum of this rank's inputs, which travels in the all-gather payload
__global__ void particle_mean_kernel(const float* __restrict__ rows, int total, int p, float* __restrict__ out, const unsigned* tail) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (tail && i == 0) out[total] = __uint_as_float(*tail);
    if (i >= total) return;
    float s = 0.0f;
    for (int j = 0; j < p; ++j) s += rows[(size_t)i * p + j];
    out[i] = s / (float)p;
}

// ---------------------------------------------------------------------------------------------
// elite refit: bitonic sort of (return desc, index asc) keys in LDS, then statistics
// ---------------------------------------------------------------------------------------------
// (make_key: common.h)
// gathered candidate returns: rank g's [m, n_local] block at g * gstride (gstride = m * n_local, + 1 with the trailing input checksum)
__device__ __forceinline__ float cand_at(const float* cand, int G, int n_local, int m, int mi, int ni, int gstride = 0) {
    const size_t gs = gstride > 0 ? (size_t)gstride : (size_t)m * n_local;
    return cand[(size_t)(ni / n_local) * gs + (size_t)mi * n_local + (ni % n_local)];
}

// candidate return: from the gathered per-candidate means, or (single-rank fused path) the particle mean
// (core/utils.py:474) taken on the fly from the row returns [m, n, p]
__device__ __forceinline__ float cand_value(const float* cand, const float* rows, int p, int G, int n_local, int m, int mi, int ni, int gstride = 0) {
    if (!rows) return cand_at(cand, G, n_local, m, mi, ni, gstride);
    const float* r = rows + ((size_t)mi * n_local + ni) * p;
    float s = 0.0f;
    for (int j = 0; j < p; ++j) s += r[j];
    return s / (float)p;
}

__device__ void bitonic_sort_lds(uint64_t* keys, int npow2) {
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__global__ void cem_refit_kernel(const float* __restrict__ cand, const float* __restrict__ rows, int p, int G, int n_local,
                                 const float* __restrict__ actions,
                                 int m, int H, int A, int K, float alpha, int npow2, const float* mean_in,
                                 const float* var_in, float* mean_out, float* var_out, int32_t* __restrict__ elites_out,
                                 float* __restrict__ plan_out, float lo, float hi, int do_clip, int part_off,
                                 unsigned* done_flag, unsigned done_val, RefitRegen rg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);
    const int mi = blockIdx.x;
    const int n = G * n_local;
    // sharded planner: every rank's payload ends with the checksum of the inputs it was fed (RefitRegen); one that differs from this
    // rank's poisons the plan with NaN on EVERY rank (each sees all checksums) -- the host raises on it (HipEngine.cem_plan_host)
    __shared__ int poison_s;
    if (threadIdx.x == 0) {
        int bad = 0;
        if (rg.gstride > 0 && rg.my_rank >= 0) {
            const unsigned mine = __float_as_uint(cand[(size_t)rg.my_rank * rg.gstride + (size_t)m * n_local]);
            for (int g = 0; g < G; ++g) bad |= __float_as_uint(cand[(size_t)g * rg.gstride + (size_t)m * n_local]) != mine;
            // the mismatch has its own signal (cadm_dist_mismatch / the words behind a staged call's completion flags): a NaN plan alone
            // is also what a single-rank call returns for a non-finite observation
            if (bad && rg.mismatch) atomicOr(rg.mismatch, 1u);
            if (bad && rg.mismatch_host) rg.mismatch_host[mi] = 1u;      // (fenced to system scope with the plan, below)
        }
        poison_s = bad;
    }
    if (rg.gstride <= 0) rg.gstride = m * n_local;
    constexpr int SMALL_N = 256, LISTMAX = 1024;
    bool done = false;
    if (n <= SMALL_N) {
        // small n: every candidate's rank by counting (keys are unique: index in the low word); the K best
        // land sorted in keys[0..K) with one barrier instead of the O(log^2 n) barriers of a sort
        uint64_t* raw = keys + npow2;
        for (int i = threadIdx.x; i < n; i += blockDim.x) raw[i] = make_key(cand_value(cand, rows, p, G, n_local, m, mi, i, rg.gstride), (uint32_t)i);
        __syncthreads();
        // thread (i, part) counts the keys below key i in its part of the key list, 8 independent LDS reads at a time; the parts meet in
        // an LDS integer (keys[] is not used yet: its first n words hold the ranks)
        int* rank_s = reinterpret_cast<int*>(keys);
        for (int i = threadIdx.x; i < n; i += blockDim.x) rank_s[i] = 0;
        __syncthreads();
        {
            const int parts = (int)blockDim.x / n > 0 ? (int)blockDim.x / n : 1;          // n <= 256, 1024 threads: >= 4
            const int i = threadIdx.x % n, part = threadIdx.x / n;
            if (part < parts) {
                const int qn = (n + parts - 1) / parts, j0 = part * qn, j1 = j0 + qn < n ? j0 + qn : n;
                const uint64_t ki = raw[i];
                int cnt = 0, j = j0;
                for (; j + 8 <= j1; j += 8) {
                    uint64_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = raw[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cnt += v[u] < ki ? 1 : 0;
                }
                for (; j < j1; ++j) cnt += raw[j] < ki ? 1 : 0;
                if (cnt) atomicAdd(&rank_s[i], cnt);
            }
        }
        __syncthreads();
        int myrank = -1;
        uint64_t mykey = 0;
        if ((int)threadIdx.x < n) { myrank = rank_s[threadIdx.x]; mykey = raw[threadIdx.x]; }
        __syncthreads();
        if (myrank >= 0 && myrank < K) keys[myrank] = mykey;
        __syncthreads();
        done = true;
    } else {
        // larger n (every rank of a sharded planner refits over the GLOBAL candidate set): radix-select the K-th smallest
        // key's value word (4 passes of 8-bit LDS histograms), collect the keys at or below it (K plus value ties) and
        // rank only those.  O(n) instead of O(n^2 / threads) LDS reads; tie order (lower index first) is untouched
        // because the final ranking uses the full (value, index) keys.
        uint64_t* list = keys + LISTMAX;                       // [LISTMAX]
        uint64_t* raw = keys + 2 * LISTMAX;                    // [n]
        unsigned* hist = reinterpret_cast<unsigned*>(raw + n); // [256] + {prefix, krem, count}
        unsigned* ctl = hist + 256;
        for (int i = threadIdx.x; i < n; i += blockDim.x) raw[i] = make_key(cand_value(cand, rows, p, G, n_local, m, mi, i, rg.gstride), (uint32_t)i);
        if (threadIdx.x == 0) { ctl[0] = 0u; ctl[1] = (unsigned)K; ctl[2] = 0u; }
        for (int pass = 3; pass >= 0; --pass) {
            for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0u;
            __syncthreads();
            const unsigned prefix = ctl[0];
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const unsigned hv = (unsigned)(raw[i] >> 32);
                if (pass == 3 || (hv >> (8 * (pass + 1))) == prefix) atomicAdd(&hist[(hv >> (8 * pass)) & 255u], 1u);
            }
            __syncthreads();
            if (threadIdx.x < 64) {                            // wave 0: 256-bin prefix scan, 4 bins per lane
                const unsigned krem = ctl[1];
                const int l = threadIdx.x;
                const unsigned h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
                const unsigned tot = h0 + h1 + h2 + h3;
                unsigned incl = tot;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned up = __shfl_up(incl, d, 64);
                    if (l >= d) incl += up;
                }
                const unsigned long long hit = __ballot(incl >= krem);       // krem <= elements under the prefix: never empty
                const int first = __ffsll((long long)hit) - 1;
                if (l == first) {
                    unsigned cum = incl - tot, b = 4u * l;
                    if (cum + h0 < krem) { cum += h0; ++b;
                        if (cum + h1 < krem) { cum += h1; ++b;
                            if (cum + h2 < krem) { cum += h2; ++b; } } }
                    ctl[0] = (prefix << 8) | b;
                    ctl[1] = krem - cum;
                }
            }
            __syncthreads();
        }
        const unsigned hvk = ctl[0];                            // value word of the K-th smallest key
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const uint64_t ki = raw[i];
            if ((unsigned)(ki >> 32) <= hvk) {
                const unsigned pos = atomicAdd(&ctl[2], 1u);
                if (pos < (unsigned)LISTMAX) list[pos] = ki;
            }
        }
        __syncthreads();
        const int cnt = (int)ctl[2];
        if (cnt <= LISTMAX) {
            for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
                const uint64_t ki = list[i];
                int rank = 0;
                for (int j = 0; j < cnt; ++j) rank += list[j] < ki ? 1 : 0;
                if (rank < K) keys[rank] = ki;
            }
            __syncthreads();
            done = true;
        } else {
            __syncthreads();                                    // more than LISTMAX ties on the K-th value: full sort below
        }
    }
    if (!done) {
        for (int i = threadIdx.x; i < npow2; i += blockDim.x)
            keys[i] = i < n ? make_key(cand_value(cand, rows, p, G, n_local, m, mi, i, rg.gstride), (uint32_t)i) : ~0ull;
        __syncthreads();
        bitonic_sort_lds(keys, npow2);                                             // tf.nn.top_k, :475
    }
    if (elites_out)
        for (int k = threadIdx.x; k < K; k += blockDim.x) elites_out[(size_t)mi * K + k] = (int32_t)(keys[k] & 0xFFFFFFFFu);
    // elite statistics: KG thread groups each take every KG-th elite of one (t, a) element, partial sums meet in LDS
    // (fixed order -> deterministic); the elites' actions stay in registers between the mean and the variance pass
    const int HA = H * A;
    const float* act_m = actions + (size_t)mi * n * HA;
    // an elite's action element: read from the candidates' buffer, or (sharded planner: this rank drew only its own shard) drawn again
    // from the counter-based RNG by its global element index -- the same number, without any exchange of action sequences
    auto elite_at = [&](uint64_t key, int ta) -> float {
        const size_t c = (size_t)(key & 0xFFFFFFFFu);
        if (!rg.on) return act_m[c * HA + ta];
        const size_t L = ((size_t)mi * n + c) * HA + ta;
        return sample_action(mean_in[(size_t)mi * HA + ta], var_in[(size_t)mi * HA + ta], nullptr, L, rg.seed, rg.call, rg.it, rg.lb, rg.ub);
    };
    float* part = reinterpret_cast<float*>(smem_raw + part_off);                  // [KG][HA]
    constexpr int MAXE = 16;                                                       // elites per thread (K <= KG * MAXE)
    const int KG = min((int)blockDim.x / HA, K) > 0 ? min((int)blockDim.x / HA, K) : 1;
    const int tq = threadIdx.x;
    const bool par = KG * MAXE >= K && HA * KG <= (int)blockDim.x;
    if (par) {
        const int ta = tq % HA, kg = tq / HA;
        const bool on = kg < KG;
        float v[MAXE];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int k = kg + i * KG;
            v[i] = (on && k < K) ? elite_at(keys[k < K ? k : 0], ta) : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < MAXE; ++i) s += v[i];
        if (on) part[kg * HA + ta] = s;
        __syncthreads();
        float nm = 0.0f;
        for (int g = 0; g < KG; ++g) nm += part[g * HA + ta];
        nm = nm / (float)K;                                                        // :482
        __syncthreads();
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const float d = v[i] - nm;
            q += (on && kg + i * KG < K) ? d * d : 0.0f;
        }
        if (on) part[kg * HA + ta] = q;
        __syncthreads();
        if (tq < HA) {
            float nv = 0.0f;
            for (int g = 0; g < KG; ++g) nv += part[g * HA + ta];
            nv = nv / (float)K;                                                    // :483
            const size_t o = (size_t)mi * HA + ta;
            float mo = mean_in[o] * alpha + (1.0f - alpha) * nm;                   // :485
            if (poison_s) mo = __uint_as_float(0x7fc00000u);                       // (ranks fed different inputs: NaN everywhere)
            mean_out[o] = mo;
            var_out[o] = var_in[o] * alpha + (1.0f - alpha) * nv;                  // :486
            if (plan_out) plan_out[o] = (do_clip && !poison_s) ? fminf(fmaxf(mo, lo), hi) : mo;   // dynamics.py:365-366 (last iteration)
        }
    } else {
        for (int ta = threadIdx.x; ta < HA; ta += blockDim.x) {
            float s = 0.0f;
            for (int k = 0; k < K; ++k) s += elite_at(keys[k], ta);
            const float nm = s / (float)K;
            float v = 0.0f;
            for (int k = 0; k < K; ++k) {
                const float d = elite_at(keys[k], ta) - nm;
                v += d * d;
            }
            const float nv = v / (float)K;
            const size_t o = (size_t)mi * HA + ta;
            float mo = mean_in[o] * alpha + (1.0f - alpha) * nm;
            if (poison_s) mo = __uint_as_float(0x7fc00000u);
            mean_out[o] = mo;
            var_out[o] = var_in[o] * alpha + (1.0f - alpha) * nv;
            if (plan_out) plan_out[o] = (do_clip && !poison_s) ? fminf(fmaxf(mo, lo), hi) : mo;
        }
    }
    // completion flag of the staged planner call (cadm_cem_plan_staged): the host polls it in pinned memory instead of
    // sleeping in hipStreamSynchronize.  Every thread's plan stores are fenced to system scope before the flag is released.
    if (done_flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done_flag + mi, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused planner step for small candidate sets (n <= 256): elite refit of iteration `it` AND the sampling of iteration it + 1,
// parallel over the (t, a) elements of the plan.  Workgroup (mi, s) ranks the n candidates itself (cheap, deterministic: every
// workgroup finds the same elites), then owns EPW consecutive elements: their elite statistics, the EMA update of mean / var
// (every element is read and written by its owner only: mean_in may alias mean_out) and the next iteration's n candidates of these elements (in place in
// `actions`: the elites' values of these elements were read before).  Arithmetic = cem_refit_kernel + sample_actions_kernel
// (same summation order, same Philox counters): the fused planner equals the stepwise composition bit for bit.
// ---------------------------------------------------------------------------------------------
#ifndef CADM_FUSED_EPW
#define CADM_FUSED_EPW 4      // elements per workgroup: n x EPW <= 1024 samples = one round of the workgroup's threads at n <= 256
#endif
// how many of the 64 keys at `q` (LDS, 16-byte aligned, every lane the same address: broadcast reads) are below ki -- the reads are
// independent and issued 16 keys at a time (a rolled loop with a run-time bound waited out one LDS round trip per key: 2.9 us of the kernel)
__device__ __forceinline__ int count_below_64(const uint64_t* q, uint64_t ki) {
    const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(q);
    int cnt = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        ulonglong2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p2[8 * b + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) cnt += (v[j].x < ki ? 1 : 0) + (v[j].y < ki ? 1 : 0);
    }
    return cnt;
}
// candidate return on the fly (cand_value) with the p row returns fetched as float4s where the layout allows; same left-to-right sum
__device__ __forceinline__ float cand_value_v4(const float* cand, const float* rows, int p, int G, int n_local, int m, int mi, int ni) {
    if (!rows) return cand_at(cand, G, n_local, m, mi, ni);
    const float* r = rows + ((size_t)mi * n_local + ni) * p;
    float s = 0.0f;
    if ((p & 3) == 0 && p <= 32 && (reinterpret_cast<uintptr_t>(r) & 15) == 0) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 4 * j < p ? reinterpret_cast<const float4*>(r)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (4 * j < p) { s += v[j].x; s += v[j].y; s += v[j].z; s += v[j].w; }
    } else {
        for (int j = 0; j < p; ++j) s += r[j];
    }
    return s / (float)p;
}
__global__ __launch_bounds__(1024) void cem_refit_sample_kernel(const float* __restrict__ cand, const float* __restrict__ rows, int p, int G,
                                                               int n_local, float* __restrict__ actions, int m, int H, int A, int K, float alpha,
                                                               const float* mean_in, const float* var_in, float* mean_out, float* var_out, float lb,
                                                               float ub, uint32_t seed, uint32_t call, int next_it) {
    __shared__ __attribute__((aligned(16))) uint64_t raw[256];
    __shared__ uint64_t keys[64];
    __shared__ int rank_q[4][256];
    __shared__ float vals[CADM_FUSED_EPW][64];       // [EPW][K <= 64] elite values of this workgroup's elements
    __shared__ float nd[2 * CADM_FUSED_EPW];
    const int HA = H * A, n = G * n_local, tid = threadIdx.x;
    const int NS = (HA + CADM_FUSED_EPW - 1) / CADM_FUSED_EPW;
    const int mi = blockIdx.x / NS, ta0 = (blockIdx.x % NS) * CADM_FUSED_EPW;
    const int ne = HA - ta0 < CADM_FUSED_EPW ? HA - ta0 : CADM_FUSED_EPW;
    const int wave = tid >> 6, lane = tid & 63;
    // (requested now, consumed behind the statistics: their round trip used to sit at the end of the kernel's dependent chain)
    float mean_pre = 0.0f, var_pre = 0.0f;
    if (wave < ne && lane == 0) { mean_pre = mean_in[(size_t)mi * HA + ta0 + wave]; var_pre = var_in[(size_t)mi * HA + ta0 + wave]; }
    // ---- elites: rank by counting (keys are unique: candidate index in the low word), as cem_refit_kernel's small-n path;
    //      the 256 x 256 comparisons (keys padded with +inf keys) are cut in 4 column quarters of 64 over the 1024 threads
    if (tid < 256) raw[tid] = tid < n ? make_key(cand_value_v4(cand, rows, p, G, n_local, m, mi, tid), (uint32_t)tid) : ~0ull;
    __syncthreads();
    {
        const int i = tid & 255, quarter = tid >> 8;
        rank_q[quarter][i] = count_below_64(raw + 64 * quarter, raw[i]);
    }
    __syncthreads();
    if (tid < n) {
        const int rk = rank_q[0][tid] + rank_q[1][tid] + rank_q[2][tid] + rank_q[3][tid];
        if (rk < K) keys[rk] = raw[tid];
    }
    __syncthreads();
    // ---- one WAVE per owned element: its K elite values (lane k: elite k), mean / biased variance in cem_refit_kernel's summation
    //      order -- KG interleaved groups of up to 16 elites whose partial sums are added in group order ("par"), or one sequential
    //      sum when that split does not cover K -- the EMA update, and the element's new (mean, var) for the sampling below.
    //      Everything between the gather and the block barrier is wave-local: LDS round trips without workgroup barriers.
    float* act_m = actions + (size_t)mi * n * HA;
    if (wave < ne) {
        const int e = wave;
        float* vv = vals[e];
        vv[lane] = lane < K ? act_m[(size_t)(keys[lane] & 0xFFFFFFFFu) * HA + ta0 + e] : 0.0f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int MAXE = 16;
        const int KG0 = 1024 / HA > 0 ? 1024 / HA : 1;
        const int KG = KG0 < K ? KG0 : K;
        const bool par = KG * MAXE >= K;
        float nm, nv;
        if (par) {
            float sum = 0.0f;
            if (lane < KG)
                for (int i = 0; i < MAXE; ++i) { const int k = lane + i * KG; sum += k < K ? vv[k] : 0.0f; }
            nm = 0.0f;
            for (int g = 0; g < KG; ++g) nm += __shfl(sum, g, 64);
            nm = nm / (float)K;                                                    // :482
            float qq = 0.0f;
            if (lane < KG)
                for (int i = 0; i < MAXE; ++i) {
                    const int k = lane + i * KG;
                    const float d = (k < K ? vv[k] : 0.0f) - nm;
                    qq += k < K ? d * d : 0.0f;
                }
            nv = 0.0f;
            for (int g = 0; g < KG; ++g) nv += __shfl(qq, g, 64);
            nv = nv / (float)K;                                                    // :483
        } else {
            float sum = 0.0f;
            for (int k = 0; k < K; ++k) sum += vv[k];
            nm = sum / (float)K;
            float v = 0.0f;
            for (int k = 0; k < K; ++k) { const float d = vv[k] - nm; v += d * d; }
            nv = v / (float)K;
        }
        if (lane == 0) {
            const size_t o = (size_t)mi * HA + ta0 + e;
            const float mo = mean_pre * alpha + (1.0f - alpha) * nm;               // :485
            const float vo = var_pre * alpha + (1.0f - alpha) * nv;                // :486
            mean_out[o] = mo; var_out[o] = vo;
            nd[e] = mo; nd[CADM_FUSED_EPW + e] = vo;
        }
    }
    __syncthreads();
    // ---- the next iteration's candidates of the owned elements (in place: the elites above were read before the barrier)
    for (int q = tid; q < n * ne; q += blockDim.x) {
        const int c = q / ne, e = q % ne;
        const size_t L = ((size_t)mi * n + c) * HA + ta0 + e;
        actions[L] = sample_action(nd[e], nd[CADM_FUSED_EPW + e], nullptr, L, seed, call, next_it, lb, ub);
    }
}

int cadm_launch_refit_sample(cadm_ctx* ctx, const float* cand_returns, const float* rows, int G, int n_local, float* actions, int m,
                             const float* mean_in, const float* var_in, float* mean_out, float* var_out, uint32_t seed, uint32_t call,
                             int next_it, hipStream_t stream) {
    // (the LAST refit of a call stays cem_refit_kernel, one workgroup per env: the same element-parallel form with the plan and the
    //  completion flags written by 45 workgroups -- a system-scope fence and a few PCIe stores each, a per-env arrival counter -- measured
    //  ~16 us against its 10.0, round 6)
    const int HA = ctx->H * ctx->A, NS = (HA + CADM_FUSED_EPW - 1) / CADM_FUSED_EPW;
    hipLaunchKernelGGL(cem_refit_sample_kernel, dim3(m * NS), dim3(1024), 0, stream, cand_returns, rows, ctx->p, G, n_local, actions, m,
                       ctx->H, ctx->A, ctx->cfg.num_elites, ctx->cfg.alpha, mean_in, var_in, mean_out, var_out, ctx->cfg.lower_bound,
                       ctx->cfg.upper_bound, seed, call, next_it);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}
// the fused step covers the rank-by-counting regime of the refit and elite sets that fit its LDS arrays
bool cadm_refit_sample_ok(const cadm_ctx* ctx, int n) { return n <= 256 && n >= ctx->cfg.num_elites && ctx->cfg.num_elites <= 64; }

// RS: first maximum over candidates (tf.argmax), gather the first action (:555-561)
__global__ void rs_select_kernel(const float* __restrict__ cand, int G, int n_local, const float* __restrict__ actions,
                                 int m, int H, int A, float* __restrict__ out, int32_t* __restrict__ best_out) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int mi = blockIdx.x;
    const int n = G * n_local;
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = cand_at(cand, G, n_local, m, mi, i);
        if (v > bv || (v == bv && i < bi) || bi == 0x7FFFFFFF) { bv = v; bi = i; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ov = sv[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (oi != 0x7FFFFFFF && (si[threadIdx.x] == 0x7FFFFFFF || ov > sv[threadIdx.x] ||
                                     (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const int best = si[0];
    if (threadIdx.x == 0 && best_out) best_out[mi] = best;
    for (int a = threadIdx.x; a < A; a += blockDim.x)
        out[(size_t)mi * A + a] = actions[((size_t)mi * n + best) * H * A + a];
}

__global__ void clip_kernel(const float* __restrict__ in, float* __restrict__ out, int total, float lo, float hi, int do_clip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float v = in[i];
    out[i] = do_clip ? fminf(fmaxf(v, lo), hi) : v;
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

extern "C" int cadm_sample_actions_shard(cadm_ctx* ctx, const float* mean, const float* var, const float* z,
                                         uint32_t seed, uint32_t call, int it, int m, int n_global, int cand_offset, int n_local,
                                         float* actions_out, void* stream) {
    CADM_REQUIRE(ctx && mean && var && actions_out && m > 0 && n_global > 0 && cand_offset >= 0 && n_local > 0 &&
                 cand_offset + n_local <= n_global, "cadm_sample_actions: bad arguments");
    CADM_ON_DEVICE(ctx);
    const size_t total = (size_t)m * n_local * ctx->H * ctx->A;
    hipLaunchKernelGGL(sample_actions_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, mean, var, z,
                       seed, call, it, m, n_global, ctx->H, ctx->A, ctx->cfg.lower_bound, ctx->cfg.upper_bound, actions_out, cand_offset, n_local);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}
extern "C" int cadm_sample_actions(cadm_ctx* ctx, const float* mean, const float* var, const float* z,
                                   uint32_t seed, uint32_t call, int it, int m, int n_global,
                                   float* actions_out, void* stream) {
    return cadm_sample_actions_shard(ctx, mean, var, z, seed, call, it, m, n_global, 0, n_global, actions_out, stream);
}

extern "C" int cadm_sample_uniform(cadm_ctx* ctx, uint32_t seed, uint32_t call, int m, int n_global,
                                   float* actions_out, int32_t* raw_out, void* stream) {
    CADM_REQUIRE(ctx && actions_out && m > 0 && n_global > 0, "cadm_sample_uniform: bad arguments");
    CADM_ON_DEVICE(ctx);
    const size_t total = (size_t)m * n_global * ctx->H * (ctx->cfg.discrete ? 1 : ctx->A);
    hipLaunchKernelGGL(sample_uniform_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, seed, call, m,
                       n_global, ctx->H, ctx->A, ctx->cfg.discrete, actions_out, raw_out);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

// Checksum of the replicated per-call inputs of a sharded planner call: position-weighted sum of the (canonicalised) 32-bit patterns, modulo 2^32
// (exact, order-independent accumulation; NaN-safe; a permutation or a single changed value changes it).  One workgroup.
__global__ void input_checksum_kernel(const float* a0, int n0, const float* a1, int n1, const float* a2, int n2, const float* a3, int n3,
                                      const float* a4, int n4, unsigned* out) {
    __shared__ unsigned red[256];
    const float* ptr[5] = {a0, a1, a2, a3, a4};
    const int cnt[5] = {n0, n1, n2, n3, n4};
    unsigned h = 0u, base = 1u;
    for (int q = 0; q < 5; ++q) {
        if (ptr[q])
            for (int i = threadIdx.x; i < cnt[q]; i += blockDim.x) {
                // inputs are compared BY VALUE: -0.0 is +0.0, every NaN is the canonical quiet NaN (numerically equal inputs
                // with different bit patterns must not poison a plan)
                const float v = ptr[q][i];
                const unsigned bits = v != v ? 0x7fc00000u : v == 0.0f ? 0u : __float_as_uint(v);
                h += bits * (2u * (base + (unsigned)i) + 1u);
            }
        base += (unsigned)cnt[q] + 7u;
    }
    red[threadIdx.x] = h;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) *out = red[0] | 1u;      // (never the bit pattern of 0.0f)
}
int cadm_launch_input_checksum(cadm_ctx* ctx, const float* obs, const float* cp_obs, const float* cp_act, const float* mean, const float* var,
                               int m, unsigned* out, hipStream_t s) {
    const int HA = ctx->H * ctx->A, Hh = ctx->cfg.history_length;
    hipLaunchKernelGGL(input_checksum_kernel, dim3(1), dim3(256), 0, s, obs, m * ctx->D, cp_obs, cp_obs ? m * ctx->D * Hh : 0, cp_act,
                       cp_act ? m * ctx->A * Hh : 0, mean, m * HA, var, m * HA, out);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

extern "C" int cadm_particle_mean(cadm_ctx* ctx, const float* returns_rows, int m, int n_local,
                                  float* cand_returns, void* stream) {
    CADM_REQUIRE(ctx && returns_rows && cand_returns && m > 0 && n_local > 0, "cadm_particle_mean: bad arguments");
    CADM_ON_DEVICE(ctx);
    const int total = m * n_local;
    hipLaunchKernelGGL(particle_mean_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, returns_rows,
                       total, ctx->p, cand_returns, (const unsigned*)nullptr);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}
int cadm_launch_particle_mean_tail(cadm_ctx* ctx, const float* returns_rows, int m, int n_local, float* cand_returns, const unsigned* tail,
                                   hipStream_t stream) {
    const int total = m * n_local;
    hipLaunchKernelGGL(particle_mean_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, returns_rows, total, ctx->p, cand_returns, tail);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

int cadm_launch_refit(cadm_ctx* ctx, const float* cand_returns, const float* rows, int G, int n_local, const float* actions,
                      int m, const float* mean_in, const float* var_in, float* mean_out, float* var_out, int32_t* elites_out,
                      float* plan_out, hipStream_t stream, const RefitRegen* regen) {
    RefitRegen rg{};
    if (regen) rg = *regen;
    rg.lb = ctx->cfg.lower_bound; rg.ub = ctx->cfg.upper_bound;
    const int n = G * n_local;
    CADM_REQUIRE(n >= ctx->cfg.num_elites, "cadm_cem_refit: n_candidates %d < num_elites %d (tf.nn.top_k would fail)",
                 n, ctx->cfg.num_elites);
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    // counting path: keys[npow2] + raw[npow2]; select path: keys/list[2 * 1024] + raw[n] + histogram; sort fallback: keys[npow2]
    size_t lds_keys = (size_t)npow2 * sizeof(uint64_t) * (n <= 256 ? 2 : 1);
    if (n > 256) {
        const size_t sel = (size_t)(2 * 1024 + n) * sizeof(uint64_t) + 260 * sizeof(unsigned);
        lds_keys = sel > lds_keys ? sel : lds_keys;
        lds_keys = (lds_keys + 15) & ~(size_t)15;
    }
    const int HA = ctx->H * ctx->A;
    const int KG = 1024 / HA > 0 ? 1024 / HA : 1;
    const size_t lds = lds_keys + (size_t)KG * HA * sizeof(float);
    CADM_REQUIRE(lds <= 156 * 1024 && npow2 <= 16384,
                 "cadm_cem_refit: n_candidates %d exceeds the in-LDS sort capacity (16384)", n);
    const void* fn = reinterpret_cast<const void*>(&cem_refit_kernel);
    if (!ctx->attr_done.count(fn)) {    // per ctx = per device
        CADM_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        ctx->attr_done.insert(fn);
    }
    hipLaunchKernelGGL(cem_refit_kernel, dim3(m), dim3(1024), lds, stream, cand_returns, rows, ctx->p, G, n_local, actions,
                       m, ctx->H, ctx->A, ctx->cfg.num_elites, ctx->cfg.alpha, npow2, mean_in, var_in, mean_out, var_out,
                       elites_out, plan_out, ctx->cfg.lower_bound, ctx->cfg.upper_bound, ctx->cfg.discrete ? 0 : 1, (int)lds_keys,
                       plan_out ? ctx->plan_done : nullptr, ctx->plan_done_val, rg);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

extern "C" int cadm_cem_refit(cadm_ctx* ctx, const float* cand_returns, int G, int n_local, const float* actions,
                              int m, float* mean_io, float* var_io, int32_t* elites_out, void* stream) {
    CADM_REQUIRE(ctx && cand_returns && actions && mean_io && var_io && G > 0 && n_local > 0 && m > 0,
                 "cadm_cem_refit: bad arguments");
    CADM_ON_DEVICE(ctx);
    return cadm_launch_refit(ctx, cand_returns, nullptr, G, n_local, actions, m, mean_io, var_io, mean_io, var_io, elites_out,
                             nullptr, (hipStream_t)stream, nullptr);
}

extern "C" int cadm_cem_refit_regen(cadm_ctx* ctx, const float* cand_returns, int G, int n_local, int m, float* mean_io, float* var_io,
                                    uint32_t seed, uint32_t call, int it, int32_t* elites_out, void* stream) {
    CADM_REQUIRE(ctx && cand_returns && mean_io && var_io && G > 0 && n_local > 0 && m > 0, "cadm_cem_refit_regen: bad arguments");
    CADM_ON_DEVICE(ctx);
    RefitRegen rg{};
    rg.on = 1; rg.seed = seed; rg.call = call; rg.it = it; rg.my_rank = -1;
    return cadm_launch_refit(ctx, cand_returns, nullptr, G, n_local, nullptr, m, mean_io, var_io, mean_io, var_io, elites_out,
                             nullptr, (hipStream_t)stream, &rg);
}

extern "C" int cadm_rs_select(cadm_ctx* ctx, const float* cand_returns, int G, int n_local, const float* actions,
                              int m, float* first_action_out, int32_t* best_out, void* stream) {
    CADM_REQUIRE(ctx && cand_returns && actions && first_action_out && G > 0 && n_local > 0 && m > 0,
                 "cadm_rs_select: bad arguments");
    CADM_ON_DEVICE(ctx);
    hipLaunchKernelGGL(rs_select_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, cand_returns, G, n_local, actions,
                       m, ctx->H, ctx->A, first_action_out, best_out);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}

int cadm_launch_clip(const float* in, float* out, int total, float lo, float hi, int do_clip, hipStream_t s) {
    hipLaunchKernelGGL(clip_kernel, dim3((total + 255) / 256), dim3(256), 0, s, in, out, total, lo, hi, do_clip);
    CADM_CHECK_HIP(hipGetLastError());
    return CADM_OK;
}
