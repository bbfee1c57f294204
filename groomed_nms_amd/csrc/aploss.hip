// aploss.hip -- the after-NMS AP loss on gfx950: the immediate consumer of the rescored scores.
//
// Reference: lib/loss/aploss.py:14-87 (AP-loss, Chen et al. CVPR 2019), called once per image on the NMS output
// (lib/loss/rpn_3d.py:1117-1131).  The reference walks the positives in ascending-logit order in a PYTHON loop, every
// trip doing O(N) tensor ops (:50-68); loss and gradient both come out of forward (:69-78), backward only scales (:80-85).
//
// Here: one workgroup per image, everything in LDS.
//   1. classify + compact (packed block scan): positives -> sortable keys, "valid" negatives (logit >= min positive - delta, :32-35)
//   2. sort the positive VALUES (block_sort, u32 keys; ties need no index: equal logits get equal precision and add equal terms)
//   3. per positive p (one thread each): a_p = sum_k clamp((fg_k - x_p)/(2 delta) + 0.5, 0, 1) + 0.5,  b_p likewise over the valid
//      negatives (:52-60), double accumulators
//   4. running maximum of cur_p = a_p/(a_p+b_p) over the sorted order (:63-66) = an inclusive max-scan; the rescale factor of
//      the trips that do not raise it
//   5. per valid negative j (one thread each): grad_j = sum_p [clamp(...)/(a_p+b_p)] * scale_p, accumulated in fp32 IN the
//      reference's order (ascending positives, :61-67)
//   6. grad[positives] = -(1 - prec)/F, grad[negatives] = grad_j/F, loss = 1 - mean(prec)  (:70-78)
// delta is 1.0 whatever the caller asks for, as in the reference (:16).
#include <cstdlib>
#include "nms_kernels.h"

namespace {

using namespace gnms;

constexpr int kApMaxN = 4096;      // LDS budget: 6 arrays x 4 B x N = 96 KiB
constexpr int kApThreads = 1024;

__device__ __forceinline__ unsigned asc_key(float v) {            // ascending-sortable image of a float (NaN last)
    if (v != v) return 0xfffffffeu;
    const unsigned u = __float_as_uint(v + 0.0f);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float asc_key_decode(unsigned k) {
    if (k == 0xfffffffeu) return __uint_as_float(0x7fc00000u);
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float rank_term(float v, float x, float two_delta) {
    const float t = (v - x) / two_delta + 0.5f;                     // :52-53, :55-56
    return t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
}

// kApE elements per thread: the workgroup ranks up to P = 1024 * kApE boxes (LDS arrays and the sort are sized by P).
template <int kApE>
__global__ __launch_bounds__(kApThreads) void aploss_kernel(const float* __restrict__ logits, const float* __restrict__ targets, int N,
                                                      const int* __restrict__ counts, float positive_label, float negative_label,
                                                      float* __restrict__ loss, float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int P = kApThreads * kApE;
    unsigned* keys = reinterpret_cast<unsigned*>(smem);               // [P] sorted positive logits (as keys)
    float* bgv = reinterpret_cast<float*>(keys + P);                  // [P] valid negative logits, compacted in index order
    float* denom = bgv + P;                                           // [P] a_p + b_p
    float* scale = denom + P;                                         // [P] rescale factor of trip p (1 when the maximum rises)
    float* mprec = scale + P;                                         // [P] running maximum = prec of position p
    float* bgg = mprec + P;                                           // [P] gradient of the compacted negatives
    __shared__ float red_f[16];
    __shared__ unsigned long long red_u[16];
    __shared__ double red_d[16];
    const int b = blockIdx.x;
    const int n = gnms_count(counts, b, N);
    const float* lg = logits + (size_t)b * N;
    const float* tg = targets + (size_t)b * N;
    float* gr = grad + (size_t)b * N;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float two_delta = 2.0f;                                     // delta = 1.0 (:16)

    // ---- pass 1: max(targets) (:26) and the smallest positive logit (:32) ----
    float v[kApE], tv[kApE];
    float tmax = -INFINITY, fmin = INFINITY;
#pragma unroll
    for (int e = 0; e < kApE; ++e) {
        const int i = t * kApE + e;
        v[e] = 0.0f; tv[e] = 0.0f;
        if (i < n) {
            v[e] = lg[i]; tv[e] = tg[i];
            tmax = fmaxf(tmax, tv[e]);
            if (tv[e] == positive_label) fmin = fminf(fmin, v[e]);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { tmax = fmaxf(tmax, __shfl_xor(tmax, off, 64)); fmin = fminf(fmin, __shfl_xor(fmin, off, 64)); }
    if (lane == 0) { red_f[wave] = tmax; red_d[wave] = (double)fmin; }
    for (int i = t; i < N; i += blockDim.x) gr[i] = 0.0f;            // :18 grad = zeros
    __syncthreads();
    tmax = red_f[0]; fmin = (float)red_d[0];
    for (int w = 1; w < 16; ++w) { tmax = fmaxf(tmax, red_f[w]); fmin = fminf(fmin, (float)red_d[w]); }
    __syncthreads();
    if (n == 0 || !(tmax > 0.0f) || fmin == INFINITY) {              // no positives (:26-28): loss 0, zero gradient
        if (t == 0) loss[b] = 0.0f;
        return;
    }
    const float threshold_logit = fmin - 1.0f;                        // :32

    // ---- pass 2: classify, compact (positives -> keys, valid negatives -> bgv) ----
    int cls[kApE];                                                    // 1 positive, 2 valid negative, 0 neither
    unsigned long long packed = 0;                                    // positives | negatives << 32
#pragma unroll
    for (int e = 0; e < kApE; ++e) {
        const int i = t * kApE + e;
        cls[e] = 0;
        if (i < n) {
            if (tv[e] == positive_label) cls[e] = 1;
            else if (tv[e] == negative_label && v[e] >= threshold_logit) cls[e] = 2;   // :35
        }
        packed += (cls[e] == 1) ? 1ull : (cls[e] == 2 ? (1ull << 32) : 0ull);
    }
    unsigned long long inc = packed;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long u = shfl_up_u64(inc, off);
        if (lane >= off) inc += u;
    }
    if (lane == 63) red_u[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, total = 0;
    for (int w = 0; w < 16; ++w) { const unsigned long long u = red_u[w]; if (w < wave) base += u; total += u; }
    const int F = (int)(total & 0xffffffffu), G = (int)(total >> 32);
    unsigned long long run = base + inc - packed;
    int pos_of[kApE];                                                 // compact position of this thread's elements
    unsigned r[kApE];
#pragma unroll
    for (int e = 0; e < kApE; ++e) r[e] = ~0u;
#pragma unroll
    for (int e = 0; e < kApE; ++e) {
        pos_of[e] = -1;
        if (cls[e] == 1) { pos_of[e] = (int)(run & 0xffffffffu); run += 1ull; }
        else if (cls[e] == 2) { pos_of[e] = (int)(run >> 32); bgv[pos_of[e]] = v[e]; run += 1ull << 32; }
    }
    // positives into the sort: scatter them to keys[] by compact position first, then load thread-contiguous
    for (int i = t; i < P; i += blockDim.x) keys[i] = ~0u;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kApE; ++e) if (cls[e] == 1) keys[pos_of[e]] = asc_key(v[e]);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kApE; ++e) r[e] = keys[t * kApE + e];
    __syncthreads();
    block_sort<kApE, unsigned>(r, keys, P);                      // ascending positive logits (:47)

    // ---- pass 3: a_p, b_p per positive position: one wave per positive, lanes stride the positives and the negatives ----
    for (int p = wave; p < F; p += kApThreads / 64) {
        const float x = asc_key_decode(keys[p]);
        double sa = 0.0, sb = 0.0;
        for (int k = lane; k < F; k += 64) sa += (double)rank_term(asc_key_decode(keys[k]), x, two_delta);
        for (int j = lane; j < G; j += 64) sb += (double)rank_term(bgv[j], x, two_delta);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { sa += __shfl_xor(sa, off, 64); sb += __shfl_xor(sb, off, 64); }
        if (lane == 0) {
            const float a = (float)sa + 0.5f;                         // :58
            const float bsum = (float)sb;                             // :60
            denom[p] = a + bsum;
            mprec[p] = a / (a + bsum);                                // current_prec (:62); becomes the running maximum below
        }
    }
    __syncthreads();
    // ---- pass 4: running maximum over the sorted order (:63-66) ----
    {
        float cur[kApE], loc[kApE];
        float m = 0.0f;                                               // max_prec starts at 0 (:48)
#pragma unroll
        for (int e = 0; e < kApE; ++e) {
            const int p = t * kApE + e;
            cur[e] = (p < F) ? mprec[p] : 0.0f;
            m = (m <= cur[e]) ? cur[e] : m;                           // same comparison as :63
            loc[e] = m;
        }
        float incm = m;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float u = __shfl_up(incm, off, 64);
            if (lane >= off) incm = fmaxf(incm, u);
        }
        float before = __shfl_up(incm, 1, 64);
        if (lane == 0) before = 0.0f;
        if (lane == 63) red_f[wave] = incm;
        __syncthreads();
        float carry = 0.0f;
        for (int w = 0; w < wave; ++w) carry = fmaxf(carry, red_f[w]);
        float prev = fmaxf(before, carry);                            // running maximum before this thread's first position
#pragma unroll
        for (int e = 0; e < kApE; ++e) {
            const int p = t * kApE + e;
            if (p < F) {
                const bool rises = prev <= cur[e];                    // :63
                scale[p] = rises ? 1.0f : (1.0f - prev) / (1.0f - cur[e]);   // :66
                prev = rises ? cur[e] : prev;
                mprec[p] = prev;                                      // prec[ii] = max_prec (:68)
            }
        }
    }
    __syncthreads();
    // ---- pass 5: gradient of every valid negative, terms added in ascending-positive order (:61-67) ----
    for (int j = t; j < G; j += blockDim.x) {
        const float vj = bgv[j];
        float g = 0.0f;
        for (int p = 0; p < F; ++p) {
            float term = rank_term(vj, asc_key_decode(keys[p]), two_delta) / denom[p];   // :61
            const float sc = scale[p];
            if (sc != 1.0f) term *= sc;                                                  // :66 (x * 1.0f == x anyway)
            g += term;                                                                   // :67
        }
        bgg[j] = g;
    }
    __syncthreads();
    // ---- pass 6: outputs ----
    const float fnum = (float)(F > 1 ? F : 1);                        // :73
    double sp = 0.0;
    for (int p = t; p < F; p += blockDim.x) sp += (double)mprec[p];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sp += __shfl_xor(sp, off, 64);
    if (lane == 0) red_d[wave] = sp;
#pragma unroll
    for (int e = 0; e < kApE; ++e) {
        const int i = t * kApE + e;
        if (cls[e] == 1) {
            const int p = lower_bound_lds<unsigned>(keys, F, asc_key(v[e]));   // any position with this logit carries the same prec
            gr[i] = (-(1.0f - mprec[p])) / fnum;                               // :71, :75
        } else if (cls[e] == 2) {
            gr[i] = bgg[pos_of[e]] / fnum;                                     // :70, :75
        }
    }
    __syncthreads();
    if (t == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red_d[w];
        loss[b] = 1.0f - (float)s / fnum;                             // :77-78
    }
}


// ------------------------------------------------------------------------------------------------
// Large images (N > 4096, up to GNMS_MAX_BOXES) and many positives: the same computation spread over the machine.
//   ap_prepare_kernel  one workgroup per image: classify, compact, sort the positive values (passes 1-2 above) into global scratch
//   ap_rank_kernel     one WAVE per positive, 16 per workgroup, all positives of all images at once: a_p, b_p (pass 3) -- the
//                      O(F (F + G)) part that took 0.9 ms on one CU at F = 1024, N = 4096
//   ap_scan_kernel     one workgroup per image: running maximum, rescale factors, loss, gradient of the positives (passes 4, 6)
//   ap_neg_grad_kernel 256 valid negatives per workgroup: their gradients, positives walked in ascending order through LDS tiles
//                      (pass 5: the reference's order of additions)
// Scratch per image (stream-ordered temporary): 8 N words + 4.
// ------------------------------------------------------------------------------------------------
struct ApScratch {
    unsigned* keys;      // [N] sorted positive logits (as keys)
    unsigned* poskey;    // [N] key of the k-th positive in index order
    int* posidx;         // [N] box index of the k-th positive in index order
    float* bgv;          // [N] valid negative logits in index order
    int* negidx;         // [N] their box indices
    float* denom;        // [N] a_p + b_p
    float* mprec;        // [N] current_prec, then the running maximum
    float* scale;        // [N]
    int* meta;           // [4] F, G, has positives
};
__host__ __device__ inline size_t ap_scratch_words(int N) { return (size_t)8 * N + 4; }
__device__ __forceinline__ ApScratch ap_scratch(float* base, int N, int b) {
    unsigned* p = reinterpret_cast<unsigned*>(base) + (size_t)b * ap_scratch_words(N);
    ApScratch S;
    S.keys = p; S.poskey = p + N; S.posidx = reinterpret_cast<int*>(p + 2 * (size_t)N); S.bgv = reinterpret_cast<float*>(p + 3 * (size_t)N);
    S.negidx = reinterpret_cast<int*>(p + 4 * (size_t)N); S.denom = reinterpret_cast<float*>(p + 5 * (size_t)N);
    S.mprec = reinterpret_cast<float*>(p + 6 * (size_t)N); S.scale = reinterpret_cast<float*>(p + 7 * (size_t)N);
    S.meta = reinterpret_cast<int*>(p + 8 * (size_t)N);
    return S;
}

template <int kApE>
__global__ __launch_bounds__(kApThreads) void ap_prepare_kernel(const float* __restrict__ logits, const float* __restrict__ targets, int N,
                                                                const int* __restrict__ counts, float positive_label, float negative_label,
                                                                float* __restrict__ loss, float* __restrict__ grad, float* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int P = kApThreads * kApE;
    unsigned* keys = reinterpret_cast<unsigned*>(smem);               // [P]
    __shared__ float red_f[16];
    __shared__ unsigned long long red_u[16];
    __shared__ double red_d[16];
    const int b = blockIdx.x;
    const int n = gnms_count(counts, b, N);
    const float* lg = logits + (size_t)b * N;
    const float* tg = targets + (size_t)b * N;
    float* gr = grad + (size_t)b * N;
    ApScratch S = ap_scratch(scratch, N, b);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float v[kApE], tv[kApE];
    float tmax = -INFINITY, fmin = INFINITY;
#pragma unroll
    for (int e = 0; e < kApE; ++e) {
        const int i = t * kApE + e;
        v[e] = 0.0f; tv[e] = 0.0f;
        if (i < n) {
            v[e] = lg[i]; tv[e] = tg[i];
            tmax = fmaxf(tmax, tv[e]);
            if (tv[e] == positive_label) fmin = fminf(fmin, v[e]);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { tmax = fmaxf(tmax, __shfl_xor(tmax, off, 64)); fmin = fminf(fmin, __shfl_xor(fmin, off, 64)); }
    if (lane == 0) { red_f[wave] = tmax; red_d[wave] = (double)fmin; }
    for (int i = t; i < N; i += blockDim.x) gr[i] = 0.0f;
    __syncthreads();
    tmax = red_f[0]; fmin = (float)red_d[0];
    for (int w = 1; w < 16; ++w) { tmax = fmaxf(tmax, red_f[w]); fmin = fminf(fmin, (float)red_d[w]); }
    __syncthreads();
    if (n == 0 || !(tmax > 0.0f) || fmin == INFINITY) {              // no positives (:26-28)
        if (t == 0) { loss[b] = 0.0f; S.meta[0] = 0; S.meta[1] = 0; S.meta[2] = 0; }
        return;
    }
    const float threshold_logit = fmin - 1.0f;
    int cls[kApE];
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < kApE; ++e) {
        const int i = t * kApE + e;
        cls[e] = 0;
        if (i < n) {
            if (tv[e] == positive_label) cls[e] = 1;
            else if (tv[e] == negative_label && v[e] >= threshold_logit) cls[e] = 2;
        }
        packed += (cls[e] == 1) ? 1ull : (cls[e] == 2 ? (1ull << 32) : 0ull);
    }
    unsigned long long inc = packed;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long u = shfl_up_u64(inc, off);
        if (lane >= off) inc += u;
    }
    if (lane == 63) red_u[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, total = 0;
    for (int w = 0; w < 16; ++w) { const unsigned long long u = red_u[w]; if (w < wave) base += u; total += u; }
    const int F = (int)(total & 0xffffffffu), G = (int)(total >> 32);
    unsigned long long run = base + inc - packed;
    for (int i = t; i < P; i += blockDim.x) keys[i] = ~0u;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kApE; ++e) {
        const int i = t * kApE + e;
        if (cls[e] == 1) {
            const int k = (int)(run & 0xffffffffu);
            const unsigned key = asc_key(v[e]);
            keys[k] = key; S.poskey[k] = key; S.posidx[k] = i;
            run += 1ull;
        } else if (cls[e] == 2) {
            const int j = (int)(run >> 32);
            S.bgv[j] = v[e]; S.negidx[j] = i;
            run += 1ull << 32;
        }
    }
    __syncthreads();
    // the F positive keys sit at the front, padded with ~0: sort the next power of two above F, not all P slots (round 5: F = 4096 of
    // N = 16384 sorted 16384 slots on one CU, ~100 us of the launch)
    bool done = false;
#define GNMS_AP_SORT(EE)                                                                       \
    if constexpr (kApE > EE) {                                                                 \
        if (!done && F <= kApThreads * EE) {                                                   \
            unsigned rs[EE];                                                                   \
            _Pragma("unroll") for (int e = 0; e < EE; ++e) rs[e] = keys[t * EE + e];           \
            __syncthreads();                                                                   \
            block_sort<EE, unsigned>(rs, keys, kApThreads * EE);                               \
            done = true;                                                                       \
        }                                                                                      \
    }
    GNMS_AP_SORT(1) GNMS_AP_SORT(2) GNMS_AP_SORT(4) GNMS_AP_SORT(8)
#undef GNMS_AP_SORT
    if (!done) {
        unsigned r[kApE];
#pragma unroll
        for (int e = 0; e < kApE; ++e) r[e] = keys[t * kApE + e];
        __syncthreads();
        block_sort<kApE, unsigned>(r, keys, P);
    }
    for (int k = t; k < F; k += blockDim.x) S.keys[k] = keys[k];
    if (t == 0) { S.meta[0] = F; S.meta[1] = G; S.meta[2] = 1; }
}

// (round 5: kRankPPW positives per wave -- every value loaded once for all of them; one positive per wave streamed 64 KiB through the L2 for each
// of 4096 positives at N = 16384 and was bound by that.  The terms are summed in fp32 for 32 trips at a time -- at most 32 terms of [0, 1] per
// lane: exact to 2^-19 -- and carried on in double.)
// (with few positives -- fewer than a wave per SIMD of the machine -- one positive per wave keeps more waves in flight: kRankPPW = 1 below 2048)
template <int kRankPPW>
__device__ __forceinline__ void ap_rank_body(const ApScratch& S, const int F, const int G) {
    const int lane = threadIdx.x & 63;
    const int p0 = (blockIdx.x * 16 + (threadIdx.x >> 6)) * kRankPPW;
    if (p0 >= F) return;
    float x[kRankPPW];
#pragma unroll
    for (int u = 0; u < kRankPPW; ++u) x[u] = asc_key_decode(S.keys[min(p0 + u, F - 1)]);
    double sa[kRankPPW], sb[kRankPPW];
#pragma unroll
    for (int u = 0; u < kRankPPW; ++u) { sa[u] = 0.0; sb[u] = 0.0; }
    for (int k0 = 0; k0 < F; k0 += 64 * 32) {
        float acc[kRankPPW];
#pragma unroll
        for (int u = 0; u < kRankPPW; ++u) acc[u] = 0.0f;
        const int kend = min(F, k0 + 64 * 32);
        for (int k = k0 + lane; k < kend; k += 64) {
            const float v = asc_key_decode(S.keys[k]);
#pragma unroll
            for (int u = 0; u < kRankPPW; ++u) acc[u] += rank_term(v, x[u], 2.0f);
        }
#pragma unroll
        for (int u = 0; u < kRankPPW; ++u) sa[u] += (double)acc[u];
    }
    for (int j0 = 0; j0 < G; j0 += 64 * 32) {
        float acc[kRankPPW];
#pragma unroll
        for (int u = 0; u < kRankPPW; ++u) acc[u] = 0.0f;
        const int jend = min(G, j0 + 64 * 32);
        for (int j = j0 + lane; j < jend; j += 64) {
            const float v = S.bgv[j];
#pragma unroll
            for (int u = 0; u < kRankPPW; ++u) acc[u] += rank_term(v, x[u], 2.0f);
        }
#pragma unroll
        for (int u = 0; u < kRankPPW; ++u) sb[u] += (double)acc[u];
    }
#pragma unroll
    for (int u = 0; u < kRankPPW; ++u) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { sa[u] += __shfl_xor(sa[u], off, 64); sb[u] += __shfl_xor(sb[u], off, 64); }
        if (lane == 0 && p0 + u < F) {
            const float a = (float)sa[u] + 0.5f;
            const float bsum = (float)sb[u];
            S.denom[p0 + u] = a + bsum;
            S.mprec[p0 + u] = a / (a + bsum);
        }
    }
}
__global__ __launch_bounds__(1024) void ap_rank_kernel(int N, float* __restrict__ scratch) {
    const ApScratch S = ap_scratch(scratch, N, blockIdx.y);
    const int F = S.meta[0], G = S.meta[1];
    if (F >= 2048) ap_rank_body<4>(S, F, G);
    else ap_rank_body<1>(S, F, G);
}

template <int kApE>
__global__ __launch_bounds__(kApThreads) void ap_scan_kernel(int N, float* __restrict__ scratch, float* __restrict__ loss, float* __restrict__ grad) {
    __shared__ float red_f[16];
    __shared__ double red_d[16];
    const int b = blockIdx.x;
    ApScratch S = ap_scratch(scratch, N, b);
    if (!S.meta[2]) return;
    const int F = S.meta[0];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float cur[kApE];
    float m = 0.0f;
#pragma unroll
    for (int e = 0; e < kApE; ++e) {
        const int p = t * kApE + e;
        cur[e] = (p < F) ? S.mprec[p] : 0.0f;
        m = (m <= cur[e]) ? cur[e] : m;
    }
    float incm = m;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float u = __shfl_up(incm, off, 64);
        if (lane >= off) incm = fmaxf(incm, u);
    }
    float before = __shfl_up(incm, 1, 64);
    if (lane == 0) before = 0.0f;
    if (lane == 63) red_f[wave] = incm;
    __syncthreads();
    float carry = 0.0f;
    for (int w = 0; w < wave; ++w) carry = fmaxf(carry, red_f[w]);
    float prev = fmaxf(before, carry);
    double sp = 0.0;
#pragma unroll
    for (int e = 0; e < kApE; ++e) {
        const int p = t * kApE + e;
        if (p < F) {
            const bool rises = prev <= cur[e];
            S.scale[p] = rises ? 1.0f : (1.0f - prev) / (1.0f - cur[e]);
            prev = rises ? cur[e] : prev;
            S.mprec[p] = prev;
            sp += (double)prev;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sp += __shfl_xor(sp, off, 64);
    if (lane == 0) red_d[wave] = sp;
    __syncthreads();                                                   // mprec[] complete (global, same workgroup)
    const float fnum = (float)(F > 1 ? F : 1);
    float* gr = grad + (size_t)b * N;
    for (int k = t; k < F; k += blockDim.x) {                          // gradient of the positives: any position with this logit carries its prec
        const unsigned key = S.poskey[k];
        int lo = 0, hi = F;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (S.keys[mid] < key) lo = mid + 1; else hi = mid; }
        gr[S.posidx[k]] = (-(1.0f - S.mprec[lo])) / fnum;
    }
    if (t == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red_d[w];
        loss[b] = 1.0f - (float)s / fnum;
    }
}

__global__ __launch_bounds__(256) void ap_neg_grad_kernel(int N, float* __restrict__ scratch, float* __restrict__ grad) {
    __shared__ float2 txw[256];                                        // (logit of the positive, scale / denominator)
    const int b = blockIdx.y;
    ApScratch S = ap_scratch(scratch, N, b);
    const int F = S.meta[0], G = S.meta[1];
    if ((int)blockIdx.x * 256 >= G) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const float vj = (j < G) ? S.bgv[j] : 0.0f;
    float g = 0.0f;
    for (int p0 = 0; p0 < F; p0 += 256) {
        const int p = p0 + threadIdx.x;
        // (round 5: the positive's weight scale / denominator once per positive, not a division per (negative, positive) pair -- G x F of them,
        // 50 M per image at F = 4096, N = 16384, were 0.4 ms of the launch; the product differs from divide-then-scale by an ulp, far inside
        // the tolerance the loss is held to against the reference's own fp32 sums)
        if (p < F) txw[threadIdx.x] = make_float2(asc_key_decode(S.keys[p]), S.scale[p] / S.denom[p]);
        __syncthreads();
        const int np = min(256, F - p0);
        for (int q = 0; q < np; ++q) {                                 // ascending positives: the reference's order of additions (:61-67)
            const float2 xw = txw[q];
            g += rank_term(vj, xw.x, 2.0f) * xw.y;
        }
        __syncthreads();
    }
    const float fnum = (float)(F > 1 ? F : 1);
    if (j < G) grad[(size_t)b * N + S.negidx[j]] = g / fnum;
}

}  // namespace

extern "C" int gnms_aploss(const float* logits, const float* targets, int B, int N, const int32_t* counts, float positive_label,
                           float negative_label, float* loss, float* grad, void* stream) {
    GNMS_CHECK_ARG(B >= 0 && N >= 0, "gnms_aploss: negative size");
    if (B == 0) return GNMS_OK;
    GNMS_CHECK_ARG(loss != nullptr, "gnms_aploss: loss is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) { GNMS_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float) * B, st)); return GNMS_OK; }
    if (N > GNMS_MAX_BOXES) {
        gnms_set_error("gnms_aploss: N=%d exceeds GNMS_MAX_BOXES=%d", N, GNMS_MAX_BOXES);
        return GNMS_ERR_UNSUPPORTED;
    }
    GNMS_CHECK_ARG(logits && targets && grad, "gnms_aploss: null pointer");
    // One workgroup per image ranks up to 4096 boxes entirely in LDS; larger images -- and smaller batches of images from 2048 boxes,
    // whose F x N product makes the one-CU version slow -- run the four-kernel version that spreads the positives over the machine.
    if (N > kApMaxN || (N >= 2048 && B <= 64)) {
        gnms_async_buffer scratch_buf;                                // returned to the pool on every exit, the early error returns included
        GNMS_CHECK_HIP(scratch_buf.alloc((size_t)B * ap_scratch_words(N) * sizeof(float), st));
        float* scratch = scratch_buf.as<float>();
        int P2 = 1024;
        while (P2 < N) P2 <<= 1;
        const int E = P2 / kApThreads;
        const size_t lds = (size_t)P2 * 4;
#define GNMS_AP_PREP(EE)                                                                                                              \
    do {                                                                                                                              \
        if (lds > 64 * 1024)                                                                                                          \
            GNMS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ap_prepare_kernel<EE>),                                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                \
        ap_prepare_kernel<EE><<<B, kApThreads, lds, st>>>(logits, targets, N, counts, positive_label, negative_label, loss, grad, scratch); \
    } while (0)
        switch (E) { case 1: GNMS_AP_PREP(1); break; case 2: GNMS_AP_PREP(2); break; case 4: GNMS_AP_PREP(4); break; case 8: GNMS_AP_PREP(8); break;
                     default: GNMS_AP_PREP(16); break; }
#undef GNMS_AP_PREP
        ap_rank_kernel<<<dim3(gnms_div_up(N, 16), B), 1024, 0, st>>>(N, scratch);
        switch (E) { case 1: ap_scan_kernel<1><<<B, kApThreads, 0, st>>>(N, scratch, loss, grad); break;
                     case 2: ap_scan_kernel<2><<<B, kApThreads, 0, st>>>(N, scratch, loss, grad); break;
                     case 4: ap_scan_kernel<4><<<B, kApThreads, 0, st>>>(N, scratch, loss, grad); break;
                     case 8: ap_scan_kernel<8><<<B, kApThreads, 0, st>>>(N, scratch, loss, grad); break;
                     default: ap_scan_kernel<16><<<B, kApThreads, 0, st>>>(N, scratch, loss, grad); break; }
        ap_neg_grad_kernel<<<dim3(gnms_div_up(N, 256), B), 256, 0, st>>>(N, scratch, grad);
        GNMS_CHECK_LAUNCH();
        GNMS_CHECK_HIP(scratch_buf.release());
        return GNMS_OK;
    }
    if (N > 2 * kApThreads)      // > 64 KiB of dynamic LDS; set per call: the attribute is per device and the call is cheap
        GNMS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(aploss_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kApThreads * 4 * 4 * 6));
#define GNMS_APLOSS_LAUNCH(E) \
    aploss_kernel<E><<<B, kApThreads, (size_t)kApThreads * E * 4 * 6, st>>>(logits, targets, N, counts, positive_label, negative_label, loss, grad)
    if (N <= kApThreads) GNMS_APLOSS_LAUNCH(1);
    else if (N <= 2 * kApThreads) GNMS_APLOSS_LAUNCH(2);
    else GNMS_APLOSS_LAUNCH(4);
#undef GNMS_APLOSS_LAUNCH
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
