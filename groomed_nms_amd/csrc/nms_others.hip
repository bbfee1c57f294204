// nms_others.hip -- the comparison baselines of the reference's lib/nms_others.py on gfx950:
//   navneeth_soft_nms  lib/nms_others.py:6-116   Soft-NMS (Bodla et al.) with in-place row swaps and a `keep_orig` index array
//   girshick_nms       lib/nms_others.py:119-150 greedy NMS with a pixel `shift`  -> classic_nms.hip (gnms_nms_sorted_shift)
//
// Soft-NMS is sequential in its OUTER loop by construction (iteration i selects the best remaining score after all earlier decays),
// but everything inside one iteration is independent per box: the arg max, the decay of every live score against the selected box,
// and the decision which boxes fall below the threshold.  What looks order dependent -- the reference discards a box by swapping the
// LAST live row into its place and examining that row next (:98-110) -- has a closed form: with S survivors among the live slots
// (i, live), the first S slots keep their own survivors, and the h-th discarded slot among them (ascending) receives the h-th
// survivor counted from the END of the old live range.  So one iteration is: block arg max -> swap -> one decay pass -> one block
// scan -> one permutation, ~6 barriers, on ONE workgroup (the slot arrays live in global memory / L2; N <= GNMS_MAX_BOXES).
// The reference is a Python loop of O(N^2) scalar operations; N = 4096 takes it ~20 s, this kernel ~10 ms.
//
// Arithmetic follows the reference's types: geometry and stored scores in the array's dtype T (fp64 for the NumPy float64 input the
// reference's own test feeds, test/test_differentiable_nms_forward.py:111; fp32 for float32 input), the union `float(...)`-ed to a
// double (:78), overlap, weight and the product weight * score in double, rounded to T on the store (:93) -- NumPy 1.14 scalar
// promotion, the reference's pinned version (dependencies/conda.txt); for float64 input (the tested case) every step is fp64.
#include "nms_kernels.h"

namespace {

using namespace gnms;

constexpr int kNone = 0x7fffffff;
// candidate (s, p) beats (bs, bp): a greater score, or the same score in an earlier slot
__device__ __forceinline__ bool better(double s, int p, double bs, int bp) {
    if (p == kNone) return false;
    if (bp == kNone) return true;
    if (bs < s) return true;
    if (s < bs) return false;
    return p < bp;
}

template <typename T>
__global__ __launch_bounds__(1024) void soft_nms_kernel(const T* __restrict__ boxes, int n, int dim, double sigma, double Nt, double threshold,
                                                        int method, T shift, int* __restrict__ sid, T* __restrict__ ss, int* __restrict__ tmp_id,
                                                        T* __restrict__ tmp_s, long long* __restrict__ keep, int* __restrict__ num_out) {
    __shared__ double red_s[16];
    __shared__ int red_p[16];
    __shared__ int wave_cnt[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int p = t; p < n; p += 1024) { sid[p] = p; ss[p] = boxes[(size_t)p * dim + 4]; }     // keep_orig = arange (:15)
    __syncthreads();
    int live = n;
    for (int i = 0; i < live; ++i) {
        // ---- the best remaining score; the FIRST maximum wins (strict '<' at :32).  NaN never compares greater: a NaN in slot i
        //      stays selected (:19-20), a NaN further back is never selected ----
        double bs = 0.0;
        int bp = kNone;
        for (int p = i + t; p < live; p += 1024) {
            const double v = (double)ss[p];
            if (v == v && (bp == kNone || bs < v)) { bs = v; bp = p; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double os = __shfl_xor(bs, off, 64);
            const int op = __shfl_xor(bp, off, 64);
            if (better(os, op, bs, bp)) { bs = os; bp = op; }
        }
        if (lane == 0) { red_s[wave] = bs; red_p[wave] = bp; }
        __syncthreads();
        if (t == 0) {
            double b = red_s[0];
            int p = red_p[0];
            for (int w = 1; w < 16; ++w)
                if (better(red_s[w], red_p[w], b, p)) { b = red_s[w]; p = red_p[w]; }
            const T si = ss[i];
            if (si != si || p == kNone) p = i;
            const int idi = sid[i], idp = sid[p];                     // swap slot i with the slot of the maximum (:37-60)
            const T sp = ss[p];
            sid[i] = idp; sid[p] = idi;
            ss[i] = sp; ss[p] = si;
        }
        __syncthreads();
        // ---- decay every live score against the selected box (:64-96); flag the ones that fall below the threshold ----
        const int bi = sid[i];
        const T tx1 = boxes[(size_t)bi * dim], ty1 = boxes[(size_t)bi * dim + 1], tx2 = boxes[(size_t)bi * dim + 2], ty2 = boxes[(size_t)bi * dim + 3];
        const int L = live - i - 1;
        const int chunk = (L + 1023) / 1024;                          // <= 16 slots per thread, contiguous: the scan keeps slot order
        const int p0 = i + 1 + t * chunk;
        unsigned dead = 0u;                                           // bit e: slot p0 + e is discarded
        int nsurv = 0;
        for (int e = 0; e < chunk; ++e) {
            const int p = p0 + e;
            if (p >= live) break;
            const int bj = sid[p];
            const T x1 = boxes[(size_t)bj * dim], y1 = boxes[(size_t)bj * dim + 1], x2 = boxes[(size_t)bj * dim + 2], y2 = boxes[(size_t)bj * dim + 3];
            bool gone = false;
            const T area = (x2 - x1 + shift) * (y2 - y1 + shift);                               // :72
            const T iw = (tx2 < x2 ? tx2 : x2) - (tx1 > x1 ? tx1 : x1) + shift;                 // :73 (Python min / max)
            if (iw > (T)0) {
                const T ih = (ty2 < y2 ? ty2 : y2) - (ty1 > y1 ? ty1 : y1) + shift;             // :75
                if (ih > (T)0) {
                    const double ua = (double)((tx2 - tx1 + shift) * (ty2 - ty1 + shift) + area - iw * ih);   // :77 float(...)
                    const double ov = (double)(iw * ih) / ua;                                    // :78
                    double weight;
                    if (method == 1) weight = (ov > Nt) ? 1.0 - ov : 1.0;                        // :80-84
                    else if (method == 2) weight = exp(-(ov * ov) / sigma);                      // :85-86
                    else weight = (ov > Nt) ? 0.0 : 1.0;                                         // :87-91
                    const T ns = (T)(weight * (double)ss[p]);                                    // :93
                    ss[p] = ns;
                    gone = (double)ns < threshold;                                                   // :97
                }
            }
            if (gone) dead |= 1u << e; else ++nsurv;
        }
        // ---- block scan of the survivor counts (slot order) ----
        const int inc = (int)gnms_add_scan32((unsigned)nsurv);
        if (lane == 63) wave_cnt[wave] = inc;
        __syncthreads();
        int base = 0, total = 0;
        for (int w = 0; w < 16; ++w) { const int c = wave_cnt[w]; if (w < wave) base += c; total += c; }
        const int S = total;                                           // survivors among the slots (i, live)
        const int bound = i + 1 + S;                                   // new end of the live range
        const int front_surv_before = base + inc - nsurv;              // survivors in the slots before this thread's chunk
        // the survivors at or behind `bound` fill the discarded slots in front of it: the h-th hole (ascending) gets the h-th filler
        // counted from the END (the reference pulls the last live row into every hole, :99-110)
        int sb = front_surv_before;
        for (int e = 0; e < chunk; ++e) {
            const int p = p0 + e;
            if (p >= live) break;
            const bool gone = (dead >> e) & 1u;
            if (!gone) {
                if (p >= bound) {                                      // a filler; rank from the end = (S - 1) - (survivors before this slot)
                    const int drank = (S - 1) - sb;
                    tmp_id[drank] = sid[p];
                    tmp_s[drank] = ss[p];
                }
                ++sb;
            }
        }
        __syncthreads();
        sb = front_surv_before;
        for (int e = 0; e < chunk; ++e) {
            const int p = p0 + e;
            if (p >= live) break;
            const bool gone = (dead >> e) & 1u;
            if (gone && p < bound) {                                   // a hole: its ascending rank = discarded slots before it
                const int h = (p - (i + 1)) - sb;
                sid[p] = tmp_id[h];
                ss[p] = tmp_s[h];
            }
            if (!gone) ++sb;
        }
        __syncthreads();
        live = bound;
    }
    for (int p = t; p < live; p += 1024) keep[p] = (long long)sid[p];   // keep_orig[:N] (:116)
    if (t == 0) *num_out = live;
}

template <typename T>
int run_soft_nms(const void* boxes, int n, int dim, double sigma, double Nt, double threshold, int method, double shift, int64_t* keep,
                 int32_t* num_out, void* workspace, hipStream_t st) {
    char* ws = (char*)workspace;
    const size_t n4 = gnms_align_up((size_t)n * 4, 256), n8 = gnms_align_up((size_t)n * 8, 256);
    int* sid = (int*)ws;
    int* tmp_id = (int*)(ws + n4);
    T* ss = (T*)(ws + 2 * n4);
    T* tmp_s = (T*)(ws + 2 * n4 + n8);
    soft_nms_kernel<T><<<1, 1024, 0, st>>>((const T*)boxes, n, dim, sigma, Nt, threshold, method, (T)shift, sid, ss, tmp_id, tmp_s,
                                          (long long*)keep, num_out);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

}  // namespace

extern "C" size_t gnms_soft_nms_workspace_bytes(int n) {
    if (n <= 0) return 0;
    return 2 * gnms_align_up((size_t)n * 4, 256) + 2 * gnms_align_up((size_t)n * 8, 256);
}

extern "C" int gnms_soft_nms(const void* boxes, int n, int boxes_dim, int is_fp64, double sigma, double Nt, double threshold, int method,
                             double shift, int64_t* keep, int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream) {
    GNMS_CHECK_ARG(n >= 0 && boxes_dim >= 5, "gnms_soft_nms: bad shape (n=%d dim=%d)", n, boxes_dim);
    GNMS_CHECK_ARG(num_out != nullptr, "gnms_soft_nms: num_out is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { GNMS_CHECK_HIP(hipMemsetAsync(num_out, 0, sizeof(int32_t), st)); return GNMS_OK; }
    if (n > GNMS_MAX_BOXES) { gnms_set_error("gnms_soft_nms: n=%d exceeds GNMS_MAX_BOXES=%d", n, GNMS_MAX_BOXES); return GNMS_ERR_UNSUPPORTED; }
    GNMS_CHECK_ARG(boxes && keep && workspace, "gnms_soft_nms: null pointer");
    if (workspace_bytes < gnms_soft_nms_workspace_bytes(n)) { gnms_set_error("gnms_soft_nms: workspace too small"); return GNMS_ERR_WORKSPACE; }
    GNMS_CHECK_ARG((uintptr_t)workspace % 256 == 0, "gnms_soft_nms: workspace must be 256-byte aligned");
    if (is_fp64) return run_soft_nms<double>(boxes, n, boxes_dim, sigma, Nt, threshold, method, shift, keep, num_out, workspace, st);
    return run_soft_nms<float>(boxes, n, boxes_dim, sigma, Nt, threshold, method, shift, keep, num_out, workspace, st);
}
