// classic_nms.hip -- classical hard NMS for gfx950 behind the reference's C symbol `_nms`.
//
// Reference: lib/nms/nms_kernel.cu:24-32 devIoU (+1 pixel), :34-78 nms_kernel (64x64 tiles, one
// thread per row looping 64 columns, strict '>'), :91-144 `_nms` (H2D, kernel, D2H of the whole
// bit matrix, sequential HOST scan), lib/nms/gpu_nms.hpp:1-2.
//
// Here: the bit matrix is built column-major in rank blocks by the same wave64 scheme as the GrooMeD
// bit-matrix kernel (a lane owns 4 candidate-suppressor columns and accumulates a 64-bit word over
// the 64 rows of a rank block; no per-thread 64-iteration loop, no shared memory), only tiles that a
// suppressor can reach are computed (suppressor index < end of the rank block), and the scan runs ON
// the device (leaders_kernel), so only keep[] and the count cross PCIe.
#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>
#include <string.h>
#include "nms_kernels.h"

namespace {

using namespace gnms;

__device__ __forceinline__ float bcastf(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// boxes [n][dim] sorted by score; W[kb][c] bit r = devIoU(box[64 kb + r], box[c]) > thresh
// shift: the pixel convention (x2 - x1 + shift): 1 in nms_kernel.cu:24-32, a parameter of lib/nms_others.py:119 girshick_nms.
// keep_le: 0 = suppress when IoU > thresh (nms_kernel.cu:71); 1 = keep only IoU <= thresh (nms_others.py:146: a NaN overlap suppresses)
// what the leader scan expects of a sort that never ran: boxes arrive sorted, rank == index.  The workgroups of the first row block do it on the side
// (COLS columns each) -- the mask kernels read none of it, the scan runs a launch later
// the counters, the call counter and the hand-off granules of the leader scan (what the sort kernels do for the layer), by ONE workgroup
__device__ __forceinline__ void classic_init_counters(const ImgPtrs& I) {
    if (threadIdx.x < 8) I.misc[threadIdx.x] = (threadIdx.x == 2) ? 1 : 0;
    if (threadIdx.x == 8) I.misc[8] = gnms_next_epoch(I.misc[8]);
    for (int i = threadIdx.x; i < 17 * 32; i += blockDim.x) I.gran[i] = 0ull;
}
template <int COLS>
__device__ __forceinline__ void classic_init_part(int n, char* ws, const gnms_ws_layout& L) {
    ImgPtrs I = img_ptrs(ws, L, 0);
    const int base = blockIdx.x * COLS, end = min(n, base + COLS);
    for (int k = base + (int)threadIdx.x; k < end; k += 256) { I.order[k] = k; I.rankof[k] = k; }
    if (blockIdx.x == 0) classic_init_counters(I);
}

// J columns per lane: a wave's tile is 64 rows x 64 J columns, ~25 VALU instructions per pair (one IEEE division) all on ONE SIMD -- 11 us for
// J = 4.  Up to ~8000 boxes such tiles are fewer than the machine's SIMDs (n = 500: 16 busy waves, 21 us; n = 4096: 24 us), and J = 1 gives
// four times the waves a quarter of the work each (n = 500: 22.5 -> 8.8 us; n = 4096: 24 -> 20 with the 1D grid below); above, J = 4 keeps the box loads amortised.
// UPPER (round 6, n <= GNMS_MAX_BOXES): the tiles whose COLUMN block is not in front of the row block -- W[kb][c] for c >= 64 kb: whom a box of
// block kb suppresses among the later boxes -- instead of those a suppressor can reach.  The overlap is symmetric bit for bit (sums, minima and
// maxima commute), so the two triangles hold the same decisions; this one is what the layer's leader scan on ONE WORKGROUP PER SUPER-BLOCK
// reads (leaders_sb_body: a super-block's table and its pulls from the earlier super-blocks), which replaces the one-workgroup general scan
// -- 23.6 us at n = 4096, ~450 at 16384 -- behind `_nms`.
template <int J, bool UPPER = false>
__global__ __launch_bounds__(256) void classic_mask_kernel(const float* __restrict__ boxes, int n, int dim, float thresh, float shift, int keep_le,
                                                           char* ws, gnms_ws_layout L, int init) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int kb, c0;
    if (J == 1) {
        // a 1D grid over the tiles a suppressor can reach only (column block <= row block: NB (NB + 1) / 2 of them, four per workgroup).  As a 2D
        // grid the idle upper triangle and the busy lower one landed on different CUs -- workgroup id mod 256 keeps a CU in ONE column chunk --
        // and the busiest CU ran four full workgroups: 24 us at n = 4096 where the arithmetic is 11.
        if (init) {
            ImgPtrs I0 = img_ptrs(ws, L, 0);
            for (int k = blockIdx.x * 256 + (int)threadIdx.x; k < n; k += gridDim.x * 256) { I0.order[k] = k; I0.rankof[k] = k; }
            if (blockIdx.x == 0) classic_init_counters(I0);
        }
        const int w = blockIdx.x * 4 + wave, nbk = (n + 63) >> 6;
        if (w >= nbk * (nbk + 1) / 2) return;
        int big = (int)((sqrtf(8.0f * (float)w + 1.0f) - 1.0f) * 0.5f);
        while (big * (big + 1) / 2 > w) --big;
        while ((big + 1) * (big + 2) / 2 <= w) ++big;
        const int small = w - big * (big + 1) / 2;                   // small <= big: the pair of blocks of this tile
        kb = UPPER ? small : big;
        c0 = (UPPER ? big : small) * 64;
    } else {
        if (init && blockIdx.y == 0) classic_init_part<256 * J>(n, ws, L);
        kb = blockIdx.y;
        c0 = (blockIdx.x * 4 + wave) * (64 * J);
    }
    const int k0 = kb * 64;
    if (k0 >= n || c0 >= n) return;
    if (UPPER ? (c0 + 64 * J <= k0) : (c0 >= k0 + 64)) return;      // lower: suppressors come from ranks < k0 + 64; upper: the suppressed from ranks >= k0
    ImgPtrs I = img_ptrs(ws, L, 0);
    float bx1[J], by1[J], bx2[J], by2[J], bs[J];
    int col[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        col[j] = c0 + J * lane + j;
        const float* p = boxes + (size_t)(col[j] < n ? col[j] : n - 1) * dim;
        bx1[j] = p[0]; by1[j] = p[1]; bx2[j] = p[2]; by2[j] = p[3];
        bs[j] = (bx2[j] - bx1[j] + shift) * (by2[j] - by1[j] + shift);         // nms_kernel.cu:30
    }
    const int myr = min(k0 + lane, n - 1);
    const float* q = boxes + (size_t)myr * dim;
    const float rx1 = q[0], ry1 = q[1], rx2 = q[2], ry2 = q[3];
    const float rs = (rx2 - rx1 + shift) * (ry2 - ry1 + shift);                // :29
    const int nrows = min(64, n - k0);
    unsigned lo[J], hi[J];
#pragma unroll
    for (int j = 0; j < J; ++j) lo[j] = hi[j] = 0u;
    // (round 6) the decision without the division where that is safe, as bitmask_boxes_body takes it: with d = fma(-thresh, uni, inter) (one
    // rounding, sign exact) and uni > 0, |d| > guard * uni puts the exact quotient more than 8 ulp of the threshold's magnitude away from it, so
    // its fp32 rounding lies on the same side and `ov > thresh` is `d > 0`; a row with a pair inside the band (or uni <= 0, NaN, inf) divides.
    const float guard = fmaxf(fabsf(thresh), 1.0f) * 9.6e-7f;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
        const float ax1 = bcastf(rx1, r), ay1 = bcastf(ry1, r), ax2 = bcastf(rx2, r), ay2 = bcastf(ry2, r), as = bcastf(rs, r);
        float inter[J], uni[J], d[J];
        bool unsure = false;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            // (v_max / v_min issued directly, the row coordinate from its SGPR: no canonicalising moves; iou3d_pair.h)
            const float left = gnms_iou3d::vmax_s(ax1, bx1[j]), right = gnms_iou3d::vmin_s(ax2, bx2[j]);   // :25
            const float top = gnms_iou3d::vmax_s(ay1, by1[j]), bottom = gnms_iou3d::vmin_s(ay2, by2[j]);   // :26
            const float width = fmaxf(right - left + shift, 0.f), height = fmaxf(bottom - top + shift, 0.f);   // :27
            inter[j] = width * height;                                            // :28
            uni[j] = as + bs[j] - inter[j];                                       // :31's denominator
            d[j] = __builtin_fmaf(-thresh, uni[j], inter[j]);
            unsure |= !(uni[j] > 0.0f) || !(uni[j] < INFINITY) || !(fabsf(d[j]) > guard * uni[j]);   // (also true for NaN)
        }
        if (__any(unsure)) {                                                      // (wave-uniform; rare)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float ov = inter[j] / uni[j];                               // :31
                const bool sup = keep_le ? !(ov <= thresh) : (ov > thresh);       // :71 / nms_others.py:146
                if (r < 32) lo[j] |= sup ? (1u << r) : 0u; else hi[j] |= sup ? (1u << (r - 32)) : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const bool sup = d[j] > 0.0f;
                if (r < 32) lo[j] |= sup ? (1u << r) : 0u; else hi[j] |= sup ? (1u << (r - 32)) : 0u;
            }
        }
    }
    const unsigned long long rowmask = (nrows >= 64) ? ~0ull : ((1ull << nrows) - 1ull);
    u64* Wk = I.W + (size_t)kb * L.NC;
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (col[j] < L.NC) Wk[col[j]] = ((((u64)hi[j]) << 32) | lo[j]) & rowmask;
}

// The same bit matrix from float64 boxes, every operation in double: lib/nms_others.py:119-150 girshick_nms computes in the dtype of
// `dets`, which is float64 for the arrays its own test feeds (test/test_differentiable_nms_forward.py:111-114) -- in fp32 an overlap
// within a rounding of `thresh` could flip a keep / suppress decision relative to the reference.  fp64 vector rate is ample for a
// test-only helper; the scan behind it is the same.
__device__ __forceinline__ double bcastd(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(b & 0xffffffffll), lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__global__ __launch_bounds__(256) void classic_mask_f64_kernel(const double* __restrict__ boxes, int n, int dim, double thresh, double shift, int keep_le,
                                                               char* ws, gnms_ws_layout L, int init) {
    if (init && blockIdx.y == 0) classic_init_part<256>(n, ws, L);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kb = blockIdx.y;
    const int k0 = kb * 64;
    const int c0 = (blockIdx.x * 4 + wave) * 64;            // one column per lane (64 x 64 tiles): doubles take two registers each
    if (k0 >= n || c0 >= n || c0 >= k0 + 64) return;
    ImgPtrs I = img_ptrs(ws, L, 0);
    const int col = c0 + lane;
    const double* p = boxes + (size_t)(col < n ? col : n - 1) * dim;
    const double bx1 = p[0], by1 = p[1], bx2 = p[2], by2 = p[3];
    const double bs = (bx2 - bx1 + shift) * (by2 - by1 + shift);
    const double* q = boxes + (size_t)min(k0 + lane, n - 1) * dim;
    const double rx1 = q[0], ry1 = q[1], rx2 = q[2], ry2 = q[3];
    const double rs = (rx2 - rx1 + shift) * (ry2 - ry1 + shift);
    const int nrows = min(64, n - k0);
    u64 word = 0ull;
    for (int r = 0; r < nrows; ++r) {
        const double ax1 = bcastd(rx1, r), ay1 = bcastd(ry1, r), ax2 = bcastd(rx2, r), ay2 = bcastd(ry2, r), as = bcastd(rs, r);
        const double w = fmax(0.0, fmin(ax2, bx2) - fmax(ax1, bx1) + shift);     // nms_others.py:139-143
        const double h = fmax(0.0, fmin(ay2, by2) - fmax(ay1, by1) + shift);
        const double inter = w * h;
        const double ov = inter / (as + bs - inter);                              // :144
        const bool sup = keep_le ? !(ov <= thresh) : (ov > thresh);               // :146
        word |= sup ? (1ull << r) : 0ull;
    }
    if (col < L.NC) I.W[(size_t)kb * L.NC + col] = word;
}

__global__ void classic_export_kernel(int n, char* ws, gnms_ws_layout L, int* __restrict__ keep, int* __restrict__ num_out) {
    ImgPtrs I = img_ptrs(ws, L, 0);
    const int nl = I.misc[0];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) *num_out = nl;
    if (t < nl) keep[t] = I.leadr[t];                      // :131 keep_out[num_to_keep++] = i
}

// the same as ONE workgroup that ends with a tag for a polling host (`_nms`: keep / num_out / tag lie in fine-grained pinned memory): every
// thread's stores are out (system scope) before the barrier, the tag leaves behind it
__global__ __launch_bounds__(1024) void classic_export_tag_kernel(char* ws, gnms_ws_layout L, int* __restrict__ keep, int* __restrict__ num_out,
                                                                  int* __restrict__ tag_ptr, int tag) {
    ImgPtrs I = img_ptrs(ws, L, 0);
    const int nl = I.misc[0];
    for (int t = threadIdx.x; t < nl; t += 1024) keep[t] = I.leadr[t];
    if (threadIdx.x == 0) *num_out = nl;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(tag_ptr, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------
// LARGE inputs in CHUNKS (round 6; fp32, n > GNMS_MAX_BOXES: the reference's `use_nms and synced` inference path hands gpu_nms every anchor of
// an image, > 100 k boxes, lib/rpn_util.py:1268).  Greedy NMS never needs the whole n x n matrix -- only (kept box, candidate) pairs:
//   for every chunk of GNMS_MAX_BOXES boxes, in score order:
//     classic_ext_kernel     which boxes of the chunk does a box KEPT SO FAR suppress?  (m x kept pair decisions straight from the boxes: the
//                            kept list is a couple of thousand boxes where the matrix row would be the 126 720 of every anchor)
//     classic_mask_kernel    the chunk's own upper block triangle, as for a small input
//     leaders_kernel         the layer's scan, one workgroup per super-block, with those boxes removed from the start (ext0)
//     classic_append_kernel  the chunk's kept boxes -> keep[] (global indices), their boxes -> the kept list, the count
// The same keep list as the reference's scan (a box is suppressed iff some EARLIER KEPT box overlaps it by more than the threshold), with
// n^2 / (2 chunks) + n kept pair decisions instead of n^2 / 2 and 33 MB of workspace instead of n^2 / 8 bytes: n = 126 720 took 5.0 ms of mask
// kernel + 6.8 ms of one-workgroup block scan (classic_scan_large_kernel below, still the float64 path).
// ------------------------------------------------------------------------------------------------
// (one workgroup per 64 candidates, its sixteen waves each take every sixteenth kept box of a staged tile: 256 workgroups at m = 16 384 where
// one thread per candidate and 256 per workgroup left three quarters of the machine idle -- 248 -> ~30 us per chunk)
__global__ __launch_bounds__(1024) void classic_ext_kernel(const float* __restrict__ boxes, int base, int m, int dim, float thresh, float shift, int keep_le,
                                                           const float4* __restrict__ kbox, const int* __restrict__ nkept_p, u64* __restrict__ extw) {
    __shared__ float4 tile[1024];
    __shared__ unsigned long long s_bits;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x * 64 + lane;                              // chunk-local index of this lane's candidate (the same in all sixteen waves)
    const int nk = *nkept_p;
    const float* q = boxes + (size_t)(base + (k < m ? k : m - 1)) * dim;
    const float bx1 = q[0], by1 = q[1], bx2 = q[2], by2 = q[3];
    const float bs = (bx2 - bx1 + shift) * (by2 - by1 + shift);
    const float guard = fmaxf(fabsf(thresh), 1.0f) * 9.6e-7f;          // (classic_mask_kernel's band: the two kernels take the same decisions)
    if (tid == 0) s_bits = 0ull;
    bool sup = false;
    for (int t0 = 0; t0 < nk; t0 += 1024) {
        __syncthreads();
        if (t0 + tid < nk) tile[tid] = kbox[t0 + tid];
        __syncthreads();
        const int cnt = min(1024, nk - t0);
        for (int j = wave; j < cnt; j += 16) {
            const float4 a = tile[j];                                  // the kept box: the ROW of the matrix entry (nms_kernel.cu:24-32)
            const float as = (a.z - a.x + shift) * (a.w - a.y + shift);
            const float left = fmaxf(a.x, bx1), right = fminf(a.z, bx2), top = fmaxf(a.y, by1), bottom = fminf(a.w, by2);
            const float width = fmaxf(right - left + shift, 0.f), height = fmaxf(bottom - top + shift, 0.f);
            const float inter = width * height;
            const float uni = as + bs - inter;
            const float d = __builtin_fmaf(-thresh, uni, inter);
            bool s1;
            if (!(uni > 0.0f) || !(uni < INFINITY) || !(fabsf(d) > guard * uni)) { const float ov = inter / uni; s1 = keep_le ? !(ov <= thresh) : (ov > thresh); }
            else s1 = d > 0.0f;
            sup |= s1;
        }
    }
    const u64 bits = __ballot(sup && k < m);
    if (lane == 0 && bits) atomicOr(&s_bits, bits);
    __syncthreads();
    if (tid == 0 && blockIdx.x * 64 < m) extw[blockIdx.x] = s_bits;
}

__global__ __launch_bounds__(1024) void classic_append_kernel(char* ws, gnms_ws_layout L, const float* __restrict__ boxes, int base, int dim,
                                                              int* __restrict__ keep, float4* __restrict__ kbox, int* __restrict__ nkept_p,
                                                              int* __restrict__ num_out, int* __restrict__ tag_ptr, int tag, int last) {
    ImgPtrs I = img_ptrs(ws, L, 0);
    const int nl = I.misc[0], at = *nkept_p;
    for (int t = threadIdx.x; t < nl; t += 1024) {
        const int g = base + I.leadr[t];
        keep[at + t] = g;
        const float* q = boxes + (size_t)g * dim;
        kbox[at + t] = make_float4(q[0], q[1], q[2], q[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();                                                   // (every thread has read the old count)
    if (threadIdx.x == 0) {
        *nkept_p = at + nl;
        if (last) {
            *num_out = at + nl;
            if (tag_ptr) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); __hip_atomic_store(tag_ptr, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
}

// LARGE inputs (n > GNMS_MAX_BOXES; the reference's `use_nms and synced` inference path hands gpu_nms every anchor, > 100k boxes,
// lib/rpn_util.py:1268): the leader machinery of the layer keeps per-image state for <= 16384 boxes, so these take the reference's own
// scan (nms_kernel.cu:118-135) on the device instead -- one workgroup, `remv` (one word per 64 boxes) in LDS:
//   for every block of 64 boxes, in order: wave 0 settles the block (the lowest box not yet removed is kept and removes what its word of
//   the diagonal tile says; <= 64 trips on registers), then all threads OR the kept boxes' words into the later blocks' `remv`
//   (thread j owns block j: the kept columns of the 512 bytes W[j][64 b .. 64 b + 63]).
// The mask is read once (its upper block triangle: n^2/16 bytes -- the reference copies all n^2/8 to the HOST and scans there).
constexpr int kClassicLargeMax = 262144;            // remv: 4096 words of LDS; W: 8.6 GB
__global__ __launch_bounds__(1024) void classic_scan_large_kernel(int n, const u64* __restrict__ W, long NC, int* __restrict__ keep,
                                                                  int* __restrict__ num_out, int* __restrict__ tag_ptr, int tag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* remv = reinterpret_cast<u64*>(smem);
    __shared__ u64 s_kept;
    __shared__ int s_num;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = (n + 63) >> 6;
    for (int j = tid; j < nb; j += 1024) remv[j] = 0ull;
    if (tid == 0) s_num = 0;
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        if (wave == 0) {
            const int k0 = b << 6;
            const int nrows = min(64, n - k0);
            u64 cur = remv[b];
            if (nrows < 64) cur |= ~((1ull << nrows) - 1ull);
            const u64 d = (lane < nrows) ? W[(size_t)b * NC + k0 + lane] : 0ull;     // whom box k0 + lane removes inside the block
            u64 kept = 0ull;
            while (~cur != 0ull) {                                                   // wave-uniform: one trip per kept box
                const int p = __builtin_ctzll(~cur);
                kept |= 1ull << p;
                cur |= readlane64(d, p) | (1ull << p);
            }
            const int base = s_num;
            if ((kept >> lane) & 1ull) keep[base + __builtin_popcountll(kept & ((1ull << lane) - 1ull))] = k0 + lane;   // :131
            if (lane == 0) { s_kept = kept; s_num = base + __builtin_popcountll(kept); }
        }
        __syncthreads();
        const u64 kept = s_kept;
        if (kept != 0ull) {
            for (int j = b + 1 + tid; j < nb; j += 1024) {
                const u64* row = W + (size_t)j * NC + ((size_t)b << 6);
                u64 acc = 0ull, m = kept;
                while (m) { acc |= row[__builtin_ctzll(m)]; m &= m - 1; }
                remv[j] |= acc;                                                      // :133-134
            }
        }
        __syncthreads();
    }
    if (tid == 0) *num_out = s_num;
    if (tag_ptr) {                                                                   // (`_nms`: see classic_export_tag_kernel; wave 0 wrote keep[])
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(tag_ptr, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

extern "C" size_t gnms_nms_workspace_bytes(int n) { return n > 0 ? gnms_make_layout(n).per_image : 0; }

namespace {
// what nms_sorted_impl really needs: the image's layout up to GNMS_MAX_BOXES (and for float64 boxes above: the whole bit matrix); fp32 above it
// works in chunks -- one chunk's layout, the kept boxes, the chunk's pre-removed words, a counter (always <= gnms_nms_workspace_bytes(n))
size_t nms_workspace_needed(int n, int is_fp64) {
    if (n <= 0) return 0;
    if (n <= GNMS_MAX_BOXES || is_fp64) return gnms_make_layout(n).per_image;
    return gnms_make_layout(GNMS_MAX_BOXES).per_image + (((size_t)n * 16 + 255) & ~(size_t)255) + (size_t)(GNMS_MAX_BOXES / 64) * 8 + 256;
}
}  // namespace

namespace {
// boxes: float [n][dim] (is_fp64 = 0) or double [n][dim] (is_fp64 = 1)
// tag_ptr: a word of fine-grained pinned memory that receives `tag` behind the last store to keep / num_out (`_nms`), or null
int nms_sorted_impl(const void* boxes, int is_fp64, int n, int boxes_dim, double thresh, double shift, int keep_le, int32_t* keep,
                    int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream, int32_t* tag_ptr = nullptr, int32_t tag = 0) {
    GNMS_CHECK_ARG(n >= 0 && boxes_dim >= 4, "gnms_nms_sorted_shift: bad shape (n=%d dim=%d)", n, boxes_dim);
    GNMS_CHECK_ARG(num_out != nullptr, "gnms_nms_sorted: num_out is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { GNMS_CHECK_HIP(hipMemsetAsync(num_out, 0, sizeof(int32_t), st)); return GNMS_OK; }
    if (n > kClassicLargeMax) { gnms_set_error("gnms_nms_sorted: n=%d exceeds %d", n, kClassicLargeMax); return GNMS_ERR_UNSUPPORTED; }
    GNMS_CHECK_ARG(boxes && keep && workspace, "gnms_nms_sorted: null pointer");
    const gnms_ws_layout L = gnms_make_layout(n);
    if (workspace_bytes < nms_workspace_needed(n, is_fp64)) { gnms_set_error("gnms_nms_sorted: workspace too small"); return GNMS_ERR_WORKSPACE; }
    char* ws = (char*)workspace;
    GNMS_CHECK_ARG(L.NB <= 65535, "gnms_nms_sorted: too many row blocks");
    // (init: the identity order and the counters the leader scan reads -- written by the first row block's workgroups, three launches instead of four)
    // upper: the triangle the super-block scan reads (fp32, n <= GNMS_MAX_BOXES); else the one a suppressor can reach (general scan, block scan)
    const bool upper = !is_fp64 && n <= GNMS_MAX_BOXES;
    auto mask = [&](int init) {
        const dim3 g1(gnms_div_up(((n + 63) / 64) * (((n + 63) / 64) + 1) / 2, 4)), g4(gnms_div_up(n, 1024), L.NB);
        if (is_fp64)
            classic_mask_f64_kernel<<<dim3(gnms_div_up(n, 256), L.NB), 256, 0, st>>>((const double*)boxes, n, boxes_dim, thresh, shift, keep_le, ws, L, init);
        else if ((long)L.NB * gnms_div_up(n, 256) / 2 < 4096) {     // fewer 64 x 256 tiles than four per SIMD: 64 x 64 tiles
            if (upper) classic_mask_kernel<1, true><<<g1, 256, 0, st>>>((const float*)boxes, n, boxes_dim, (float)thresh, (float)shift, keep_le, ws, L, init);
            else classic_mask_kernel<1><<<g1, 256, 0, st>>>((const float*)boxes, n, boxes_dim, (float)thresh, (float)shift, keep_le, ws, L, init);
        } else {
            if (upper) classic_mask_kernel<4, true><<<g4, 256, 0, st>>>((const float*)boxes, n, boxes_dim, (float)thresh, (float)shift, keep_le, ws, L, init);
            else classic_mask_kernel<4><<<g4, 256, 0, st>>>((const float*)boxes, n, boxes_dim, (float)thresh, (float)shift, keep_le, ws, L, init);
        }
    };
    if (n > GNMS_MAX_BOXES && !is_fp64) {                         // in chunks of GNMS_MAX_BOXES (see classic_ext_kernel)
        const int C = GNMS_MAX_BOXES;
        const gnms_ws_layout Lc = gnms_make_layout(C);
        float4* kbox = reinterpret_cast<float4*>(ws + Lc.per_image);                          // [n] the kept boxes, in keep order
        u64* extw = reinterpret_cast<u64*>(ws + Lc.per_image + (((size_t)n * 16 + 255) & ~(size_t)255));   // [C / 64]
        int* nkept = reinterpret_cast<int*>(extw + C / 64);
        GNMS_CHECK_HIP(hipMemsetAsync(nkept, 0, sizeof(int), st));
        const float* bf = (const float*)boxes;
        for (int base = 0; base < n; base += C) {
            const int m = std::min(C, n - base);
            if (base > 0) {
                classic_ext_kernel<<<gnms_div_up(m, 64), 1024, 0, st>>>(bf, base, m, boxes_dim, (float)thresh, (float)shift, keep_le, kbox, nkept, extw);
                GNMS_CHECK_LAUNCH();
            }
            const int nbm = (m + 63) / 64;
            if ((long)nbm * gnms_div_up(m, 256) / 2 < 4096)
                classic_mask_kernel<1, true><<<dim3(gnms_div_up(nbm * (nbm + 1) / 2, 4)), 256, 0, st>>>(bf + (size_t)base * boxes_dim, m, boxes_dim, (float)thresh, (float)shift, keep_le, ws, Lc, 1);
            else
                classic_mask_kernel<4, true><<<dim3(gnms_div_up(m, 1024), nbm), 256, 0, st>>>(bf + (size_t)base * boxes_dim, m, boxes_dim, (float)thresh, (float)shift, keep_le, ws, Lc, 1);
            GNMS_CHECK_LAUNCH();
            const size_t lds = leaders_lds_size(Lc.NB);
            if (lds > 64 * 1024) {
                const int rc = gnms_allow_lds_raw(reinterpret_cast<const void*>(leaders_kernel), lds);
                if (rc) return rc;
            }
            const int spw = leaders_chain_wgs(m, 1);
            leaders_kernel<<<spw, 1024, lds, st>>>(m, nullptr, ws, Lc, 1, 1, spw, base > 0 ? extw : nullptr);
            GNMS_CHECK_LAUNCH();
            classic_append_kernel<<<1, 1024, 0, st>>>(ws, Lc, bf, base, boxes_dim, keep, kbox, nkept, num_out, tag_ptr, tag, base + C >= n ? 1 : 0);
            GNMS_CHECK_LAUNCH();
        }
        return GNMS_OK;
    }
    if (n > GNMS_MAX_BOXES) {                                     // float64: the reference's scan on the device (classic_scan_large_kernel)
        mask(0);
        GNMS_CHECK_LAUNCH();
        classic_scan_large_kernel<<<1, 1024, (size_t)L.NB * 8, st>>>(n, img_ptrs(ws, L, 0).W, (long)L.NC, keep, num_out, tag_ptr, tag);
        GNMS_CHECK_LAUNCH();
        return GNMS_OK;
    }
    mask(1);
    GNMS_CHECK_LAUNCH();
    const size_t lds = leaders_lds_size(L.NB);
    if (lds > 64 * 1024) {
        const int rc = gnms_allow_lds_raw(reinterpret_cast<const void*>(leaders_kernel), lds);       // (remembered per device and kernel)
        if (rc) return rc;
    }
    if (upper) { const int spw = leaders_chain_wgs(n, 1); leaders_kernel<<<spw, 1024, lds, st>>>(n, nullptr, ws, L, 1, 1, spw); }   // one workgroup per super-block
    else leaders_kernel<<<1, 1024, lds, st>>>(n, nullptr, ws, L, 0, 1, 1);
    GNMS_CHECK_LAUNCH();
    if (tag_ptr) classic_export_tag_kernel<<<1, 1024, 0, st>>>(ws, L, keep, num_out, tag_ptr, tag);
    else classic_export_kernel<<<gnms_div_up(n, 256), 256, 0, st>>>(n, ws, L, keep, num_out);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
}  // namespace

extern "C" int gnms_nms_sorted_shift(const float* boxes, int n, int boxes_dim, float thresh, float shift, int keep_le, int32_t* keep,
                                     int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream) {
    return nms_sorted_impl(boxes, 0, n, boxes_dim, (double)thresh, (double)shift, keep_le, keep, num_out, workspace, workspace_bytes, stream);
}

extern "C" int gnms_nms_sorted_shift_f64(const double* boxes, int n, int boxes_dim, double thresh, double shift, int keep_le, int32_t* keep,
                                         int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream) {
    return nms_sorted_impl(boxes, 1, n, boxes_dim, thresh, shift, keep_le, keep, num_out, workspace, workspace_bytes, stream);
}

extern "C" int gnms_nms_sorted(const float* boxes, int n, int boxes_dim, float thresh, int32_t* keep, int32_t* num_out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return gnms_nms_sorted_shift(boxes, n, boxes_dim, thresh, 1.0f, 0, keep, num_out, workspace, workspace_bytes, stream);
}

// The reference's exact symbol (lib/nms/gpu_nms.hpp:1-2): host pointers, blocking (nms_kernel.cu:100-108,142-143 allocate, copy in,
// copy the whole bit matrix out and free per call).  Errors cannot be returned through this signature; unlike the
// reference (which prints and carries on, :12-19) a failure yields *num_out = 0 and the message is kept for gnms_last_error().
//
// Round 5: neither copy is a blocking hipMemcpy of pageable memory any more.  One block of fine-grained (coherent, host-mapped) pinned
// memory per device stays between the calls: the boxes go through it (a host memcpy, then an asynchronous copy to the device -- the bit-matrix
// kernel reads every box n / 64 times, which must not cross PCIe), `keep[]` and the count are written INTO it by the kernels themselves
// (classic_export_tag_kernel / classic_scan_large_kernel only ever store to them) and end with the call's tag, and
// the host polls the tag: no device-to-host copy, no stream synchronisation.  Per call with the mask kernel's small tiles (tools/nms_host.py, old and new
// library on one box, profiles/r05h_nms_host_ab.txt): n = 500 75 -> 40-50 us, n = 2000 100 -> 74-82, n = 4096 148 -> 125-128.
namespace {
struct NmsStage {
    std::mutex mu;                 // a call holds it from its first byte in the block to its last out of it (calls on the null stream never overlapped anyway)
    char* host = nullptr;          // [0, 256): the tag; then keep[n] + the count; then the boxes
    char* dev = nullptr;           // the same block as the device addresses it
    size_t bytes = 0;
    uint32_t seq = 0;
};
NmsStage g_nms_stage[64];

// blocks until the tag has arrived (or the null stream has run empty / failed); false: HIP error.  Spin, yield, sleep (gnms_poll_backoff); after
// two seconds without the tag the stream is synchronised instead (as the counts mailbox does) -- the caller holds the per-device mutex meanwhile
bool nms_wait_tag(const char* host, int32_t tag) {
    const int32_t* h = reinterpret_cast<const int32_t*>(host);
    const auto t0 = std::chrono::steady_clock::now();
    gnms_poll_backoff bo;
    for (;;) {
        if (__atomic_load_n(h, __ATOMIC_ACQUIRE) == tag) return true;
        if (bo.wait()) {
            const hipError_t q = hipStreamQuery(nullptr);
            if (q != hipSuccess && q != hipErrorNotReady) return false;
            if (q == hipSuccess || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
                return __atomic_load_n(h, __ATOMIC_ACQUIRE) == tag || hipStreamSynchronize(nullptr) == hipSuccess;
        }
    }
}
}  // namespace

extern "C" void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                     float nms_overlap_thresh, int device_id) {
    if (num_out) *num_out = 0;
    if (!keep_out || !num_out || boxes_num <= 0 || !boxes_host) return;
    if (boxes_dim < 4) { gnms_set_error("_nms: boxes_dim=%d < 4", boxes_dim); return; }   // the kernel reads 5 fields, stride boxes_dim
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { gnms_set_error("_nms: hipGetDevice failed"); return; }
    if (cur != device_id && hipSetDevice(device_id) != hipSuccess) { gnms_set_error("_nms: hipSetDevice(%d) failed", device_id); return; }   // :80-89
    const size_t bbytes = (size_t)boxes_num * boxes_dim * sizeof(float);
    const size_t wbytes = nms_workspace_needed(boxes_num, 0);
    const size_t off_ws = (bbytes + 255) / 256 * 256;
    // the pinned block: tag | keep[] + count | boxes
    const size_t st_keep = 256, st_boxes = st_keep + ((size_t)(boxes_num + 1) * 4 + 255) / 256 * 256, st_need = st_boxes + bbytes;
    NmsStage* S = (device_id >= 0 && device_id < 64) ? &g_nms_stage[device_id] : nullptr;
    std::unique_lock<std::mutex> lock;
    if (S) {
        lock = std::unique_lock<std::mutex>(S->mu);
        if (S->bytes < st_need) {
            if (S->host) { (void)hipDeviceSynchronize(); (void)hipHostFree(S->host); S->host = S->dev = nullptr; S->bytes = 0; }
            // (the block starts at 64 KiB -- a call at the reference's sizes, <= 3000 boxes -- and grows to what the largest call needed: a
            // device synchronisation and a re-allocation each time, a handful of times per process; test_classic_nms_pinned_staging walks it up)
            constexpr size_t min_bytes = (size_t)64 << 10;
            const size_t want = std::max<size_t>((st_need + (1u << 16) - 1) >> 16 << 16, min_bytes);
            void *h = nullptr, *d = nullptr;
            if (hipHostMalloc(&h, want, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                memset(h, 0, 256);
                S->host = (char*)h; S->dev = (char*)d; S->bytes = want;
            } else {
                if (h) (void)hipHostFree(h);
                (void)hipGetLastError();
                S = nullptr;                                       // no pinned block: the plain copies below
                lock.unlock();
            }
        }
    }
    char* dev = nullptr;
    // stream-ordered allocation on the null stream: the pool keeps the block between calls (the reference pays a cudaMalloc /
    // cudaFree per call, nms_kernel.cu:100-108,142-143; here that was two thirds of the call)
    const size_t off_keep_d = off_ws + wbytes;                    // (only the plain path keeps keep[] on the device)
    if (hipMallocAsync((void**)&dev, off_keep_d + (S ? 0 : (size_t)(boxes_num + 1) * 4), nullptr) != hipSuccess) { gnms_set_error("_nms: hipMallocAsync failed"); return; }
    int rc = GNMS_OK;
    hipError_t e;
    if (S) {
        memcpy(S->host + st_boxes, boxes_host, bbytes);
        e = hipMemcpyAsync(dev, S->host + st_boxes, bbytes, hipMemcpyHostToDevice, nullptr);                 // :103-106 (pinned source: asynchronous)
        int32_t* keep_p = reinterpret_cast<int32_t*>(S->dev + st_keep);
        const int32_t tag = (int32_t)(++S->seq | 0x40000000u);
        if (e == hipSuccess)
            rc = nms_sorted_impl(dev, 0, boxes_num, boxes_dim, (double)nms_overlap_thresh, 1.0, 0, keep_p, keep_p + boxes_num, dev + off_ws, wbytes, nullptr,
                                 reinterpret_cast<int32_t*>(S->dev), tag);
        if (e == hipSuccess && rc == GNMS_OK && !nms_wait_tag(S->host, tag)) e = hipErrorUnknown;
        if (e == hipSuccess && rc == GNMS_OK) {
            const int32_t* hk = reinterpret_cast<const int32_t*>(S->host + st_keep);
            const int num = hk[boxes_num];
            if (num > 0) memcpy(keep_out, hk, (size_t)num * sizeof(int));
            *num_out = num;
        }
    } else {
        int32_t* keep_d = (int32_t*)(dev + off_keep_d);
        int32_t* num_d = keep_d + boxes_num;                      // staged right behind keep[] on the device
        e = hipMemcpy(dev, boxes_host, bbytes, hipMemcpyHostToDevice);
        if (e == hipSuccess)
            rc = gnms_nms_sorted((const float*)dev, boxes_num, boxes_dim, nms_overlap_thresh, keep_d, num_d, dev + off_ws, wbytes, nullptr);
        // keep[] and the count come back in ONE blocking copy (it orders after the kernels): boxes_num + 1 ints
        std::vector<int32_t> host((size_t)boxes_num + 1);
        if (e == hipSuccess && rc == GNMS_OK) e = hipMemcpy(host.data(), keep_d, host.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess && rc == GNMS_OK) {
            const int num = host[boxes_num];
            if (num > 0) memcpy(keep_out, host.data(), (size_t)num * sizeof(int));
            *num_out = num;
        }
    }
    (void)hipFreeAsync(dev, nullptr);
    if (e != hipSuccess) gnms_set_error("_nms: HIP copy failed");
}
