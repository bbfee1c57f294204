// nms_kernels.h -- GrooMeD-NMS forward kernels for gfx950 (MI355X): sort, threshold bit-matrix, greedy
// leader scan, group attribution, grouping + default (masked) rescoring, validity split.
//
// Reference: lib/groomed_nms.py:10-129 differentiable_nms and :208-270 get_groups.
//
// Formulation (see DESIGN.md for the derivation).  get_groups repeatedly takes the highest-scoring
// remaining box as leader L and removes every remaining box i with NOT(iou[i][L] <= thr) (:249-262);
// the ones with iou[i][L] > thr, capped at group_size+1 in score order, form the group (:253-255).
// That is classical greedy NMS plus "each box remembers the first leader that removed it".  We
// therefore never materialise the score-sorted copy of the matrix (:48) nor the N x N inversion
// matrix (:65,:108):
//   K1 sort_scores   stable descending argsort per image                       (one workgroup / image, LDS bitonic)
//   K2 bitmask       the ONE full read of the N x N fp32 matrix -> N*N/8-byte bit matrix W
//                    (HBM-bound: 4 N^2 bytes in, N^2/8 out)                      <- dominant kernel
//   K3 leaders       sequential scan over 64-rank blocks on the bit matrix       (one workgroup / image)
//   K4 attribute     first-remover per box, parallel over rank blocks
//   K5 groups        membership (strict >), cap, head, CSR of groups; default rescoring fused
//   K6 finalize      clamp / threshold / second sort / valid + invalid lists / output order
#pragma once
#include <type_traits>
#include "gnms_common.h"
#include "iou3d_pair.h"
#include "iou_tile.h"

namespace gnms {
namespace {   // internal linkage: the header is included by several translation units

typedef unsigned long long u64;

// Data that one workgroup of a launch hands to another workgroup of the SAME launch (the leader scan's granules and rem[], everything in
// one_launch_kernel): the L2s of the eight XCDs are not coherent with each other inside a kernel, so such data leaves through agent-scope
// (write-through) stores and is read with agent-scope loads.
template <typename T> __device__ __forceinline__ T coh_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void coh_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float4 coh_load_f4(const float4* p) {
    const u64 a = coh_load(reinterpret_cast<const u64*>(p)), c = coh_load(reinterpret_cast<const u64*>(p) + 1);
    return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)c), __uint_as_float((unsigned)(c >> 32)));
}
__device__ __forceinline__ void coh_store_f4(float4* p, const float4 v) {
    coh_store(reinterpret_cast<u64*>(p), ((u64)__float_as_uint(v.y) << 32) | __float_as_uint(v.x));
    coh_store(reinterpret_cast<u64*>(p) + 1, ((u64)__float_as_uint(v.w) << 32) | __float_as_uint(v.z));
}

// ------------------------------------------------------------------------------------------------
// Workgroup sort of P = blockDim.x * E 64-bit keys (P a power of two, blockDim.x a multiple of 64),
// ascending.  Thread t owns elements t*E .. t*E+E-1 in registers.  A bitonic network whose stages run
//   j <  E      in registers (compile-time partner),
//   j <  64*E   as wave shuffles (partner lane = lane ^ j/E),
//   j >= 64*E   through LDS (transposed [e][t] layout: conflict-free), two barriers each.
// For P = 4096 that is 10 LDS stages instead of the 78 barrier-separated stages of a plain LDS bitonic.
// On return the sorted keys are in r[] AND in keys[0..P) (natural order), barrier included.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 shfl_xor_key(u64 v, int m) {
    const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), m, 64);
    const unsigned hi = __shfl_xor((unsigned)(v >> 32), m, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ unsigned shfl_xor_key(unsigned v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ u64 shfl_up_u64(u64 v, int off) {
    const unsigned lo = __shfl_up((unsigned)(v & 0xffffffffu), off, 64);
    const unsigned hi = __shfl_up((unsigned)(v >> 32), off, 64);
    return ((u64)hi << 32) | lo;
}

#ifdef GNMS_TIMING
__device__ long long g_sort_ticks[4];
#define GNMS_ST0() long long st__ = (long long)__builtin_amdgcn_s_memtime()
#define GNMS_STACC(slot) do { long long n__ = (long long)__builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) g_sort_ticks[slot] += n__ - st__; st__ = n__; } while (0)
#else
#define GNMS_ST0() do {} while (0)
#define GNMS_STACC(slot) do {} while (0)
#endif

template <int E, typename K>
__device__ __forceinline__ void block_sort_bitonic(K (&r)[E], K* keys, int P) {
    const int t = threadIdx.x;
    const int T = blockDim.x;
    GNMS_ST0();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= 64 * E; j >>= 1) {                    // cross-wave stages
#pragma unroll
            for (int e = 0; e < E; ++e) keys[e * T + t] = r[e];
            __syncthreads();
            const int pt = t ^ (j / E);
            const bool keepmin = (((t * E) & k) == 0) == (((t * E) & j) == 0);   // j >= E: the same for all E elements
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const K o = keys[e * T + pt];
                r[e] = ((o < r[e]) == keepmin) ? o : r[e];              // one compare per exchange
            }
            __syncthreads();
        }
        GNMS_STACC(0);
        {
            int j = (k >> 1) < 64 * E ? (k >> 1) : 32 * E;              // intra-wave stages
            for (; j >= E; j >>= 1) {
                const int m = j / E;
                const bool keepmin = (((t * E) & k) == 0) == (((t * E) & j) == 0);   // j >= E: the same for all E elements
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const K o = shfl_xor_key(r[e], m);
                    r[e] = ((o < r[e]) == keepmin) ? o : r[e];
                }
            }
        }
        GNMS_STACC(1);
#pragma unroll
        for (int jj = E / 2; jj >= 1; jj >>= 1) {                        // in-register stages
            if (jj < k) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & jj) == 0) {
                        const int i = t * E + e;
                        const bool up = (i & k) == 0;
                        const K a = r[e], b = r[e | jj];
                        if ((a > b) == up) { r[e] = b; r[e | jj] = a; }
                    }
                }
            }
        }
        GNMS_STACC(2);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) keys[t * E + e] = r[e];
    __syncthreads();
    GNMS_STACC(3);
}

// ------------------------------------------------------------------------------------------------
// block_sort: the production sort.  Same contract as block_sort_bitonic (thread t owns t*E..t*E+E-1, result in
// r[] and keys[0..P)), different cross-wave phase:
//   1. every wave sorts its own 64*E keys ascending with the register/shuffle part of the bitonic network;
//   2. the 64E-key runs are merged pairwise, log2(#waves) passes: each thread finds its E outputs by a merge-path
//      binary search on the two runs in LDS and merges them sequentially (ties take the left run).
// That replaces the 10 two-barrier LDS exchange stages + 24 shuffle stages above 64E of the pure bitonic network by
// 4 passes for 4096 keys (measured: 25 -> ~14 us for u64 keys on one CU).
// ------------------------------------------------------------------------------------------------
template <int E, typename K>
__device__ __forceinline__ void block_sort(K (&r)[E], K* keys, int P) {
    const int t = threadIdx.x;
    const int run = 64 * E < P ? 64 * E : P;                            // keys per wave-sorted run
    // ---- phase 1: wave-local bitonic sort (all runs ascending: the direction bit is dropped at k == run) ----
    for (int k = 2; k <= run; k <<= 1) {
        const int kd = (k == run) ? 0 : k;
        for (int j = k >> 1; j >= E; j >>= 1) {
            const int m = j / E;
            const bool keepmin = (((t * E) & kd) == 0) == (((t * E) & j) == 0);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const K o = shfl_xor_key(r[e], m);
                r[e] = ((o < r[e]) == keepmin) ? o : r[e];
            }
        }
#pragma unroll
        for (int jj = E / 2; jj >= 1; jj >>= 1) {
            if (jj < k) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & jj) == 0) {
                        const int i = t * E + e;
                        const bool up = (i & kd) == 0;
                        const K a = r[e], b = r[e | jj];
                        if ((a > b) == up) { r[e] = b; r[e | jj] = a; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) keys[t * E + e] = r[e];
    __syncthreads();
    // ---- phase 2: pairwise merge passes ----
    for (int p = run; p < P; p <<= 1) {
        const int d0 = t * E;
        if (d0 >= P) { __syncthreads(); __syncthreads(); continue; }      // P < blockDim.x * E: the surplus threads only keep the barriers
        const int base = d0 & ~(2 * p - 1);
        const int d = d0 - base;                                        // diagonal inside the pair [A | B], |A| = |B| = p
        const K* A = keys + base;
        const K* Bk = keys + base + p;
        int lo = d > p ? d - p : 0, hi = d < p ? d : p;
        while (lo < hi) {                                               // merge path: first a with A[a] > B[d-1-a]
            const int mid = (lo + hi) >> 1;
            if (A[mid] <= Bk[d - 1 - mid]) lo = mid + 1; else hi = mid;
        }
        int a = lo, b = d - lo;
        K ka = a < p ? A[a] : K(0), kb = b < p ? Bk[b] : K(0);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool takeA = (b >= p) || (a < p && ka <= kb);
            r[e] = takeA ? ka : kb;
            if (takeA) { ++a; ka = a < p ? A[a] : K(0); } else { ++b; kb = b < p ? Bk[b] : K(0); }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) keys[t * E + e] = r[e];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Stable LSD radix pass over P = blockDim.x * E 32-bit keys in LDS, 7 bits per pass (128 digits), in place.
// Element order is the LDS order; wave w owns the 64*E consecutive slots [w*64E, (w+1)*64E), lane l reads slots e*64 + l
// (conflict-free), so a round e covers 64 consecutive elements and the stable rank inside the wave is
//   (same-digit elements of earlier rounds, kept in a per-wave histogram) + popcount(same-digit lanes below me),
// the same-digit lane mask coming from 7 ballots.  A (digit-major, wave-minor) exclusive scan of the 128 x #waves histogram
// gives every (wave, digit) its base.  5 barriers per pass; 2 passes sort 14-bit keys (the groups' leader ranks), where the
// merge-path block_sort needs ~15 us for 4096 keys and this needs ~4.
// hist: 128 * (blockDim.x / 64) + 16 unsigned words of LDS.
// ------------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void block_radix_pass7(unsigned* keys, int shift, unsigned* hist) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
    unsigned* wtot = hist + 128 * nw;
    for (int i = t; i < 128 * nw; i += blockDim.x) hist[i] = 0u;
    unsigned k[E], rnk[E];
#pragma unroll
    for (int e = 0; e < E; ++e) k[e] = keys[wave * 64 * E + e * 64 + lane];
    __syncthreads();
    const u64 below = (1ull << lane) - 1ull;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned d = (k[e] >> shift) & 127u;
        u64 same = ~0ull;
#pragma unroll
        for (int bit = 0; bit < 7; ++bit) {
            const bool on = (d >> bit) & 1u;
            const u64 bl = __ballot(on);
            same &= on ? bl : ~bl;
        }
        const unsigned prior = hist[wave * 128 + d];
        rnk[e] = prior + (unsigned)__builtin_popcountll(same & below);
        if ((same & below) == 0ull) hist[wave * 128 + d] = prior + (unsigned)__builtin_popcountll(same);   // the digit's lowest lane
        __builtin_amdgcn_wave_barrier();                                // the next round reads what this one wrote (same wave: in order)
    }
    __syncthreads();
    // exclusive scan in (digit, wave) order: entry j = d * nw + w lives at hist[w * 128 + d]; thread t owns j = 2t, 2t + 1
    const int j0 = 2 * t, j1 = 2 * t + 1;
    const int lw = 31 - __builtin_clz(nw);                               // #waves is a power of two
    const int a0 = (j0 & (nw - 1)) * 128 + (j0 >> lw), a1 = (j1 & (nw - 1)) * 128 + (j1 >> lw);
    const unsigned c0 = hist[a0], c1 = hist[a1];
    const unsigned inc = gnms_add_scan32(c0 + c1);                       // DPP prefix sum: no ds_bpermute round trips
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    // sum of the totals of the waves before mine (<= 16 of them): lane w < wave holds total w, lane 63 of the scan has the sum
    const unsigned carry = (unsigned)__builtin_amdgcn_readlane((int)gnms_add_scan32((lane < wave) ? wtot[lane] : 0u), 63);
    const unsigned ex = carry + inc - (c0 + c1);
    hist[a0] = ex;
    hist[a1] = ex + c0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) keys[hist[wave * 128 + ((k[e] >> shift) & 127u)] + rnk[e]] = k[e];
    __syncthreads();
}

template <typename K>
__device__ __forceinline__ int lower_bound_lds(const K* keys, int n, K v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct ImgPtrs {
    int* order; float* sscore; int* rankof; int* rem; int* head; int* gpos; int* gsorted; int* gstart; int* glen; int* hlist;
    float* plead; float* pre; float* r2; int* sidx; float* xsol; float* gx; int* leadc; int* leadr; u64* leadw; int* leadpfx;
    int* misc; u64* gran; int* xidx; float4* xbox; float4* rbox; float* rec; float* xrec; u64* W;
};

__device__ __host__ __forceinline__ ImgPtrs img_ptrs(char* ws, const gnms_ws_layout& L, int b) {
    char* p = ws + (size_t)b * L.per_image;
    ImgPtrs I;
    I.order = (int*)(p + L.off_order); I.sscore = (float*)(p + L.off_sscore); I.rankof = (int*)(p + L.off_rankof); I.rem = (int*)(p + L.off_rem);
    I.head = (int*)(p + L.off_head); I.gpos = (int*)(p + L.off_gpos); I.gsorted = (int*)(p + L.off_gsorted);
    I.gstart = (int*)(p + L.off_gstart); I.glen = (int*)(p + L.off_glen); I.hlist = (int*)(p + L.off_hlist); I.plead = (float*)(p + L.off_plead);
    I.pre = (float*)(p + L.off_pre); I.r2 = (float*)(p + L.off_r2); I.sidx = (int*)(p + L.off_sidx);
    I.xsol = (float*)(p + L.off_xsol); I.gx = (float*)(p + L.off_gx); I.leadc = (int*)(p + L.off_leadc); I.leadr = (int*)(p + L.off_leadr);
    I.leadw = (u64*)(p + L.off_leadw); I.leadpfx = (int*)(p + L.off_leadpfx); I.misc = (int*)(p + L.off_misc);
    I.gran = (u64*)(p + L.off_gran); I.xidx = (int*)(p + L.off_xidx); I.xbox = (float4*)(p + L.off_xbox); I.rbox = (float4*)(p + L.off_rbox); I.rec = (float*)(p + L.off_rec); I.xrec = (float*)(p + L.off_xrec);
    I.W = (u64*)(p + L.off_W);
    return I;
}

// IoU of box a (row) against box b (column): the arithmetic of iou2d_kernel / lib/core.py:210-218,499-508, operation
// for operation, so that the from-boxes path takes exactly the decisions the matrix path takes.
__device__ __forceinline__ float pair_iou(const float4 a, const float4 b) {
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const float area_b = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
    const float inter = w * h;
    return inter / ((area_a + area_b) - inter);
}

// Where the kernels that need single overlap entries take them from.  The template parameter is an int so that the existing
// <false> / <true> instantiations keep meaning "matrix" / "2D boxes".
constexpr int kFromMatrix = 0;     // src = the N x N matrix of the image, leading dimension ld
constexpr int kFromBoxes = 1;      // src = the image's boxes [N][4]: lib/core.py iou recomputed (pair_iou, the arithmetic of iou2d_tile)
constexpr int kFromRecords = 2;    // src = the image's corner-AABB records [N][12]: 0.5 * (1 + GIoU3D) recomputed (iou3d_pair.h)

// element [ca][cb] (input indices)
// thr: the layer's nms_threshold (records only: entries inside the guard band around it take the reference's exact order, iou3d_pair.h)
template <int SRC>
__device__ __forceinline__ float overlap_at(const float* __restrict__ src, long ld, int ca, int cb, float thr) {
    if (SRC == kFromBoxes) {
        const float4* bx = reinterpret_cast<const float4*>(src);
        return pair_iou(bx[ca], bx[cb]);
    }
    if (SRC == kFromRecords) return gnms_iou3d::nms_overlap3d_pair(src + (size_t)ca * gnms_iou3d::kRec, src + (size_t)cb * gnms_iou3d::kRec, thr);
    return src[(size_t)ca * ld + cb];
}

// the image's slice of what the caller passed as `src` (records live in the workspace, not in a caller array)
template <int SRC>
__device__ __forceinline__ const float* overlap_src(const float* __restrict__ src, const ImgPtrs& I, int b, int N, long ld) {
    if (SRC == kFromRecords) return I.rec;
    return src + (SRC == kFromBoxes ? (size_t)b * N * 4 : (size_t)b * N * ld);
}

// ------------------------------------------------------------------------------------------------
// K1: stable descending argsort of the scores (lib/groomed_nms.py:41; get_groups :213)
// ------------------------------------------------------------------------------------------------
// Second, independent sort of the from-boxes path: the boxes by ascending x centre (NaN last).  bitmask_boxes_kernel walks
// its COLUMNS in this order, so that the 256 columns of a wave tile are spatial neighbours and the rows that cannot touch
// their hull are skipped.  Thread t owns elements t*E .. t*E+E-1; `keys` = P 64-bit LDS slots.
// A box the packed intersection of bitmask_boxes_body may see: finite coordinates, no negative zero, x2 >= x1, y2 >= y1
__device__ __forceinline__ bool box_orders_plainly(const float4 v) {
    auto fine = [](float c) { const unsigned u = __float_as_uint(c); return u != 0x80000000u && (u & 0x7fffffffu) < 0x7f800000u; };
    return fine(v.x) && fine(v.y) && fine(v.z) && fine(v.w) && v.z >= v.x && v.w >= v.y;
}

// Column order of the 3D bit-matrix kernel (bitmask_rec3d_slots_kernel): the cuboids arrive as pseudo boxes (x0, lx, x1, z0 + z1) and
// are ordered by (z band, x centre) -- `mode3d` equal-width bands over the image's range of z0 + z1 -- so that 64 consecutive columns
// (one SLOT of a wave tile) are a compact patch in x AND z and the rows far from it in either direction are skipped.  mode3d = 0: plain
// 2D boxes by x centre; 1: cuboids, one band.  Key = band << 60 | x key << 28 | input index (N <= 16384): distinct, NaN x last.
constexpr unsigned kColIdxMask = 0x0fffffffu;
__device__ __forceinline__ u64 column_key(const float4 v, int i, int nbands, float zlo, float zscale) {
    unsigned band = 0u;
    if (nbands > 1) {
        const float f = (v.w - zlo) * zscale;
        band = (f == f) ? (unsigned)fminf(fmaxf(f, 0.0f), (float)(nbands - 1)) : (unsigned)(nbands - 1);
    }
    return ((u64)band << 60) | ((u64)(~gnms_desc_key(v.x + v.z)) << 28) | (unsigned)i;
}
// range of z0 + z1 over the image's finite pseudo boxes -> (zlo, nbands / (zhi - zlo)); every thread of the workgroup returns the same pair
// (min / max do not depend on the order), so every workgroup of an image bands alike
__device__ __forceinline__ void block_z_bands(const float4* __restrict__ bx, int n, int nbands, float* zlo_out, float* zscale_out) {
    __shared__ float zred[2][16];
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float z = bx[i].w;
        if (fabsf(z) < INFINITY) { lo = fminf(lo, z); hi = fmaxf(hi, z); }
    }
    lo = gnms_wave_min_f(lo); hi = gnms_wave_max_f(hi);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { zred[0][wave] = lo; zred[1][wave] = hi; }
    __syncthreads();
    lo = INFINITY; hi = -INFINITY;
    for (int w = 0; w < nw; ++w) { lo = fminf(lo, zred[0][w]); hi = fmaxf(hi, zred[1][w]); }
    const float span = hi - lo;
    *zlo_out = lo;
    *zscale_out = (span > 0.0f && span < INFINITY) ? (float)nbands / span : 0.0f;
    __syncthreads();
}
// what the column sort leaves behind for rank `k` of its order: the input index, the (pseudo) box, and for cuboids the record
__device__ __forceinline__ void column_store(const ImgPtrs& I, int k, int idx, const float4 v, int mode3d) {
    I.xidx[k] = idx;
    I.xbox[k] = v;
    if (mode3d) {
        const float4* s = reinterpret_cast<const float4*>(I.rec) + (size_t)idx * 3;
        float4* d = reinterpret_cast<float4*>(I.xrec) + (size_t)k * 3;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    }
}

template <int E>
__device__ __forceinline__ void sort_boxes_by_x(const float* __restrict__ boxes, int n, const ImgPtrs& I, u64* keys, int P, int mode3d = 0) {
    const float4* bx = reinterpret_cast<const float4*>(boxes);
    float zlo = 0.0f, zscale = 0.0f;
    if (mode3d > 1) block_z_bands(bx, n, mode3d, &zlo, &zscale);
    u64 r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = threadIdx.x * E + e;
        r[e] = ~0ull;
        if (i < n) r[e] = column_key(bx[i], i, mode3d, zlo, zscale);
    }
    block_sort<E, u64>(r, keys, P);
    bool plain = true;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const int idx = (int)((unsigned)keys[k] & kColIdxMask);
        const float4 v = bx[idx];
        column_store(I, k, idx, v, mode3d);
        plain = plain && box_orders_plainly(v);
    }
    const int all_plain = __syncthreads_and(plain);
    if (threadIdx.x == 0) I.misc[6] = all_plain ? 0 : 1;              // bitmask_boxes_body: the packed intersection needs plain boxes
}

template <int E>
__global__ __launch_bounds__(1024) void sort_scores_kernel(const float* __restrict__ scores, int N, const int* __restrict__ counts,
                                                           char* ws, gnms_ws_layout L, int P, long long* __restrict__ order_out,
                                                           const float* __restrict__ boxes, int mode3d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    const int b = blockIdx.x;
    const int n = gnms_count(counts, b, N);
    const float* s = scores + (size_t)b * N;
    ImgPtrs I = img_ptrs(ws, L, b);
    if (blockIdx.y == 1) {                      // from-boxes path only: grid (B, 2)
        sort_boxes_by_x<E>(boxes + (size_t)b * N * 4, n, I, keys, P, mode3d);
        return;
    }
    u64 r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = threadIdx.x * E + e;
        r[e] = (i < n) ? (((u64)gnms_desc_key(s[i]) << 32) | (unsigned)i) : ~0ull;
    }
    block_sort<E, u64>(r, keys, P);
    int same = 1;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        int idx = k;                  // padding ranks map to themselves
        float v = 0.0f;
        if (k < n) { idx = (int)(keys[k] & 0xffffffffu); v = s[idx]; }
        same &= (idx == k);
        I.order[k] = idx;
        I.rankof[idx] = k;            // order is a permutation of [0,N) (identity on the padding)
        I.sscore[k] = v;
        if (boxes && k < n) I.rbox[k] = reinterpret_cast<const float4*>(boxes)[(size_t)b * N + idx];   // (the from-boxes layer: row boxes of the bit matrix)
        if (order_out) order_out[(size_t)b * N + k] = idx;
    }
    // misc[2] = 1 when the scores came in already sorted (both reference call sites do that: lib/loss/rpn_3d.py:731-737,
    // lib/rpn_util.py:1258-1266): rank == input index, so the bit-matrix kernel can skip the half of the matrix that no
    // leader can reach and needs no scatter.
    const int all_same = __syncthreads_and(same);
    if (threadIdx.x < 8 && !(boxes && threadIdx.x == 6)) I.misc[threadIdx.x] = (threadIdx.x == 2) ? all_same : 0;   // ([6]: the x sort's)
    if (threadIdx.x == 8) I.misc[8] = gnms_next_epoch(I.misc[8]);                   // the workspace's call counter (leaders_sb_body's hand-off tag)
    for (int i = threadIdx.x; i < 17 * 32; i += blockDim.x) I.gran[i] = 0ull;   // (and no granule of this workspace carries a tag yet)
}

// ------------------------------------------------------------------------------------------------
// K1 for N > 1024: the same two sorts (scores descending; boxes by x centre) spread over R = P/1024 workgroups per image.
//   sort_runs_kernel   every workgroup sorts one run of 1024 keys in LDS (block_sort<1>) and parks it in global scratch
//                      (the bit-matrix region W: nothing lives there before K2);
//   sort_merge_kernel  every workgroup loads ALL runs of its image into LDS and ranks its own 1024 keys: final position =
//                      own position + sum over the other runs of (number of keys below mine), R-1 branch-free binary searches
//                      that advance in lock step (their LDS reads overlap).  Keys are distinct (index in the low word).
// One workgroup per image needs 23 us at N=4096 and 103 us at N=16384 on its single CU; this needs about 10 / 20 us.
// role (blockIdx.z): 0 = scores, 1 = boxes by x centre (from-boxes path).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 sort_key_of(int role, const float* __restrict__ scores_img, const float* __restrict__ boxes_img, int i,
                                           int nbands, float zlo, float zscale) {
    if (role == 0) return ((u64)gnms_desc_key(scores_img[i]) << 32) | (unsigned)i;
    return column_key(reinterpret_cast<const float4*>(boxes_img)[i], i, nbands, zlo, zscale);
}

// (run r of image b, role) -- also called from the launch that carries a slice of the matrix write (nms_layer.hip)
__device__ __forceinline__ void sort_runs_body(const float* __restrict__ scores, const float* __restrict__ boxes, int N,
                                               const int* __restrict__ counts, char* ws, gnms_ws_layout L, int P, const int r, const int b,
                                               const int role, const int mode3d = 0) {
    __shared__ u64 keys[1024];
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int i = r * 1024 + (int)threadIdx.x;
    float zlo = 0.0f, zscale = 0.0f;
    if (role == 1 && mode3d > 1) block_z_bands(reinterpret_cast<const float4*>(boxes + (size_t)b * N * 4), n, mode3d, &zlo, &zscale);
    u64 k[1];
    k[0] = (i < n) ? sort_key_of(role, scores + (size_t)b * N, boxes ? boxes + (size_t)b * N * 4 : nullptr, i, mode3d, zlo, zscale) : ~0ull;
    block_sort<1, u64>(k, keys, 1024);
    I.W[(size_t)role * P + i] = k[0];
    if (r == 0 && role == 0 && threadIdx.x < 8) I.misc[threadIdx.x] = (threadIdx.x == 2) ? 1 : 0;   // [2] = "already sorted", cleared below
    if (r == 0 && role == 0 && threadIdx.x == 8) I.misc[8] = gnms_next_epoch(I.misc[8]);   // the workspace's call counter (leaders_sb_body's hand-off tag)
    if (r == 0 && role == 0 && threadIdx.x < 17 * 32) I.gran[threadIdx.x] = 0ull;   // (and no granule of this workspace carries a tag yet)
}

__global__ __launch_bounds__(1024) void sort_runs_kernel(const float* __restrict__ scores, const float* __restrict__ boxes, int N,
                                                         const int* __restrict__ counts, char* ws, gnms_ws_layout L, int P, int mode3d) {
    sort_runs_body(scores, boxes, N, counts, ws, L, P, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, mode3d);
}

template <int R>
__device__ __forceinline__ void sort_merge_body(const float* __restrict__ scores, const float* __restrict__ boxes, int N,
                                                const int* __restrict__ counts, char* ws, gnms_ws_layout L, long long* __restrict__ order_out,
                                                const int r, const int b, const int role, const int mode3d = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* all = reinterpret_cast<u64*>(smem);                          // [R][1024]
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int t = threadIdx.x;
    const u64* runs = I.W + (size_t)role * (R * 1024);
#pragma unroll
    for (int q = 0; q < R; ++q)
        all[q * 1024 + t] = runs[q * 1024 + t];
    __syncthreads();
    const u64 mine = all[r * 1024 + t];
    int same = 1;
    if (mine != ~0ull) {
        // what leaves with the key -- its score, its box -- is fetched by INPUT index, known before the searches: requested here, the round
        // trip runs under them (round 6: behind the rank it was one more exposed memory latency at the end of every workgroup)
        const int idx = (int)((unsigned)mine & (role == 0 ? 0xffffffffu : kColIdxMask));
        float sv = 0.0f;
        float4 bv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (role == 0) sv = scores[(size_t)b * N + idx];
        if (boxes) bv = reinterpret_cast<const float4*>(boxes)[(size_t)b * N + idx];
        int pos[R];
#pragma unroll
        for (int q = 0; q < R; ++q) pos[q] = 0;
#pragma unroll
        for (int h = 512; h >= 1; h >>= 1) {
#pragma unroll
            for (int q = 0; q < R; ++q) pos[q] += (all[q * 1024 + pos[q] + h - 1] < mine) ? h : 0;
        }
        int rank = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) rank += (q == r) ? t : pos[q] + ((all[q * 1024 + pos[q]] < mine) ? 1 : 0);
        if (role == 0) {
            same = (idx == rank);
            I.order[rank] = idx;
            I.rankof[idx] = rank;
            I.sscore[rank] = sv;
            if (boxes) I.rbox[rank] = bv;                             // (the from-boxes layer: row boxes of the bit matrix)
            if (order_out) order_out[(size_t)b * N + rank] = idx;
        } else {
            column_store(I, rank, idx, bv, mode3d);
            same = box_orders_plainly(bv);                            // (role 1: "every box of this run is plain")
        }
    }
    if (role == 1 && !__syncthreads_and(same) && t == 0) I.misc[6] = 1;   // (zeroed by sort_runs_body, a launch earlier; every writer stores 1)
    if (role == 0) {
        // padding ranks map to themselves (order is a permutation of [0, N))
        for (int k = n + r * 1024 + t; k < N; k += R * 1024) {
            I.order[k] = k; I.rankof[k] = k; I.sscore[k] = 0.0f;
            if (order_out) order_out[(size_t)b * N + k] = k;
        }
        if (!__syncthreads_and(same) && t == 0) I.misc[2] = 0;        // (every writer stores 0)
    }
}

template <int R>
__global__ __launch_bounds__(1024) void sort_merge_kernel(const float* __restrict__ scores, const float* __restrict__ boxes, int N,
                                                          const int* __restrict__ counts, char* ws, gnms_ws_layout L,
                                                          long long* __restrict__ order_out, int mode3d) {
    sort_merge_body<R>(scores, boxes, N, counts, ws, L, order_out, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, mode3d);
}


// ------------------------------------------------------------------------------------------------
// K1 for N <= 2048 (round 5): the same two sorts by COUNTING, spread over the machine.  Up to 1024 keys one workgroup per (image, role)
// sorted them in LDS: 4-5 us of merge passes on one CU inside a 9-12-us launch that left 240 CUs idle -- the longest launch in front of the
// chain at the reference's own sizes.  A key's position in the sorted order is the number of keys below it (keys are distinct: the index is
// in the low bits), and that needs no sort at all: workgroup (block, image, role) takes 64 keys, every workgroup holds ALL keys of its
// image in LDS, thread (key, segment) counts the keys of one sixteenth of the image below its own, an LDS atomic adds the sixteen counts,
// and the 64 results leave exactly as the merge kernel's do.  N / 64 workgroups per image and role: 256 at B = 8, N = 1024.
// ------------------------------------------------------------------------------------------------
// FUSED (round 6, one_launch_kernel): the same workgroup as a ROLE of the one launch of a small image.  Nothing of the workspace is
// zeroed or counted here (the launch's tag comes from its caller and the flags are strong granules, see there); everything another
// workgroup of the launch reads leaves through agent-scope stores, and the workgroup's last act is its "sorted" flag.
__device__ __forceinline__ u64 strong_gran(unsigned tag, unsigned slot) { return ((u64)tag << 32) | (u64)(unsigned)(~tag ^ (0x9E3779B9u * (slot + 1u))); }
constexpr unsigned kSlotSort = 0u;       // + 32 * role + workgroup of the sort          -> gran[1 + role][workgroup]
constexpr unsigned kSlotBits = 64u;      // + workgroup of the bit table (<= 416)        -> gran[3 .. 15][workgroup]
constexpr unsigned kSlotVerdict = 512u;  // + the verdict (1 fast tail, 2 K5 proper ran) -> gran[16][kGranVerdict]

template <int KPW, bool FUSED = false>   // keys per workgroup: 64, or 32 where that still is one round of the machine (B = 1, N = 4096: 10.5 -> ? us)
__device__ __forceinline__ void sort_count_body(const float* __restrict__ scores, const float* __restrict__ boxes, int N,
                                                const int* __restrict__ counts, char* ws, gnms_ws_layout L,
                                                long long* __restrict__ order_out, int mode3d, const int blk, const int b, const int role,
                                                const unsigned tag = 0u) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);                          // [NP] all keys of the image (padding ~0: behind everything)
    __shared__ int rk[64];
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int t = threadIdx.x, lane = t & (KPW - 1), seg = t / KPW;
    const int NP = (N + 63) & ~63;
    const float* s = scores + (size_t)b * N;
    const float* bx = boxes ? boxes + (size_t)b * N * 4 : nullptr;
    float zlo = 0.0f, zscale = 0.0f;
    if (role == 1 && mode3d > 1) block_z_bands(reinterpret_cast<const float4*>(bx), n, mode3d, &zlo, &zscale);
    const int k = blk * KPW + lane;                                    // (k < NP)
    // key k is input index k: what leaves with it -- its score, its box -- needs no index from the sorted keys and is requested here, in front
    // of everything (round 6: fetched by `idx` behind the count it was one more exposed memory round trip at the end of every workgroup)
    float pre_s = 0.0f;
    float4 pre_b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (seg == 0 && k < n) {
        if (role == 0) pre_s = s[k];
        if (bx) pre_b = reinterpret_cast<const float4*>(bx)[k];
    }
    int flag = 1;                                                      // role 1: "every box is plain" (workgroup 0 decides, below)
    if (role == 0) {
        for (int i = t; i < NP; i += 1024) keys[i] = (i < n) ? sort_key_of(0, s, bx, i, mode3d, zlo, zscale) : ~0ull;
    } else {
        // (the plain test on the box the key is made of: as a pass of its own over the boxes it was a second memory round trip of workgroup 0,
        // the one every launch of this kernel ends with)
        for (int i = t; i < NP; i += 1024) {
            u64 key = ~0ull;
            if (i < n) {
                const float4 v = reinterpret_cast<const float4*>(bx)[i];
                key = column_key(v, i, mode3d, zlo, zscale);
                flag &= box_orders_plainly(v) ? 1 : 0;
            }
            keys[i] = key;
        }
    }
    if (t < 64) rk[t] = 0;
    __syncthreads();
    const u64 mine = keys[k];
    {
        const int per = NP / (1024 / KPW);                             // a multiple of 4 (KPW = 32: the launcher asks for NP % 128 == 0)
        const u64* p = keys + seg * per;
        int c = 0;
        for (int j = 0; j < per; j += 4) c += (p[j] < mine ? 1 : 0) + (p[j + 1] < mine ? 1 : 0) + (p[j + 2] < mine ? 1 : 0) + (p[j + 3] < mine ? 1 : 0);
        if (c) atomicAdd(&rk[lane], c);
    }
    // what only one workgroup per image and role does: the "already sorted" / "every box is plain" flags (every workgroup holds every key,
    // so the first one decides alone), the counters, the call counter, the hand-off granules
    if (blk == 0 && role == 0) {
        for (int i = t; i + 1 < n; i += 1024) flag &= keys[i] < keys[i + 1];          // ascending keys in index order = the scores came in sorted
    }
    const int all = __syncthreads_and(flag);                           // (also: every count has landed in rk)
    if (blk == 0) {
        if (role == 0) {
            if constexpr (FUSED) {
                if (t < 8 && !(boxes && t == 6)) coh_store(I.misc + t, (t == 2) ? all : 0);
            } else {
                if (t < 8 && !(boxes && t == 6)) I.misc[t] = (t == 2) ? all : 0;           // ([6]: the x sort's)
                if (t == 8) I.misc[8] = gnms_next_epoch(I.misc[8]);
                for (int i = t; i < 17 * 32; i += 1024) I.gran[i] = 0ull;
            }
        } else if (t == 0) {
            if constexpr (FUSED) coh_store(I.misc + 6, all ? 0 : 1); else I.misc[6] = all ? 0 : 1;
        }
    }
    if (seg == 0) {
        if (k < n) {
            const int rank = rk[lane];
            const int idx = (int)((unsigned)mine & (role == 0 ? 0xffffffffu : kColIdxMask));
            if (role == 0) {
                if constexpr (FUSED) {
                    coh_store(I.order + rank, idx);
                    coh_store(I.rankof + idx, rank);
                    coh_store(I.sscore + rank, pre_s);
                    if (boxes) coh_store_f4(I.rbox + rank, pre_b);
                } else {
                    I.order[rank] = idx;
                    I.rankof[idx] = rank;
                    I.sscore[rank] = pre_s;
                    if (boxes) I.rbox[rank] = pre_b;
                }
                if (order_out) order_out[(size_t)b * N + rank] = idx;
            } else {
                if constexpr (FUSED) {
                    coh_store(I.xidx + rank, idx);
                    coh_store_f4(I.xbox + rank, pre_b);
                } else {
                    column_store(I, rank, idx, pre_b, mode3d);
                }
            }
        } else if (k < N && role == 0) {                               // padding ranks map to themselves (order is a permutation of [0, N))
            if constexpr (FUSED) { coh_store(I.order + k, k); coh_store(I.rankof + k, k); coh_store(I.sscore + k, 0.0f); }
            else { I.order[k] = k; I.rankof[k] = k; I.sscore[k] = 0.0f; }
            if (order_out) order_out[(size_t)b * N + k] = k;
        }
    }
    if constexpr (FUSED) {
        __builtin_amdgcn_s_waitcnt(0x0f70);                            // vmcnt(0): this wave's stores are acknowledged ...
        __syncthreads();                                               // ... every wave's are
        if (t == 0) coh_store(I.gran + (size_t)(1 + role) * 32 + blk, strong_gran(tag, kSlotSort + 32u * (unsigned)role + (unsigned)blk));
    }
}

template <int KPW>
__global__ __launch_bounds__(1024) void sort_count_kernel(const float* __restrict__ scores, const float* __restrict__ boxes, int N,
                                                          const int* __restrict__ counts, char* ws, gnms_ws_layout L,
                                                          long long* __restrict__ order_out, int mode3d) {
    sort_count_body<KPW>(scores, boxes, N, counts, ws, L, order_out, mode3d, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// ------------------------------------------------------------------------------------------------
// K2: threshold bit matrix -- the ONE full read of the N x N fp32 matrix (HBM-read bound).
// One wave = 64 rank-rows x 256 input columns.  Rows order[64*kb + r] are contiguous 4N-byte streams
// whatever the permutation, so the row gather is free; lane t accumulates, for each of its 4 columns c,
//     word(c) = sum_r  !(iou[order[64 kb + r]][c] <= thr) << r
// i.e. column c of the thresholded matrix with its bits already in RANK order, and scatters it to
// W[kb][rankof[c]] so that every later kernel reads W contiguously (rank x rank space).
// Tuning (tools/bw_variants.hip, MI355X): ROLLED row loops with 8 x 1-KiB non-temporal loads in flight per
// wave keep the kernel at 63 VGPRs = 8 waves/SIMD; that beats the fully unrolled 64-load version
// (256 VGPRs, 1 wave/SIMD) by 18 % and reaches 5.6 TB/s with the scatter (6.0 TB/s without).
// ------------------------------------------------------------------------------------------------
constexpr int kMaskWaves = 8;       // waves per workgroup: 8 x 256 = 2048 columns
constexpr int kMaskRB = 8;          // rows (1-KiB loads) in flight per wave

__device__ __forceinline__ float4 load_nt_f4(const float* p) {
    float4 v;
    v.x = __builtin_nontemporal_load(p);
    v.y = __builtin_nontemporal_load(p + 1);
    v.z = __builtin_nontemporal_load(p + 2);
    v.w = __builtin_nontemporal_load(p + 3);
    return v;
}

template <bool VEC, int WAVES = kMaskWaves, int RB = kMaskRB>
__global__ __launch_bounds__(WAVES * 64) void bitmask_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                                  float thr, char* ws, gnms_ws_layout L, int full) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int kb = blockIdx.y;
    const int n = gnms_count(counts, b, N);
    const int k0 = kb * 64;
    // column chunk rotated by the row block: with pre-sorted scores half the tiles exit below, and an un-rotated grid
    // would leave that work on every other XCD (blocks are dealt round-robin to the 8 XCDs)
    const int bx = (blockIdx.x + kb) % gridDim.x;
    const int c0 = (bx * WAVES + wave) * 256;
    // `full` with ONE 16-wave workgroup per rank block (N <= 4096): the row of W is collected in LDS, by column rank, and leaves as
    // one coalesced write -- N scattered 8-byte stores per row block otherwise (the triangle alone is half of them)
    __shared__ u64 rowbuf[(WAVES == 16) ? 4096 : 1];
    const bool rowbuffered = WAVES == 16 && full && gridDim.x == 1 && L.NC <= 4096;
    if (k0 >= n) return;                                                 // (workgroup-uniform)
    ImgPtrs I = img_ptrs(ws, L, b);
    const bool ident = I.misc[2] != 0;                                   // scores were already sorted: rank == input index
    const bool idle = c0 >= n || (ident && c0 >= k0 + 64);               // (pre-sorted: no leader of this row block lives in these columns)
    if (idle && !rowbuffered) return;
    const float* m = iou + (size_t)b * N * ld;

    const int myrank = k0 + lane;
    const int myrow = (myrank < n) ? I.order[myrank] : I.order[k0];      // clamp to a valid row; bits masked below
    const int nrows = min(64, n - k0);
    const u64 rowmask = (nrows >= 64) ? ~0ull : ((1ull << nrows) - 1ull);

    int col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) col[j] = VEC ? (c0 + 4 * lane + j) : (c0 + lane + 64 * j);
    // VEC reads 16 B at col[0]; legal while col[0]+3 < ld (ld % 4 == 0).  Columns >= n produce words nobody stores.
    const bool active = VEC ? (col[0] + 3 < ld) : true;

    unsigned wd[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if (!idle) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll 1
        for (int rb = 0; rb < 32; rb += RB) {
            float v[RB][4];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int row = __builtin_amdgcn_readlane(myrow, half * 32 + rb + u);
                const float* p = m + (size_t)row * ld;
                if (VEC) {
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (active) t = load_nt_f4(p + col[0]);
                    v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = (col[j] < n) ? __builtin_nontemporal_load(p + col[j]) : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const unsigned bit = 1u << (rb + u);
#pragma unroll
                for (int j = 0; j < 4; ++j) wd[half][j] |= !(v[u][j] <= thr) ? bit : 0u;   // lib/groomed_nms.py:250 (NaN -> removed)
            }
        }
    }
    }
    // scatter each column word to the column's RANK: downstream kernels then read W contiguously
    u64* Wk = I.W + (size_t)kb * L.NC;
    int rk[4];
    if (ident) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rk[j] = col[j];
    } else if (VEC && col[3] < n) {
        const int4 t = *reinterpret_cast<const int4*>(I.rankof + col[0]);
        rk[0] = t.x; rk[1] = t.y; rk[2] = t.z; rk[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) rk[j] = (col[j] < n) ? I.rankof[col[j]] : col[j];       // (padding: rank == index)
    }
    if (rowbuffered) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 w = (col[j] < n && !idle) ? ((((u64)wd[1][j] << 32) | wd[0][j]) & rowmask) : 0ull;
            if (col[j] < L.NC && rk[j] >= 0 && rk[j] < L.NC) rowbuf[rk[j]] = w;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L.NC; i += WAVES * 64) Wk[i] = rowbuf[i];
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // only words a leader scan can read: the column must outrank some row of the block (the others are never looked at) -- or,
        // `full`, the whole row: wsym_check_kernel then decides whether the thresholded matrix is symmetric and the scan may pull
        if (col[j] < n && (full || rk[j] < k0 + 64)) Wk[rk[j]] = (((u64)wd[1][j] << 32) | wd[0][j]) & rowmask;
    }
}

// K2 for few, small images (round 5): the kernel above gives a wave 64 rows x 256 columns, eight batches of eight 1-KiB loads one after the
// other -- at N = 500, B = 1 that is 16 busy waves on the whole machine and eight memory latencies in a row (8.8 us for 1 MB).  Here the 64
// rows of a rank block are dealt to the eight waves of a workgroup, eight rows each: ONE batch of loads per wave.  A wave's eight rows
// are eight consecutive bits = one BYTE of the column's word, so the partial results meet in LDS as the bytes of the words (byte stores),
// and four of the waves scatter the finished words.  Stores exactly the words bitmask_kernel<true> stores.  Needs ld % 4 == 0 and a
// 16-byte aligned matrix (VEC).
__global__ __launch_bounds__(512) void bitmask_small_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                            float thr, char* ws, gnms_ws_layout L, int full) {
    __shared__ u64 words[256];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int kb = blockIdx.y;
    const int n = gnms_count(counts, b, N);
    const int k0 = kb * 64;
    if (k0 >= n) return;
    const int c0 = (int)((blockIdx.x + kb) % gridDim.x) * 256;           // (rotated like bitmask_kernel's chunks)
    ImgPtrs I = img_ptrs(ws, L, b);
    const bool ident = I.misc[2] != 0;
    if (c0 >= n || (ident && c0 >= k0 + 64)) return;                     // (workgroup-uniform)
    const float* m = iou + (size_t)b * N * ld;
    const int myrank = k0 + lane;
    const int myrow = (myrank < n) ? I.order[myrank] : I.order[k0];
    const int nrows = min(64, n - k0);
    const u64 rowmask = (nrows >= 64) ? ~0ull : ((1ull << nrows) - 1ull);
    // the scatterers' ranks, asked for before the rows so that the two latencies overlap
    const int mycol = c0 + (int)threadIdx.x;
    int rk = mycol;
    if (threadIdx.x < 256 && mycol < n && !ident) rk = I.rankof[mycol];
    const int col0 = c0 + 4 * lane;
    const bool active = col0 + 3 < ld;
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int row = __builtin_amdgcn_readlane(myrow, wave * 8 + u);
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) v[u] = load_nt_f4(m + (size_t)row * ld + col0);
    }
    unsigned by[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const unsigned bit = 1u << u;
        by[0] |= !(v[u].x <= thr) ? bit : 0u;                            // lib/groomed_nms.py:250 (NaN -> removed)
        by[1] |= !(v[u].y <= thr) ? bit : 0u;
        by[2] |= !(v[u].z <= thr) ? bit : 0u;
        by[3] |= !(v[u].w <= thr) ? bit : 0u;
    }
    unsigned char* bytes = reinterpret_cast<unsigned char*>(words);
#pragma unroll
    for (int j = 0; j < 4; ++j) bytes[(4 * lane + j) * 8 + wave] = (unsigned char)by[j];
    __syncthreads();
    if (threadIdx.x < 256 && mycol < n && (full || rk < k0 + 64)) (I.W + (size_t)kb * L.NC)[rk] = words[threadIdx.x] & rowmask;
}

// ------------------------------------------------------------------------------------------------
// K2s: is the thresholded matrix SYMMETRIC?  (matrix-in layer, round 3.)  The callers of differentiable_nms(scores, iou) hand it
// iou(boxes, boxes) -- symmetric -- but the interface does not say so, and the general scan (candidates push leader by leader: ~130
// cycles per leader on one wave) is what makes uniform boxes cost 132 us in K3 where the pulling scan of the from-boxes layer takes
// 25-30.  With the rows of W stored in full (bitmask_kernel, `full`) symmetry is a property of W alone: for every pair of rank blocks
// (kb <= jb) the 64 x 64 bit block W[kb][64 jb ..] must be the transpose of W[jb][64 kb ..].  One wave per pair: both blocks are one
// coalesced 512-byte load each, the transpose is six butterfly stages across the lanes.  The verdict lands in misc[3] (1 = not
// symmetric, or not decidable: pre-sorted scores store only the triangle); K3 / K4 read it when called with sym = 2.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 transpose64(u64 x, int lane) {            // lane r, bit c  <->  lane c, bit r
#pragma unroll
    for (int j = 32; j >= 1; j >>= 1) {
        const u64 m = j == 32 ? 0x00000000ffffffffull : j == 16 ? 0x0000ffff0000ffffull : j == 8 ? 0x00ff00ff00ff00ffull
                    : j == 4 ? 0x0f0f0f0f0f0f0f0full : j == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;   // bit positions with bit j clear
        const u64 t = (u64)__shfl_xor((unsigned long long)x, j, 64);
        x = (lane & j) ? (((t >> j) & m) | (x & ~m)) : ((x & m) | ((t << j) & ~m));
    }
    return x;
}

// (round 4: no density gate any more -- with the scan on one workgroup per super-block the symmetric path is the faster one for dense
// images too)
// one pair (kb <= jb) of 64 x 64 bit blocks, pair index pr row-major over the upper triangle, by one wave: true = not each other's transpose
__device__ __forceinline__ bool wsym_pair_bad(const ImgPtrs& I, const gnms_ws_layout& L, const int n, const int nb, const int pr, const int lane) {
    auto rows_of = [&](int blk) { const int r = min(64, n - blk * 64); return r >= 64 ? ~0ull : ((1ull << r) - 1ull); };
    // pr -> (kb, jb): pairs before row kb = kb * nb - kb (kb - 1) / 2
    const float a = (float)(2 * nb + 1);
    int kb = (int)((a - sqrtf(a * a - 8.0f * (float)pr)) * 0.5f);
    kb = kb < 0 ? 0 : (kb >= nb ? nb - 1 : kb);
    while (kb > 0 && kb * nb - kb * (kb - 1) / 2 > pr) --kb;
    while ((kb + 1) * nb - (kb + 1) * kb / 2 <= pr) ++kb;
    const int jb = kb + (pr - (kb * nb - kb * (kb - 1) / 2));
    // A: rows = ranks of block kb (bits), columns = ranks of block jb (lanes); Bm the other way round
    const u64 A = (jb * 64 + lane < n) ? (I.W[(size_t)kb * L.NC + jb * 64 + lane] & rows_of(kb)) : 0ull;
    const u64 Bm = (kb * 64 + lane < n) ? (I.W[(size_t)jb * L.NC + kb * 64 + lane] & rows_of(jb)) : 0ull;
    return A != transpose64(Bm, lane);
}

// (four pairs per wave, eight loads in flight, measured slower in round 3: 13.0 against 10.7 us at B = 8, N = 4096)
__global__ __launch_bounds__(256) void wsym_check_kernel(int N, const int* __restrict__ counts, char* ws, gnms_ws_layout L) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int nb = (n + 63) >> 6;
    const int pairs = nb * (nb + 1) / 2;
    const int pr0 = blockIdx.x * 4 + (threadIdx.x >> 6);                         // pair index, row-major over kb <= jb
    if (pr0 >= pairs) return;
    if (I.misc[2] != 0) { if (pr0 == 0 && lane == 0) I.misc[3] = 1; return; }   // pre-sorted scores: W holds the triangle only
    const bool bad = wsym_pair_bad(I, L, n, nb, pr0, lane);
    if (__any(bad) && lane == 0) I.misc[3] = 1;                          // (every writer stores 1; the sort zeroed it)
}

// The same check as a ROLE of the tail launch (round 5, the matrix-in layer with the fast tail): workgroup `wg` of `nwg` checker workgroups
// (the FIRST workgroups of the grid: they wait for nobody) takes every nwg-th group of 16 pairs of every image, while the chain workgroups
// already run the symmetric scan on the assumption that the check will pass; the image's last chain workgroup waits for the verdict --
// misc[7] == nwg workgroups done, misc[3] != 0: some pair failed -- before anything is final, and falls back to the general scan if it has to.
// (As a launch of its own the check cost 9.9 us in front of the tail, B = 8, N = 4096.)  Images whose scores came in sorted (misc[2]: W
// holds the triangle only) are not checked at all: everybody reads misc[2] and takes the general scan.
__device__ __forceinline__ void wsym_check_in_launch(int N, const int* __restrict__ counts, char* ws, gnms_ws_layout L, const int B, const int wg,
                                                     const int nwg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int b = 0; b < B; ++b) {
        const int n = gnms_count(counts, b, N);
        ImgPtrs I = img_ptrs(ws, L, b);
        if (I.misc[2] != 0) continue;
        const int nb = (n + 63) >> 6;
        const int pairs = nb * (nb + 1) / 2;
        bool bad = false;
        for (int pr = wg * nw + wave; pr < pairs; pr += nwg * nw) bad |= wsym_pair_bad(I, L, n, nb, pr, lane);
        if (__any(bad) && lane == 0) __hip_atomic_store(I.misc + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0x0f70);                              // vmcnt(0): the flag is out before this workgroup counts as done
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(I.misc + 7, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// K2b: threshold bit matrix straight from the boxes (the from-boxes path: no read of the N x N fp32 matrix).
// Wave tile = 64 rank-rows (block kb) x 256 columns; the overlap of each pair is recomputed from the two boxes and
// thresholded in registers, bit-identical to thresholding the matrix.  Bound: fp32 VALU; HBM traffic is 16 N bytes in
// and <= N^2/8 bytes out per image.
// ------------------------------------------------------------------------------------------------
// Columns in X ORDER + row culling.
// The column a lane owns only decides WHICH word it writes (W[kb][rank of the column]); the order in which columns are
// dealt to lanes is free.  Dealing them by ascending x centre (sort_boxes_by_x) makes the 256 columns of a wave tile
// spatial neighbours: their hull is a narrow strip of the image, and a row box that does not reach into the hull has
// intersection 0 with all 256 columns -> its bit is 0 in every word and the row is skipped (exact: with finite positive
// areas and thr >= 0, inter = +0 and uni > 0 give 0 <= thr).  On detector-like inputs (boxes ~100 px on a 1760 px wide
// canvas) a tile keeps 15-20 % of its 64 rows.  The price: the rank triangle can no longer be skipped per tile (a tile
// holds columns of every rank); words no leader scan reads (column rank >= 64 (kb+1)) are computed but not stored.
__device__ __forceinline__ float wave_min_f(float v) { return gnms_wave_min_f(v); }   // DPP network: no ds_bpermute round trips
__device__ __forceinline__ float wave_max_f(float v) { return gnms_wave_max_f(v); }

// CPL = columns per lane: 4 (wave tile 64 x 256) when there are plenty of tiles, 1 (64 x 64) for small problems, where the 64-row
// chain of a wave is the critical path and four times as many waves share it.
// KBW = rank blocks per wave: 1, or 4 for large images (the column side -- gathers, ranks, hull -- is then paid once per 256 rows
// instead of once per 64: at N=16384 a tile keeps ~6 of its 64 rows, so that fixed part dominated).
// ROWBUF (N <= 4096, KBW = 1): the workgroup is 16 waves = ALL column chunks of ONE rank block; the words go to an LDS copy of the
// row W[kb][.] (indexed by column rank) and leave as one coalesced write of the 64 (kb+1) words a leader scan can read.  Without it
// every lane scatters its 8-byte words to global memory: 1 M scattered stores per launch at B=8, N=4096 = ~15 us of store
// throughput, about a third of it exposed.
// CHUNKLOOP (ROWBUF, N > 4096): a wave walks the column chunks wave, wave + 16, ... of its rank block (at N = 16384 four of them), all
// into the one LDS row of N words (128 KiB), which then leaves as one coalesced write: since round 2 the rows of W are stored in full
// (the scan pulls), so the scatter version issues N^2 / 64 scattered 8-byte stores per image -- 256 MB per step at B = 8, N = 16384.
template <int CPL, int KBW, bool ROWBUF, bool CHUNKLOOP = false>
__device__ __forceinline__ void bitmask_boxes_body(const float* __restrict__ boxes, int N, const int* __restrict__ counts, float thr, char* ws,
                                                   gnms_ws_layout L, const int b, const int bx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = gnms_count(counts, b, N);
    constexpr int kCols = 64 * CPL;
    const int nchunk = (N + kCols - 1) / kCols;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* rowbuf = reinterpret_cast<u64*>(smem);                       // ROWBUF: [NC] words of row kb, by column rank
    ImgPtrs I = img_ptrs(ws, L, b);
#ifdef GNMS_TIMING   // developer (tools/bits_ticks.py): phase ticks of wave 0 / wave 15 of rank block 32 of image 0, into xsol
    long long bt__[6]; int bti__ = 0;
#define GNMS_BT() do { bt__[bti__++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define GNMS_BT() do {} while (0)
#endif
    GNMS_BT();
#ifdef GNMS_TIMING
    const long long rt0__ = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // ROWBUF with KBW = 2 (round 6; B = 8, N = 4096): a workgroup takes TWO rank blocks, their rows of W side by side in LDS.  With one
    // block per workgroup the launch was 512 workgroups, two per CU, each pulling all 4096 column boxes, their indices and ranks (96 KiB)
    // through the CU for ~3 us of arithmetic: 48 MB of L2 -> CU traffic in the first microseconds, the late waves' loads 6-8 us behind, and
    // the CU's second workgroup (younger waves) finishing 7-8 us after the first (profiles/r06j_bits_timeline.txt: image 0 done at 15 us,
    // image 7 at 23).  256 workgroups of two blocks pull half the bytes and have their CU to themselves.
    if (ROWBUF) {
        const int kbr = bx * KBW;                             // KBW rank blocks per workgroup, wave w = column chunk w
        if (kbr * 64 >= n) return;
        for (int i = threadIdx.x; i < KBW * L.NC; i += blockDim.x) rowbuf[i] = 0ull;
        __syncthreads();
    }
    GNMS_BT();
    const int nwaves = blockDim.x >> 6;
    for (int cq = 0; cq < (CHUNKLOOP ? (nchunk + nwaves - 1) / nwaves : 1); ++cq) {
    const int tile = ROWBUF ? bx * nchunk + wave : bx * 4 + wave;
    const int kbg = ROWBUF ? bx : tile / nchunk;        // kbg = group of KBW consecutive rank blocks
    const int chunk = ROWBUF ? wave + cq * nwaves : tile - kbg * nchunk;   // ROWBUF: one chunk per wave (CHUNKLOOP: every 16th)
    const int c0 = chunk * kCols;
    const bool idle = (kbg * KBW >= L.NB || kbg * KBW * 64 >= n || c0 >= n || chunk >= nchunk);   // (ragged images)
    if (!ROWBUF && idle) return;
    if (!idle) {                                                     // (a chunk LOOP here cost 25 % in code quality at N = 4096: 24.5 -> 31 us)
    // the row boxes of the first rank block are requested before the column side is worked on
    float4 rb_next = I.rbox[min(kbg * KBW * 64 + lane, n - 1)];     // (rank order, written by the score sort: no order -> box gather)
    float4 cb[CPL];
    float carea[CPL];
    int crank[CPL];
    bool cok = true;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        // (round 6: column 64 j + lane, not CPL lane + j.  With four CONSECUTIVE columns per lane the lanes of one load instruction sat 64 bytes
        // apart -- 64 requests to the texture pipe per instruction, four instructions over the same lines -- and the sixteen waves' column
        // phases queued behind each other at ~0.7 k ticks a wave (wave 0: 8 k, wave 15: 18 k, profiles/r06k_bits_timeline.txt); which lane
        // holds which column is free: the word goes to the column's rank)
        const int p = c0 + 64 * j + lane;
        const int pp = p < n ? p : n - 1;                             // clamped duplicates: harmless in the hull, never stored
        cb[j] = I.xbox[pp];
        crank[j] = (p < n) ? I.rankof[I.xidx[pp]] : 0x7fffffff;
        carea[j] = (cb[j].z - cb[j].x) * (cb[j].w - cb[j].y);
        cok &= (carea[j] > 0.0f) && (carea[j] < INFINITY);
    }
    // Decision !(fl(inter/uni) <= thr) WITHOUT the division.  With d = fma(-thr, uni, inter) (one rounding, sign exact):
    //   inter/uni - thr = d/uni,  so  |d| > guard*uni  puts the exact quotient more than `guard` (8 ulp of the threshold)
    // away from thr, hence its fp32 rounding on the same side, and the pair is decided by the sign of d.  That needs
    // uni > 0 and finite, which holds whenever both boxes have a positive finite area (inter <= min(area) in fp32 as in
    // exact arithmetic because subtraction/multiplication round monotonically, so uni >= max(area) > 0): checked once per
    // column box and per row.  Rows/columns that fail, and the pairs inside the guard band, take the exact IEEE division.
    // (two workgroups per CU leave 64 VGPRs: the columns' ranks, not needed before the words are parked, wait in LDS meanwhile)
    constexpr bool STASH = ROWBUF && !CHUNKLOOP && CPL == 4;
    int* const crank_lds = reinterpret_cast<int*>(smem + (size_t)KBW * L.NC * 8) + threadIdx.x;   // [CPL][1024] (behind the KBW rows)
    if (STASH) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) crank_lds[j * 1024] = crank[j];
    }
    const float guard = fmaxf(fabsf(thr), 1.0f) * 9.6e-7f;            // 8 ulp at the threshold's magnitude (>= 1 for tiny thresholds)
    const bool cols_ok = __all(cok);
    const bool plain_img = I.misc[6] == 0;                            // (sort_boxes_by_x / sort_merge_body: no box that is not plain)
    // hull of the tile's columns
    float hx0 = cb[0].x, hx1 = cb[0].z, hy0 = cb[0].y, hy1 = cb[0].w;
#pragma unroll
    for (int j = 1; j < CPL; ++j) { hx0 = fminf(hx0, cb[j].x); hx1 = fmaxf(hx1, cb[j].z); hy0 = fminf(hy0, cb[j].y); hy1 = fmaxf(hy1, cb[j].w); }
    hx0 = wave_min_f(hx0); hy0 = wave_min_f(hy0); hx1 = wave_max_f(hx1); hy1 = wave_max_f(hy1);
    const bool cull = cols_ok && (thr >= 0.0f);
    GNMS_BT();
#pragma unroll 1
    for (int kw = 0; kw < KBW; ++kw) {
        const int kb = kbg * KBW + kw;
        const int k0 = kb * 64;
        if (kb >= L.NB || k0 >= n) break;
        const float4 rb = rb_next;
        if (KBW > 1 && kw + 1 < KBW && kb + 1 < L.NB && k0 + 64 < n) rb_next = I.rbox[min(k0 + 64 + lane, n - 1)];   // next block's rows
        const float rarea = (rb.z - rb.x) * (rb.w - rb.y);
        const int nrows = min(64, n - k0);
        const bool row_fine = (rarea > 0.0f) && (rarea < INFINITY);
        const u64 rows_ok = __ballot(row_fine);
        // the rows that reach into the hull
        const bool reaches = (rb.z > hx0) && (rb.x < hx1) && (rb.w > hy0) && (rb.y < hy1);
        const u64 active = __ballot((lane < nrows) && (!(cull && row_fine) || reaches));
        unsigned wd[2][CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) { wd[0][j] = 0u; wd[1][j] = 0u; }
        // PLAIN images (misc[6] == 0: every box finite, x2 >= x1, y2 >= y1, no negative zero; decided by the x sort): the intersection
        // as iou_tile.h derives it -- w = med3(min3(fl(ax2 - bx1), fl(bx2 - ax1), cw), 0, rw), bit for bit the reference's
        // relu(min - max) -- packed over column pairs, and the decided bit taken from the SIGN of thr * uni - inter (an arithmetic shift
        // and one v_and_or per entry instead of compare, select, or): 11.5 VALU slots per entry instead of 17.
        bool fast_rows = false;
        if constexpr (CPL == 4) fast_rows = plain_img && cols_ok && rows_ok == ~0ull;   // (lanes past the last row hold copies of it)
        if constexpr (CPL == 4) if (fast_rows) {
            using gnms_iou::gnms_f2;
            gnms_f2 cx1[2], cy1[2], cx2[2], cy2[2], cw[2], ch[2], ca[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cx1[j >> 1][j & 1] = cb[j].x; cy1[j >> 1][j & 1] = cb[j].y; cx2[j >> 1][j & 1] = cb[j].z; cy2[j >> 1][j & 1] = cb[j].w;
                cw[j >> 1][j & 1] = cb[j].z - cb[j].x; ch[j >> 1][j & 1] = cb[j].w - cb[j].y;
                ca[j >> 1][j & 1] = carea[j];
            }
            const float rw = rb.z - rb.x, rh = rb.w - rb.y;
            const gnms_f2 sthr = {thr, thr}, sguard = {guard, guard};
            // one row against the lane's four columns: nd = thr * uni - inter (its sign decides), true if some pair lies in the guard band
            auto row_core = [&](int r, gnms_f2 (&inter)[2], gnms_f2 (&uni)[2], gnms_f2 (&nd)[2]) -> bool {
                const float ax1 = gnms_iou::bcast(rb.x, r), ay1 = gnms_iou::bcast(rb.y, r), ax2 = gnms_iou::bcast(rb.z, r), ay2 = gnms_iou::bcast(rb.w, r);
                const float aw = gnms_iou::bcast(rw, r), ah = gnms_iou::bcast(rh, r), aa = gnms_iou::bcast(rarea, r);
                const gnms_f2 sx1 = {ax1, ax1}, sy1 = {ay1, ay1}, sx2 = {ax2, ax2}, sy2 = {ay2, ay2}, sa = {aa, aa};
                bool unsure = false;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const gnms_f2 dx1 = sx2 - cx1[p], dx2 = cx2[p] - sx1;
                    const gnms_f2 dy1 = sy2 - cy1[p], dy2 = cy2[p] - sy1;
                    const gnms_f2 w = {gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dx1.x, dx2.x, cw[p].x), aw),
                                       gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dx1.y, dx2.y, cw[p].y), aw)};
                    const gnms_f2 h = {gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dy1.x, dy2.x, ch[p].x), ah),
                                       gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dy1.y, dy2.y, ch[p].y), ah)};
                    inter[p] = w * h;
                    uni[p] = (sa + ca[p]) - inter[p];                     // row box is `a`, column (leader) box is `b`
                    nd[p] = __builtin_elementwise_fma(sthr, uni[p], -inter[p]);   // = -fma(-thr, uni, inter) exactly (one rounding either way)
                    const gnms_f2 g = sguard * uni[p];
                    unsure |= !(fabsf(nd[p].x) > g.x) || !(fabsf(nd[p].y) > g.y);   // also true for NaN
                }
                return __any(unsure);
            };
            // The loop over the surviving rows is BRANCH-FREE, two rows per trip: every bit is taken from the sign of nd, the rows
            // with a pair in the guard band are only noted (`redo`) and re-decided with the IEEE division behind the loop.  The kernel
            // is bound by its slowest waves (dense x stretches keep 3x the rows of the mean), not by VALU throughput: what counts
            // is the latency of ONE wave's row chain, and two independent rows in flight shorten it.
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                unsigned todo = (unsigned)(half ? (active >> 32) : (active & 0xffffffffull));
                unsigned redo = 0u;
                while (todo) {                                        // wave-uniform
                    const int r0 = __builtin_ctz(todo);
                    todo &= todo - 1u;
                    const int r1 = todo ? __builtin_ctz(todo) : r0;   // (odd count: the last row twice -- the same bits again)
                    todo &= todo - 1u;                                // (0 stays 0)
                    gnms_f2 i0[2], u0[2], n0[2], i1[2], u1[2], n1[2];
                    const bool q0 = row_core(half * 32 + r0, i0, u0, n0);
                    const bool q1 = row_core(half * 32 + r1, i1, u1, n1);
                    const unsigned b0 = 1u << r0, b1 = 1u << r1;
                    redo |= (q0 ? b0 : 0u) | (q1 ? b1 : 0u);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                     // decided pairs: d = -nd != 0, the bit is "nd negative"
                        wd[half][j] |= (unsigned)(__float_as_int(n0[j >> 1][j & 1]) >> 31) & b0;
                        wd[half][j] |= (unsigned)(__float_as_int(n1[j >> 1][j & 1]) >> 31) & b1;
                    }
                }
                while (redo) {                                        // rare: a few rows per image
                    const int rr = __builtin_ctz(redo);
                    redo &= redo - 1u;
                    gnms_f2 ii[2], uu[2], nn[2];
                    row_core(half * 32 + rr, ii, uu, nn);
                    const unsigned bit = 1u << rr;
#pragma unroll
                    for (int j = 0; j < 4; ++j) wd[half][j] = (wd[half][j] & ~bit) | (!(ii[j >> 1][j & 1] / uu[j >> 1][j & 1] <= thr) ? bit : 0u);
                }
            }
        }
        if (!fast_rows) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned todo = (unsigned)(half ? (active >> 32) : (active & 0xffffffffull));
            while (todo) {                                            // wave-uniform loop over the surviving rows
                const int rr = __builtin_ctz(todo);
                todo &= todo - 1u;
                const int r = half * 32 + rr;
                const float ax1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.x), r));
                const float ay1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.y), r));
                const float ax2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.z), r));
                const float ay2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.w), r));
                const float aa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rarea), r));
                const unsigned bit = 1u << rr;
                float inter4[CPL], uni4[CPL], d4[CPL];
                bool unsure = false;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    // (v_min / v_max issued directly, the row coordinate straight from its SGPR: fminf / fmaxf on loaded values cost a
                    // canonicalising v_max x, x each -- 12 of the ~85 VALU instructions of a row; same values, iou3d_pair.h)
                    const float w = fmaxf(gnms_iou3d::vmin_s(ax2, cb[j].z) - gnms_iou3d::vmax_s(ax1, cb[j].x), 0.0f);
                    const float h = fmaxf(gnms_iou3d::vmin_s(ay2, cb[j].w) - gnms_iou3d::vmax_s(ay1, cb[j].y), 0.0f);
                    inter4[j] = w * h;
                    uni4[j] = (aa + carea[j]) - inter4[j];                // row box is `a`, column (leader) box is `b`
                    d4[j] = __builtin_fmaf(-thr, uni4[j], inter4[j]);
                    unsure |= !(fabsf(d4[j]) > guard * uni4[j]);          // also true for NaN
                }
                if (!(cols_ok && ((rows_ok >> r) & 1ull)) || __any(unsure)) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) wd[half][j] |= !(inter4[j] / uni4[j] <= thr) ? bit : 0u;
                } else {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) wd[half][j] |= (d4[j] > 0.0f) ? bit : 0u;
                }
            }
        }
        }   // !fast_rows
        // the FULL row of W: the overlap is symmetric, and with all columns present the leader scan can pull (leaders_body, sym)
        u64* Wk = ROWBUF ? rowbuf + (size_t)kw * L.NC : I.W + (size_t)kb * L.NC;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int cr = STASH ? crank_lds[j * 1024] : crank[j];
            if (cr != 0x7fffffff) Wk[cr] = ((u64)wd[1][j] << 32) | wd[0][j];
        }
    }
    }   // !idle
    }   // chunks of this wave
    GNMS_BT();
    if (ROWBUF) {
        __syncthreads();
        GNMS_BT();
        for (int kw = 0; kw < KBW; ++kw) {
            const int kb = bx * KBW + kw;
            if (kb >= L.NB || kb * 64 >= n) break;
            u64* Wk = I.W + (size_t)kb * L.NC;
            for (int i = threadIdx.x; i < L.NC; i += blockDim.x) Wk[i] = rowbuf[(size_t)kw * L.NC + i];   // the whole row, coalesced
        }
    }
    GNMS_BT();
#ifdef GNMS_TIMING
    if (ROWBUF && !CHUNKLOOP && b == 0 && bx == (int)gridDim.x / 2 && (threadIdx.x == 0 || threadIdx.x == 960)) {
        long long* o = reinterpret_cast<long long*>(I.xsol) + (threadIdx.x ? 8 : 0);
        for (int q = 1; q < 6; ++q) o[q] += bt__[q] - bt__[q - 1];
    }
    // (round 6) the launch's timeline: every workgroup's start and end on the constant-rate clock (s_memrealtime, 100 MHz, one clock for all
    // XCDs), last call only -- tools/bits_ticks.py
    if (ROWBUF && !CHUNKLOOP && threadIdx.x == 0 && bx < 256) {
        long long* o = reinterpret_cast<long long*>(I.xsol) + 32 + 2 * bx;
        o[0] = rt0__; o[1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#endif
}

// (ROWBUF without the chunk loop: two 16-wave workgroups per CU = 8 waves per SIMD, i.e. at most 64 VGPRs -- asked for explicitly)
template <int CPL, int KBW, bool ROWBUF = false, bool CHUNKLOOP = false>
__global__ __launch_bounds__(ROWBUF ? 1024 : 256, (ROWBUF && !CHUNKLOOP) ? (KBW > 1 ? 4 : 8) : 1) void bitmask_boxes_kernel(const float* __restrict__ boxes, int N, const int* __restrict__ counts,
                                                            float thr, char* ws, gnms_ws_layout L) {
    bitmask_boxes_body<CPL, KBW, ROWBUF, CHUNKLOOP>(boxes, N, counts, thr, ws, L, (int)blockIdx.z, (int)blockIdx.x);
}

// The scatter variant (no LDS row) with the row groups dealt to the XCDs, as bitmask_rec3d_culled_kernel does (round 4b): workgroup x runs on
// XCD x mod 8, its four waves are four consecutive column chunks of one row group, consecutive workgroups of an XCD walk the chunks of a
// row group -- every row of W is completed in ONE L2.  Needs a whole number of workgroups per row group (N a multiple of 1024).
template <int KBW>
__global__ __launch_bounds__(256) void bitmask_boxes_pinned_kernel(const float* __restrict__ boxes, int N, const int* __restrict__ counts, float thr,
                                                                   char* ws, gnms_ws_layout L) {
    const int wpk = ((N + 255) >> 8) >> 2;                           // workgroups per row group
    const int x = (int)blockIdx.x, s = x >> 3;
    const int kbg = (s / wpk) * 8 + (x & 7);
    bitmask_boxes_body<4, KBW, false, false>(boxes, N, counts, thr, ws, L, (int)blockIdx.z, kbg * wpk + (s % wpk));
}

// ------------------------------------------------------------------------------------------------
// K2c: threshold bit matrix of the 3D NMS overlap straight from the cuboid records (gnms_forward_with_iou3d: the matrix is an
// output, the layer does not read it back).  Wave tile = 64 rank-rows x 256 RANK columns (records gathered through `order`);
// every pair runs gnms_iou3d::nms_overlap3d -- the very instruction sequence iou3d_nms_fast_kernel wrote the matrix with -- and
// is thresholded in registers: the same bits as bitmask_kernel on that matrix.  Only tiles a leader can reach (column chunk
// below the end of the row block) exist; they are numbered in one dimension so that every wave has equal work.
// Bound: fp32 VALU, ~29 slots per pair on N^2/2 pairs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_tile(int id, int* kb, int* chunk) {     // tiles before row block kb = 4q + r:  2q(q+1) + r(q+1)
    int q = (int)((sqrtf(1.0f + 2.0f * (float)id) - 1.0f) * 0.5f);
    while (2 * q * (q + 1) > id) --q;
    while (2 * (q + 1) * (q + 2) <= id) ++q;
    const int rem = id - 2 * q * (q + 1);
    const int r = rem / (q + 1);
    *kb = 4 * q + r;
    *chunk = rem - r * (q + 1);
}
__host__ __device__ inline int tri_tile_count(int NB) {                // number of reachable tiles for NB row blocks
    const int q = NB >> 2, r = NB & 3;
    return 2 * q * (q + 1) + r * (q + 1);
}

__global__ __launch_bounds__(256) void bitmask_rec3d_kernel(int N, const int* __restrict__ counts, float thr, char* ws, gnms_ws_layout L) {
    using namespace gnms_iou3d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int n = gnms_count(counts, b, N);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= tri_tile_count(L.NB)) return;
    int kb, chunk;
    tri_tile(tile, &kb, &chunk);
    const int k0 = kb * 64, c0 = chunk * 256;
    if (k0 >= n || c0 >= n) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    Cols2 cols[2];
    int col[4];
    unsigned colbad = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        col[j] = c0 + 4 * lane + j;
        const float4* p = reinterpret_cast<const float4*>(I.rec + (size_t)I.order[col[j] < n ? col[j] : n - 1] * kRec);
        const float4 e = p[2];
        cols2_set(cols[j >> 1], j & 1, p[0], p[1], e);
        colbad |= (e.w != 0.0f) ? (1u << j) : 0u;
    }
    const float4* rp = reinterpret_cast<const float4*>(I.rec + (size_t)I.order[min(k0 + lane, n - 1)] * kRec);
    const float4 ru = rp[0], rv = rp[1], re = rp[2];
    const int nrows = min(64, n - k0);
    const bool cols_sane = __all(colbad == 0u);
    unsigned wd[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int rend = min(32, nrows - half * 32);
        for (int rr = 0; rr < rend; ++rr) {
            const int r = half * 32 + rr;
            auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), r)); };
            Row a;
            a.vol = bc(ru.x); a.y0 = bc(ru.y); a.y1 = bc(ru.z); a.x0 = bc(ru.w); a.x1 = bc(rv.x); a.z0 = bc(rv.y); a.z1 = bc(rv.z);
            a.lx = bc(re.x); a.ly = bc(re.y); a.lz = bc(re.z); a.bad = bc(re.w);
            const unsigned bit = 1u << rr;
            float q[4];
            nms_overlap3d_guarded4(a, cols, colbad, cols_sane, thr, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) wd[half][j] |= !(q[j] <= thr) ? bit : 0u;
        }
    }
    u64* Wk = I.W + (size_t)kb * L.NC;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (col[j] < n) Wk[col[j]] = ((u64)wd[1][j] << 32) | wd[0][j];
}

// K2c with spatially ordered columns and row culling (the 3D counterpart of bitmask_boxes_kernel).  For GIoU a DISJOINT pair can
// still exceed the threshold, so "does not reach the hull" is not enough; what holds for two boxes separated along an axis (x or z) by
// a gap g >= 0 is
//     i3 = 0,   q = u3 / (2 vh),   u3 = vol_a + vol_b <= (lx_a + lx_b) * max(ly) * max(lz),   vh >= (lx_a + lx_b + g) * max(ly) * max(lz)
//     =>  q <= (lx_a + lx_b) / (2 (lx_a + lx_b + g))  <=  thr     as soon as     g >= (lx_a + lx_b) * (1 / (2 thr) - 1).
// A row is skipped against a set of columns when its gap to the set's hull satisfies that with the set's largest extent and an ADDITIVE
// 1e-3 on the factor (relative margin 2e-3 thr on q: three orders above the fp32 rounding of the matrix kernel, so the thresholded matrix
// has a 0 there too).  Needs finite positive extents on both sides and thr >= 0.01; everything else is evaluated.
// Round 4b: the cull works per SLOT.  The columns come in (z band, x centre) order (column_key; records in that order: xrec), a wave tile
// is 4 slots of 64 consecutive columns (lane l holds column 64 j + l of slot j), each slot a compact patch in x and z with its own hulls,
// and a row is evaluated only against the slots it can reach -- one column wide (nms_overlap3d_guarded1: v_pk_* run at the scalar rate on
// gfx950, so four single evaluations cost what two packed ones did).  With one x-ordered 256-column hull per tile 27 % of the (row, tile)
// pairs survived at N = 4096 (the gap bound lets boxes ~4 units apart through, the scene is 60 x 55); per slot ~12 % do.
// Tile numbering.  A tile scatters its 256 words over a whole row of W (column p goes to word rank(p)); what completes the 64-byte lines
// is the L2.  `pinned` (N > 4096: rows of >= 64 KiB): the row groups are dealt to the XCDs (workgroup x runs on XCD x mod 8; the four waves of
// a workgroup are four consecutive chunks of one row group, consecutive workgroups of an XCD walk the chunks of a row group), so that every
// row is written through ONE L2 by workgroups that run together: B = 8, N = 16384 409 -> 331 us, 8192 135 -> 129.  Otherwise ROW GROUP
// FASTEST (round 4): what a tile costs is set by its column chunk, hardly by its row group (ranks are spatially random); numbered chunk-fastest
// the heavy tiles of an image recurred on one or two XCDs, where they queued eight deep per CU while the others ran dry (LABNOTES.md 3.2c).
// (Persistent waves claiming tiles from a counter, round 4b: 75 us instead of 46 at N = 4096, 481 instead of 409 at 16384 -- ~1 700 claims
// per image on one address serialise in the memory-side atomic unit; LABNOTES R4b.)
template <int KBW>
__global__ __launch_bounds__(256) void bitmask_rec3d_culled_kernel(int N, const int* __restrict__ counts, float thr, char* ws, gnms_ws_layout L, int pinned) {
    using namespace gnms_iou3d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int n = gnms_count(counts, b, N);
    const int nchunk = (N + 255) >> 8;
    const int nkbg = (L.NB + KBW - 1) / KBW;
    int chunk, kbg;
    if (pinned) {
        const int wpk = (nchunk + 3) >> 2, x = blockIdx.x, s = x >> 3;
        kbg = (s / wpk) * 8 + (x & 7);
        chunk = (s % wpk) * 4 + wave;
    } else {
        const int tile = blockIdx.x * 4 + wave;
        chunk = tile / nkbg;
        kbg = tile - chunk * nkbg;
    }
    const int c0 = chunk * 256;
    if (chunk >= nchunk || kbg >= nkbg || kbg * KBW >= L.NB || kbg * KBW * 64 >= n || c0 >= n) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    const bool thr_ok = (thr >= 0.01f) && (thr < INFINITY);
    const float kappa = fmaxf(1.0f / (2.0f * thr) - 1.0f, 0.0f) + 1e-3f;
    {
    Col1 col[4];
    int crank[4];
    bool colbad[4], sane[4], cullj[4];
    float hx0[4], hx1[4], mlx[4], hz0[4], hz1[4], mlz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = c0 + 64 * j + lane;
        const int pp = p < n ? p : n - 1;
        const float4* rp = reinterpret_cast<const float4*>(I.xrec) + (size_t)pp * 3;
        const float4 u = rp[0], v = rp[1], e = rp[2];
        col1_set(col[j], u, v, e);
        colbad[j] = e.w != 0.0f;
        crank[j] = (p < n) ? I.rankof[I.xidx[pp]] : 0x7fffffff;
        const bool cok = (e.x > 0.0f) && (e.y > 0.0f) && (e.z > 0.0f) && (u.x > 0.0f) && (u.x < INFINITY);   // extents and volume positive, finite
        hx0[j] = wave_min_f(u.w); hx1[j] = wave_max_f(v.x); mlx[j] = wave_max_f(e.x);
        hz0[j] = wave_min_f(v.y); hz1[j] = wave_max_f(v.z); mlz[j] = wave_max_f(e.z);
        cullj[j] = __all(cok) && thr_ok;
        sane[j] = __all(!colbad[j]);
    }
#pragma unroll 1
    for (int kw = 0; kw < KBW; ++kw) {
        const int kb = kbg * KBW + kw;
        const int k0 = kb * 64;
        if (kb >= L.NB || k0 >= n) break;
        const float4* rp = reinterpret_cast<const float4*>(I.rec + (size_t)I.order[min(k0 + lane, n - 1)] * kRec);
        const float4 ru = rp[0], rv = rp[1], re = rp[2];
        const int nrows = min(64, n - k0);
        const bool row_fine = (re.x > 0.0f) && (re.y > 0.0f) && (re.z > 0.0f) && (ru.x > 0.0f) && (ru.x < INFINITY);
        u64 act[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gap = fmaxf(hx0[j] - rv.x, ru.w - hx1[j]);    // >= 0: the row box lies beside the slot's hull (x0 = ru.w, x1 = rv.x)
            const float gapz = fmaxf(hz0[j] - rv.z, rv.y - hz1[j]);   // z0 = rv.y, z1 = rv.z
            const bool skip = cullj[j] && row_fine && (((gap >= 0.0f) && (gap >= (re.x + mlx[j]) * kappa)) ||
                                                       ((gapz >= 0.0f) && (gapz >= (re.z + mlz[j]) * kappa)));
            act[j] = __ballot((lane < nrows) && !skip);
        }
        const u64 any = (act[0] | act[1]) | (act[2] | act[3]);
        unsigned wd[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned todo = (unsigned)(half ? (any >> 32) : (any & 0xffffffffull));
            while (todo) {
                const int rr = __builtin_ctz(todo);
                todo &= todo - 1u;
                const int r = half * 32 + rr;
                auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), r)); };
                Row a;
                a.vol = bc(ru.x); a.y0 = bc(ru.y); a.y1 = bc(ru.z); a.x0 = bc(ru.w); a.x1 = bc(rv.x); a.z0 = bc(rv.y); a.z1 = bc(rv.z);
                a.lx = bc(re.x); a.ly = bc(re.y); a.lz = bc(re.z); a.bad = bc(re.w);
                const unsigned bit = 1u << rr;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if ((act[j] >> r) & 1ull) {                        // wave-uniform
                        const float q = nms_overlap3d_guarded1(a, col[j], colbad[j], sane[j], thr);
                        wd[half][j] |= !(q <= thr) ? bit : 0u;
                    }
                }
            }
        }
        u64* Wk = I.W + (size_t)kb * L.NC;                             // full rows (symmetric overlap): see bitmask_boxes_kernel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (crank[j] != 0x7fffffff) Wk[crank[j]] = ((u64)wd[1][j] << 32) | wd[0][j];
        }
    }
    }
}

// (Round 3 tried this kernel with the LDS row buffer that won for the 2D boxes at large N -- one 16-wave workgroup per rank block, the
// full row of W written coalesced: at 106 VGPRs only one such workgroup fits a CU and the record gathers of the column side are exposed:
// B = 8, N = 16384 step 2.20 -> 2.33 ms, N = 8192 0.593 -> 0.616.  The scatter kernel stays.)

// ------------------------------------------------------------------------------------------------
// K3: leaders (= the boxes classical greedy NMS keeps).  The scan is inherently sequential over rank
// blocks; what must NOT be on that sequential path is global-memory latency.  Ranks are processed in
// super-blocks of kSB = 16 blocks (1024 ranks), software-pipelined inside ONE workgroup of 16 waves:
//   * removed-words of ALL blocks live in LDS (accAll[NB]);
//   * while wave 0 resolves super-block sb (registers + LDS only), waves 1..15 prefetch, for super-block
//     sb+1, the speculative triangular table  Xs[(b,b')][lane] = W[b'][rank(b,lane)]  (what each candidate
//     of block b would remove in blocks b' >= b of its own super-block): contiguous 512-B rows of W;
//   * the new leaders' words are PUSHED into the removed-words of all later blocks by gathering, per target block, exactly the
//     leaders' words (lane j = j-th leader of the super-block, from a list the resolve appends to): the 16 blocks of
//     super-block sb+1 by all waves right after the resolve ("near", one block per wave, the only part the next resolve
//     waits for), the blocks beyond by waves 1..15 WHILE wave 0 resolves sb+1 ("far").  B=8, N=4096: 28.6 -> 24.5 us.
// Resolve of one block: lane b' carries the removed-word of block b'; the 64 ranks are resolved on the
// scalar unit visiting only the leaders (s_ff1 on ~removed); each new leader ORs its table row into the
// lanes of the later blocks of the super-block.
// ------------------------------------------------------------------------------------------------
constexpr int kSB = 16;
constexpr int kSBPairs = kSB * (kSB + 1) / 2;
constexpr int kTabPer = (kSBPairs * 64 + 959) / 960;            // table entries per prefetching thread (waves 1..15)

#ifdef GNMS_TIMING   // developer instrumentation (tools/phase_ticks.py): s_memtime deltas of thread 0 (GNMS_RACC: thread 960) of image 0's workgroups
// accumulate in LDS -- round 5: a global read-modify-write per marker stalled wave 0 for a memory round trip, which the next barrier then
// waited for (every slot read ~1 us too long) -- and go to ws gx[] of image 0 once, at the end of the kernel (GNMS_TFLUSH)
__device__ __forceinline__ long long* gnms_tbuf() { __shared__ long long buf[32]; return buf; }
#define GNMS_TINIT() do { if (threadIdx.x < 32) gnms::gnms_tbuf()[threadIdx.x] = 0; __syncthreads(); } while (0)
#define GNMS_TFLUSH(ws_, L_, img_) do { __syncthreads(); if (threadIdx.x < 32 && (img_) == 0 && gnms::gnms_tbuf()[threadIdx.x]) \
        atomicAdd(reinterpret_cast<unsigned long long*>(gnms::img_ptrs(ws_, L_, 0).gx) + threadIdx.x, (unsigned long long)gnms::gnms_tbuf()[threadIdx.x]); } while (0)
#define GNMS_T0() long long t__ = (long long)__builtin_amdgcn_s_memtime()
#define GNMS_TACC(slot) do { long long n__ = (long long)__builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && b == 0) gnms::gnms_tbuf()[slot] += n__ - t__; t__ = n__; } while (0)   /* image 0's workgroup */
#define GNMS_TACC_IF(cond, slot) do { long long n__ = (long long)__builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && (cond)) gnms::gnms_tbuf()[slot] += n__ - t__; t__ = n__; } while (0)
#else
#define GNMS_TINIT() do {} while (0)
#define GNMS_TFLUSH(ws_, L_, img_) do {} while (0)
#define GNMS_TACC_IF(cond, slot) do {} while (0)
#define GNMS_T0() do {} while (0)
#define GNMS_TACC(slot) do {} while (0)
#endif

// workgroup barrier that waits for this wave's LDS operations only (not for its global loads / stores, as __syncthreads does): for
// hand-offs that live in LDS.  Beside the matrix writers the drain of a wave's stores takes microseconds.
// (LDS-only fences around the barrier builtin, not inline asm with a "memory" clobber: behind the clobber the compiler re-derived
// everything it had loaded from memory -- kernel arguments, counts[b] -- after EVERY barrier, ~1000 cycles per step of a loop of them)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ u64 uniform64(u64 v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readlane64(u64 v, int lane) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xffffffffu), lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ int tri_index(int b, int bp) { return b * kSB - (b * (b - 1)) / 2 + (bp - b); }   // b <= bp < kSB

// exclusive OR-scan across the lanes of a wave
__device__ __forceinline__ u64 wave_or_exclusive_scan(u64 v, int lane) {
    u64 up = shfl_up_u64(v, 1);                                     // lane i <- lane i-1
    if (lane == 0) up = 0ull;
    return gnms_or_scan64(up);                                      // DPP inclusive OR-scan (gnms_common.h)
}

__host__ __device__ __forceinline__ size_t leaders_lds_layout(int NB, size_t* off_acc, size_t* off_lm, size_t* off_cand, size_t* off_pair) {
    size_t o = (size_t)kSBPairs * 64 * 8;                       // Xs
    *off_acc = o; o += (size_t)((NB + 1) & ~1) * 8;             // accAll[NB]
    *off_lm = o; o += (size_t)((NB + 1 + kSB) & ~1) * 8;        // leader masks of all blocks (+ kSB of padding the sym scan may read)
    *off_cand = o; o += 2 * (size_t)kSB * 64 * 4 + 128;        // leader ranks of the last two super-blocks + their counts (sym scan: list, claims, stamps)
    *off_pair = o; o += 2 * kSBPairs * 4;                       // pair -> (b, b')
    return o;
}

// dynamic LDS of leaders_kernel / leaders_body for an image of NB rank blocks (the one definition every launch site uses)
__host__ __device__ __forceinline__ size_t leaders_lds_size(int NB) {
    size_t a, l, c, p;
    return leaders_lds_layout(NB, &a, &l, &c, &p);
}

// epilogue of the leader scan, off the sequential path: leader lists, per-block words, running counts (lmask[nb] in LDS is final;
// `scratch` = >= (8 + nb) ints of LDS nobody else uses any more)
__device__ __forceinline__ void leaders_epilogue(const ImgPtrs& I, const u64* lmask, int nb, int* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // exclusive prefix of popcounts over blocks: thread i < nb owns block i (nb <= 256)
    int cnt = (tid < nb) ? __builtin_popcountll(lmask[tid]) : 0;
    const int inc = (int)gnms_add_scan32((unsigned)cnt);          // DPP prefix sum
    int* wsum = scratch;
    if (lane == 63 && wave < 4) wsum[wave] = inc;
    __syncthreads();
    int basew = 0;
    for (int w = 0; w < 4; ++w) if (w < wave) basew += wsum[w];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int* pfx = wsum + 8;                                          // [nb] exclusive prefix
    if (tid < nb) {
        pfx[tid] = basew + inc - cnt;
        I.leadw[tid] = lmask[tid];
        I.leadpfx[tid + 1] = basew + inc;
    }
    __syncthreads();
    for (int kb = wave; kb < nb; kb += 16) {
        const u64 mine = lmask[kb];
        if ((mine >> lane) & 1ull) {
            const int slot = pfx[kb] + __builtin_popcountll(mine & ((1ull << lane) - 1ull));
            const int k = (kb << 6) + lane;
            I.leadc[slot] = I.order[k];
            I.leadr[slot] = k;
        }
    }
    if (tid == 0) { I.misc[0] = total; I.leadpfx[0] = 0; }
}

// ------------------------------------------------------------------------------------------------
// K3 for a SYMMETRIC thresholded matrix whose rows W holds in full (from-boxes / from-records bit matrices; a matrix
// wsym_check_kernel passed) -- round 4: ONE WORKGROUP PER SUPER-BLOCK (16 blocks = 1024 ranks), the workgroups of an image on
// different CUs, handing the leader masks down the chain.
//
// What a super-block's leaders depend on is small -- the leader masks of the super-blocks before it, 128 bytes each -- and what
// turns those masks into "removed by an earlier leader" is large but has nothing sequential in it: rank (T, lane) overlaps a
// leader of source block bb iff  W[bb][rank] & mask[bb] != 0  (the matrix is symmetric), one coalesced 512-byte row segment per
// (source block, target block), 1 MiB per image at N = 4096, 16 MiB at 16384.  One workgroup per image pulled all of that through
// one CU's memory pipeline (~24 bytes per clock measured: 60k of the scan's 120k ticks on uniform boxes) between resolves that 15 of
// its 16 waves sat out.  Here workgroup j owns super-block j: it requests the row segments of source super-block s for its own 16
// blocks BEFORE the masks of s exist (the addresses do not depend on them), takes the masks when the workgroup of s publishes them,
// ANDs -- sources in ascending order, so the first hit of a rank is its claimer (rem[], what K4 used to compute) -- then resolves its
// own super-block, publishes, and stores rem[] for its ranks.  The chain is nsb hand-offs long; everything heavy runs beside it.
//
// HAND-OFF.  A mask travels as two 8-byte granules {epoch, 32 mask bits} written by single write-through stores and polled with
// agent-scope loads: a granule is valid iff its tag is the workspace's call counter (misc[8], advanced by the sort kernels of every
// call -- so stale granules of earlier calls, or of earlier replays of a captured graph, never match; the same kernels also zero the
// image's granules, so that whatever a recycled allocation held where they now lie cannot carry the tag by accident).
// No fence anywhere: beside the matrix writers one agent-scope release costs more than the whole scan (nms_layer.hip).  rem[] goes
// out through write-through stores as well and is read back (groups_body / attribute_*) with agent-scope loads; the workgroup of the
// LAST super-block waits for the other workgroups' "rem stored" granules and carries on with K4..K6 of the image.
// A workgroup only ever waits for workgroups with a LOWER block index (dispatched before it).
//
// RESOLVE (inside a super-block).  The leaders are the unique solution of
//     L[r] = !ext[r]  &&  no q < r (same super-block) with L[q] and overlap(q, r).
// Wave tb owns block tb and iterates: from the current masks of the blocks before it (LDS, read at immediate offsets) and its table
// words t[bb] = W[kb0 + bb][rank (tb, lane)] (registers) it recomputes its own mask (the in-block fixed point of round 2), all 16
// waves at once, one LDS barrier per step, until a step changes no mask; a wave whose earlier blocks did not change since it last
// looked skips the step.  Block tb is exact after step tb + 1 at the latest, NMS inputs settle in 3-5 steps.  (What a step costs is
// LDS and VALU issue, 16 waves on 4 SIMDs: ~250 cycles for the barrier, ~500 for the 16 x 15 mask reads, ~1000 for the AND/ORs --
// tools/../build/mb/bar.hip -- so the parallel resolve is no faster than the scalar-unit resolve of round 2 per super-block; what it
// buys is that no wave idles and nothing else has to be overlapped with it.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 gran_load(const u64* p) {
    return (u64)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_store(u64* p, u64 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// rem[] as the leader scan leaves it: agent-scope accesses (see above; plain ones would do between kernels, these do everywhere)
__device__ __forceinline__ int rem_load(const int* rem, int k) { return __hip_atomic_load(rem + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rem_store(int* rem, int k, int v) { __hip_atomic_store(rem + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The super-blocks [j0, j1) of image `b`, one after the other (j1 <= nsb of the image; workgroup `me` of the image's `nwg` chain
// workgroups, which own contiguous ascending ranges -- so a workgroup only ever waits for workgroups with a lower index).  Returns true
// in the workgroup that may go on with K4..K6 of the image: the one whose range ends with the image's last super-block, after every
// rem[] entry of the image is visible to it.
//
// STAGE (round 5, the fast tail; STAGE = the overlap source, kFromMatrix / kFromBoxes / kFromRecords, or kNoStage): masked groups, hard sort.
// The workgroup that resolved a super-block knows, per rank k, the leader lr = rem[k] that took it -- and with it everything K5 computes for
// k as long as two things hold for the IMAGE: every leader is a member of its own group (overlap(L, L) > thr: any box of positive finite
// area), so head == leader, and no group is longer than the cap, so every member is kept.  It therefore evaluates
//     member = overlap(k, lr) > thr;  plead = prune(overlap);  pre = s_k - plead * s_lr  (s_k for a leader);  r2 = clamp(pre)
// right behind its resolve, for its own 1024 ranks, beside the chain (the masks are out by then), and stores head / plead / pre / r2
// write-through in front of its "rem stored" granule; a leader outside its own group raises the granule's `complex` bit.  The image's last
// workgroup then only has to check the two conditions (fast_final_body) before K6 -- no sort of the groups on the path to the
// probabilities -- and the groups' CSR the backward reads is built beside K6 by one more workgroup (csr_build_body).
// Returns 0 in the workgroups that are done, 1 in the image's last one (2: STAGE and some leader is outside its own group).
constexpr int kNoStage = -1;
constexpr int kFastGroupMax = 256;   // longest group the fast tail handles (default cap: 101)
constexpr int kGranVerdict = 30;     // gran[16][30]: the last workgroup's verdict for csr_build_body (payload 1 = fast tail, 2 = K5 ran, nothing to do)

// FUSED (round 6, one_launch_kernel; a single super-block: j0 = 0, j1 = 1): sort, bit table and this chain are workgroups of ONE launch.
// The sort's outputs are read with agent-scope loads (the caller has waited for the sort's flags); the super-block's table is not
// gathered from W but copied from the image the table workgroups leave where W lies (one_launch_bits_*: already in this layout, word
// (source block bb <= target block tb, target lane) = the bits M[target][source] -- the reference's own orientation, :250, so no
// symmetry is assumed and none is checked), behind a wait for their `nflags` flags; `tag` = the launch's call counter value.
template <int STAGE = kNoStage, bool FUSED = false>
__device__ __forceinline__ int leaders_sb_body(int N, const int* __restrict__ counts, char* ws, gnms_ws_layout L, const int b, const int j0,
                                               const int j1, const int me, const float* __restrict__ stage_src = nullptr, long stage_ld = 0,
                                               const float stage_thr = 0.0f, const float stage_temp = 0.0f, const int stage_prune = 0,
                                               const int Ppow2 = 0, const unsigned tag = 0u, const int nflags = 0, const size_t lds_side = 0,
                                               const u64* __restrict__ ext0 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    size_t oa, ol, oc, op;
    leaders_lds_layout(L.NB, &oa, &ol, &oc, &op);
    u64* Xs = reinterpret_cast<u64*>(smem);                      // [kSBPairs][64] table of the own super-block
    u64* lmask = reinterpret_cast<u64*>(smem + ol);              // [NB + kSB] leader masks: the earlier super-blocks' as they arrive, then the own
    int* pair_b = reinterpret_cast<int*>(smem + op);             // [kSBPairs]
    int* pair_bp = pair_b + kSBPairs;
    int* stamp = reinterpret_cast<int*>(smem + oc);              // [1] the last resolve step that changed a mask
    int* bstamp = stamp + 4;                                     // [kSB] per block of the super-block: the last step that changed its mask
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    // (the wave index through readfirstlane: as tid >> 6 it is a VECTOR value to the compiler, and every `bb < tb` below became a 64-bit
    // lane mask in an SGPR pair -- 186 spilled SGPRs and a v_readlane / s_nop pair around every use)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = (n + 63) >> 6;
    const int nsb = max(1, (nb + kSB - 1) / kSB);                  // (an EMPTY image still has one -- empty -- super-block: leaders_chain's count; its
                                                                   // workgroup must reach K4..K6, the only writers of the image's outputs)
    const u64 below = (1ull << lane) - 1ull;
    const bool last_wg = j1 == nsb;                                // this workgroup ends with the image's last super-block
    int have = 0;                                                  // super-blocks [0, have) have their masks in lmask (workgroup-uniform)
    int complex_img = 0;                                           // STAGE: a leader of this workgroup's ranks is outside its own group
    const u64 epoch = FUSED ? ((u64)tag << 32) : ((u64)(unsigned)I.misc[8] << 32);
    GNMS_T0();
    // table layout: pair (source block bb <= target block tb) at tb (tb + 1) / 2 + bb -- a target's words are consecutive, so the wave
    // that owns it reads them at immediate offsets from one base (the general scan's layout needs the triangular index per word:
    // ~12 scalar instructions each, and sixteen waves share the CU's one scalar unit)
    if (tid < kSB) for (int bb = 0; bb <= tid; ++bb) { pair_b[tid * (tid + 1) / 2 + bb] = bb; pair_bp[tid * (tid + 1) / 2 + bb] = tid; }
    for (int i = tid; i < nb + kSB; i += 1024) lmask[i] = 0ull;
    __syncthreads();
    for (int j = j0; j < j1; ++j) {
    const int kb0 = j * kSB;
    const int nblk = min(kSB, nb - kb0);
    const bool last_sb = j == nsb - 1;
    // STAGE: what the rank needs of itself (score, input index, box) is requested before anything is waited for
    float st_sk = 0.0f;
    int st_ck = 0;
    float4 st_bk = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (STAGE != kNoStage) {
        const int kk = ((kb0 + (wave < nblk ? wave : 0)) << 6) + lane;
        if constexpr (FUSED) {
            // the whole image's order / scores (/ boxes) by rank into LDS, beside everything else (smem + lds_side): the rank's own values and,
            // behind the resolve, its leader's come from there -- one gather level (the overlap entry) instead of two
            int* ordA = reinterpret_cast<int*>(smem + lds_side);
            float* sscA = reinterpret_cast<float*>(ordA + 1024);
            float4* rbxA = reinterpret_cast<float4*>(smem + lds_side + 8192);
            if (tid < n) {
                ordA[tid] = coh_load(I.order + tid);
                sscA[tid] = coh_load(I.sscore + tid);
                if (STAGE == kFromBoxes) rbxA[tid] = coh_load_f4(I.rbox + tid);
            }
        } else if (wave < nblk && kk < n) {
            st_sk = I.sscore[kk];
            st_ck = I.order[kk];
            if (STAGE == kFromBoxes) st_bk = I.rbox[kk];
        }
    }
    if (tid == 0) *stamp = 0;
    if (tid < kSB) bstamp[tid] = -1;
    {   // the own super-block's table: no dependence on anybody (in flight while the first masks are waited for)
        constexpr int kTabAll = (kSBPairs * 64 + 1023) / 1024;
        u64 tw[kTabAll];
        if constexpr (FUSED) {                                     // the table workgroups' flags, then their image of the table as it stands
            if (wave == 0) {
                for (int f0 = 0; f0 < nflags; f0 += 64) {         // (<= 13 x 32 flags: gran[3 .. 15])
                    const int f = f0 + lane < nflags ? f0 + lane : f0;
                    const u64 want = strong_gran(tag, kSlotBits + (unsigned)f);
                    const u64* g = I.gran + (size_t)3 * 32 + f;
                    while (__ballot(gran_load(g) != want) != 0ull) __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if constexpr (STAGE != kNoStage) {
                const int kk = ((kb0 + (wave < nblk ? wave : 0)) << 6) + lane;
                if (wave < nblk && kk < n) {
                    st_sk = reinterpret_cast<const float*>(smem + lds_side + 4096)[kk];
                    st_ck = reinterpret_cast<const int*>(smem + lds_side)[kk];
                    if (STAGE == kFromBoxes) st_bk = reinterpret_cast<const float4*>(smem + lds_side + 8192)[kk];
                }
            }
            const int npw = (nb * (nb + 1) / 2) * 64;              // the image's pairs (target block < nb): what the table workgroups wrote
#pragma unroll
            for (int u = 0; u < kTabAll; ++u) {
                const int e = tid + u * 1024;
                tw[u] = (e < npw) ? coh_load(I.W + e) : 0ull;
            }
        } else {
#pragma unroll
            for (int u = 0; u < kTabAll; ++u) {
                const int e = tid + u * 1024;
                tw[u] = 0ull;
                if (e < kSBPairs * 64) {
                    const int pr = e >> 6;
                    const int bb = pair_b[pr], bp = pair_bp[pr];
                    const int k = (kb0 + bp) * 64 + (e & 63);      // row block = the SOURCE block bb, column = target rank (bp, lane)
                    if (bp < nblk && k < n) tw[u] = I.W[(size_t)(kb0 + bb) * L.NC + k];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kTabAll; ++u) {
            const int e = tid + u * 1024;
            if (e < kSBPairs * 64) Xs[e] = tw[u];
        }
    }
    GNMS_TACC_IF(b == 0 && last_sb, 0);
    const int tb = wave;                                           // this wave's block of the super-block
    const bool live = tb < nblk;
    const int kT = ((kb0 + (live ? tb : 0)) << 6) + lane;          // this lane's rank
    lds_barrier();                                                 // the table is in LDS (and the previous super-block of this workgroup is done with it and the stamps)
    // ---- resolve of the own super-block: every wave its block, steps until no mask changes ----
#ifdef GNMS_TIMING
    long long r__ = (long long)__builtin_amdgcn_s_memtime();
#define GNMS_RACC(slot) do { long long n__ = (long long)__builtin_amdgcn_s_memtime(); if (threadIdx.x == 960 && b == 0 && last_sb) gnms_tbuf()[slot] += n__ - r__; r__ = n__; } while (0)
#else
#define GNMS_RACC(slot) do {} while (0)
#endif
    const int tbc = live ? tb : 0;
    const int k0 = (kb0 + tbc) << 6;
    u64 tailmask = 0ull;                                           // past the image's last rank: never leaders
    if (live) { const int nrows = min(64, n - k0); if (nrows < 64) tailmask = ~((1ull << nrows) - 1ull); }
    const u64* myX = Xs + (size_t)(tbc * (tbc + 1) / 2) * 64 + lane;   // this wave's table words: source block bb at myX[bb * 64]
    const u64* myL = lmask + kb0;                                  // (lmask is padded by kSB entries: no clamp)
    const u64 cs = live ? (myX[tbc * 64] & below) : 0ull;          // earlier ranks of the block that overlap rank k0 + lane
    unsigned tlo[kSB - 1], thi[kSB - 1];
#pragma unroll
    for (int bb = 0; bb < kSB - 1; ++bb) {                          // (entries past tb belong to other targets: masked)
        const u64 x = myX[bb * 64];
        tlo[bb] = (bb < tb) ? (unsigned)(x & 0xffffffffu) : 0u;
        thi[bb] = (bb < tb) ? (unsigned)(x >> 32) : 0u;
    }
    u64 mine = 0ull;                                               // this block's leader mask as last published (lmask[kb0 + tb])
    GNMS_RACC(21);
    // (Round 5 ran this fixed point SPECULATIVELY while the workgroup waits -- cold with no external removals, then warm after every source
    // ANDed in -- so that only a confirming step or two would be left behind the last source's masks.  Measured: no gain; on NMS inputs every
    // source removes ranks in all 16 blocks, the repair cascades through the blocks like a cold resolve (17 steps in all where there were 4,
    // the final call as long as before, the chain 10 k ticks longer).  One call, behind the last source.)
    u64 cur_prev = ~0ull;
    bool first = true;
    int looked = 0;                                                // the step of this wave's last look at the earlier masks
    int step = 1;
    auto resolve = [&](const u64 ext) {
        const u64 fixed = ext | tailmask;                          // never leaders: taken by earlier super-blocks, or past the image's last rank
        bool force = true;
        for (;;) {
            if (live) {                                            // (wave-uniform)
                // did a block before mine change since I last looked?  (lane bb holds block bb's stamp)
                const bool dirty = force || __ballot(lane < tb && bstamp[lane < kSB ? lane : 0] >= looked) != 0ull;
                if (dirty) {
                    looked = step;
                    unsigned vlo = 0u, vhi = 0u;
                    if (!first) {                                  // (the very first look finds every mask of the super-block still zero)
#pragma unroll
                        for (int bb = 0; bb < kSB - 1; ++bb) {      // (one batch of reads: behind a branch per group of blocks they serialise)
                            const u64 l = myL[bb];
                            vlo |= tlo[bb] & (unsigned)(l & 0xffffffffu);
                            vhi |= thi[bb] & (unsigned)(l >> 32);
                        }
                    }
                    const u64 cur = fixed | __ballot((vlo | vhi) != 0u);
                    GNMS_RACC(22);
                    if (first || cur != cur_prev) {                // (the same `cur` gives the same leaders)
                        cur_prev = cur;
                        u64 leaders = 0ull;
                        if (~cur != 0ull) {
                            const bool cand = ((cur >> lane) & 1ull) == 0ull;
                            leaders = ~cur;
                            for (;;) {                             // in-block fixed point: positions < t are final after t rounds
                                const u64 nl = __ballot(cand && (cs & leaders) == 0ull);
                                if (nl == leaders) break;
                                leaders = nl;
                            }
                        }
                        if (leaders != mine) {
                            mine = leaders;
                            if (lane == 0) { lmask[kb0 + tb] = leaders; bstamp[tb] = step; *stamp = step; }
                        }
                    }
                    first = false;
                    GNMS_RACC(23);
                }
            }
            force = false;
            lds_barrier();
            GNMS_RACC(24);
            const int last = *stamp;
            ++step;
            if (step - last > 1) break;                            // a whole step without a change (a faster wave may have stamped a later step: still <=)
        }
    };
    // ---- the earlier super-blocks, in ascending order: row segments requested, masks awaited, ANDed ----
    int cl = -1;                                                   // rank of the first leader (of an earlier super-block) that overlaps it
    for (int s = 0; s < j; ++s) {
        const int s0 = s * kSB;
        u64 wv[kSB];
#pragma unroll
        for (int bb = 0; bb < kSB; ++bb) wv[bb] = (live && kT < n) ? I.W[(size_t)(s0 + bb) * L.NC + kT] : 0ull;
        if (wave == 0 && s >= have) {                              // the masks of super-block s (another workgroup's, not seen yet): 32 granules, lane g polls granule g
            const u64* g = I.gran + (size_t)s * 32 + (lane & 31);
            u64 v = gran_load(g);
            while (__ballot((v & 0xffffffff00000000ull) != epoch) != 0ull) { __builtin_amdgcn_s_sleep(2); v = gran_load(g); }
            reinterpret_cast<unsigned*>(lmask + s0)[lane & 31] = (unsigned)(v & 0xffffffffu);   // (lanes 32..63 write the same words again)
        }
        lds_barrier();
        int c = -1;
#pragma unroll
        for (int bb = 0; bb < kSB; ++bb) {
            const u64 m = wv[bb] & lmask[s0 + bb];
            if (c < 0 && m != 0ull) c = ((s0 + bb) << 6) + __builtin_ctzll(m);
        }
        if (cl < 0) cl = c;
    }
    if (j + 1 > have) have = j + 1;                                // (every super-block up to j is in lmask from here on: polled above, or this workgroup's own)
    // (ext0, classical NMS in chunks -- classic_nms.hip: word kb of it = ranks of block kb that something OUTSIDE this scan has removed already)
    const u64 ext = __ballot(cl >= 0) | ((ext0 != nullptr && live) ? ext0[kb0 + tbc] : 0ull);   // removed-word of the wave's block (earlier super-blocks' leaders)
    GNMS_TACC_IF(b == 0 && last_sb, 1);
    if constexpr (FUSED) {
        // A single super-block with nobody to hand masks to: ONE wave walks the blocks in order -- lane = rank of the block, the table words of
        // block t2 against the masks of the blocks before it (wave-uniform: scalar registers), then the in-block fixed point -- no barrier and no
        // repeated steps: ~150-300 cycles per block where the sixteen-wave fixed point above takes 4-7 steps of ~1200 (N = 500: 4.3 -> ~1 us)
        if (wave == 0) {
            u64 m[kSB];
#pragma unroll
            for (int t2 = 0; t2 < kSB; ++t2) {
                m[t2] = 0ull;
                if (t2 < nblk) {                                   // (wave-uniform)
                    const u64* X = Xs + (size_t)(t2 * (t2 + 1) / 2) * 64 + lane;
                    u64 hit = 0ull;
#pragma unroll
                    for (int bb = 0; bb < t2; ++bb) hit |= X[bb * 64] & m[bb];
                    const u64 csq = X[t2 * 64] & below;
                    const int nr = min(64, n - ((kb0 + t2) << 6));
                    const u64 cur = __ballot(hit != 0ull) | (nr < 64 ? ~((1ull << nr) - 1ull) : 0ull);
                    u64 leaders = ~cur;
                    if (leaders != 0ull) {
                        const bool cand = ((cur >> lane) & 1ull) == 0ull;
                        for (;;) {                                 // in-block fixed point: positions < t are final after t rounds
                            const u64 nl = __ballot(cand && (csq & leaders) == 0ull);
                            if (nl == leaders) break;
                            leaders = nl;
                        }
                    }
                    m[t2] = leaders;
                    if (lane == 0) lmask[kb0 + t2] = leaders;
                }
            }
        }
        lds_barrier();
        mine = live ? lmask[kb0 + tb] : 0ull;
    } else {
        resolve(ext);
    }
#ifdef GNMS_TIMING
    if (threadIdx.x == 0 && b == 0 && last_sb) gnms_tbuf()[20] += step;
#endif
    // ---- publish the masks (the chain's critical path ends here), then rem[] ----
    if (!last_sb && wave == 0 && lane < 32) {
        const unsigned half = reinterpret_cast<const unsigned*>(lmask + kb0)[lane];
        gran_store(I.gran + (size_t)j * 32 + lane, epoch | half);
    }
    GNMS_TACC_IF(b == 0 && last_sb, 2);
    // rem[] of the wave's block: itself for a leader, else the first claimer -- an earlier super-block's (cl), else the first earlier
    // block of this super-block with a leader in the table word, else the own block
    {
        const int k = k0 + lane;
        int c = -1;
#pragma unroll
        for (int bb = 0; bb < kSB - 1; ++bb) {
            const u64 l = myL[bb];
            const unsigned mlo = tlo[bb] & (unsigned)(l & 0xffffffffu), mhi = thi[bb] & (unsigned)(l >> 32);
            if (c < 0 && (mlo | mhi) != 0u) c = ((kb0 + bb) << 6) + (mlo ? __builtin_ctz(mlo) : 32 + __builtin_ctz(mhi));
        }
        if (c < 0) { const u64 m = cs & mine; c = (m != 0ull) ? k0 + __builtin_ctzll(m) : k; }
        int r = k;
        if (((mine >> lane) & 1ull) == 0ull) r = ((ext >> lane) & 1ull) ? cl : c;
        if constexpr (STAGE != kNoStage) {
            // the last workgroup's last super-block stays in LDS for fast_final_body (the table is dead: every wave has passed a barrier behind
            // its last read of it): r2, head and input index by rank
            const bool park = last_wg && j == j1 - 1;
            float* r2L = reinterpret_cast<float*>(smem);                             // (finalize_fast_body's layout)
            int* hdL = reinterpret_cast<int*>(smem + (size_t)Ppow2 * 4);
            int* ordL = reinterpret_cast<int*>(smem + (size_t)Ppow2 * 8);
            if (live && k < n) {
                const int lr = r;
                float sl = st_sk, ov;
                if (STAGE == kFromBoxes) {
                    float4 bl = st_bk;
                    if (lr != k) {
                        if constexpr (FUSED) { bl = reinterpret_cast<const float4*>(smem + lds_side + 8192)[lr]; sl = reinterpret_cast<const float*>(smem + lds_side + 4096)[lr]; }
                        else { bl = I.rbox[lr]; sl = I.sscore[lr]; }
                    }
                    ov = pair_iou(st_bk, bl);
                } else {
                    int cb = st_ck;
                    if (lr != k) {
                        if constexpr (FUSED) { cb = reinterpret_cast<const int*>(smem + lds_side)[lr]; sl = reinterpret_cast<const float*>(smem + lds_side + 4096)[lr]; }
                        else { cb = I.order[lr]; sl = I.sscore[lr]; }
                    }
                    ov = overlap_at<STAGE>(overlap_src<STAGE>(stage_src, I, b, N, stage_ld), stage_ld, st_ck, cb, stage_thr);
                }
                float pre = 0.0f, pl = 0.0f;
                int hd = -1;
                if (ov > stage_thr) {                              // strict > (:249)
                    hd = lr;
                    if (lr == k) pre = st_sk;
                    else { pl = gnms_prune(ov, stage_thr, stage_temp, stage_prune); pre = st_sk - pl * sl; }
                } else if (lr == k) complex_img = 1;               // a leader outside its own group (NaN / <= thr diagonal): K5 proper decides the heads
                const float r2 = pre < 0.0f ? 0.0f : (pre > 1.0f ? 1.0f : pre);      // torch.clamp keeps NaN
                if (park) { r2L[k] = r2; hdL[k] = hd; ordL[k] = st_ck; }
                rem_store(I.rem, k, r);
                rem_store(I.head, k, hd);
                __hip_atomic_store(I.plead + k, pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(I.pre + k, pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(I.r2 + k, r2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (live && k < n) rem_store(I.rem, k, r);
        }
    }
    GNMS_TACC_IF(b == 0 && last_sb, 4);
    }   // super-blocks of this workgroup
    if (!last_wg) {
        // every wave waits for the acknowledgement of its OWN write-through stores (s_waitcnt vmcnt(0): the barrier alone is an
        // s_barrier, which orders nothing in memory -- ADVICE r4), then the barrier, then "rem stored" goes out
        __builtin_amdgcn_s_waitcnt(0x0f70);                        // vmcnt(0) (gfx9 encoding: vmcnt = bits 3:0 + 15:14; expcnt / lgkmcnt left at their maxima)
        const int cx = (STAGE != kNoStage) ? __syncthreads_or(complex_img) : (__syncthreads(), 0);
        if (tid == 0) gran_store(I.gran + (size_t)16 * 32 + me, epoch | 1ull | (cx ? 2ull : 0ull));
        return 0;
    }
    // ---- the image's last super-block: the other workgroups' rem[] must be there, then the per-image epilogue (all masks are in LDS) ----
    if (wave == 0 && me > 0) {
        const u64* g = I.gran + (size_t)16 * 32 + (lane < me ? lane : 0);
        u64 v = gran_load(g);
        while (__ballot((v & 0xffffffff00000000ull) != epoch) != 0ull) { __builtin_amdgcn_s_sleep(2); v = gran_load(g); }
        if (STAGE != kNoStage && __ballot((v & 2ull) != 0ull) != 0ull) complex_img = 1;
    }
    GNMS_TACC_IF(b == 0 && last_wg, 3);
    if constexpr (STAGE != kNoStage) {
        return __syncthreads_or(complex_img) ? 2 : 1;              // (leaders_epilogue: only the slow path needs the lists -- fast_final_body)
    } else {
        __syncthreads();
        leaders_epilogue(I, lmask, nb, reinterpret_cast<int*>(Xs));
        GNMS_TACC_IF(b == 0 && last_wg, 4);
        return 1;
    }
}

// The four per-image stages K3..K6 are written as device functions (`*_body`, 1024 threads, image index `b`) so that they
// run either as kernels of their own (thin wrappers below) or back to back inside ONE launch (tail_kernel).
// leaders_body: the GENERAL scan (matrix in, possibly asymmetric; classical NMS), one workgroup per image.  Table entry
// (b, b')[lane] = W[b'][rank(b, lane)], what candidate (b, lane) would remove in block b', pushed leader by leader; wave 0 resolves on
// the scalar unit (~130 cycles per leader) while waves 1..15 prefetch and push.  Symmetric matrices take leaders_sb_body above.
__device__ __forceinline__ void leaders_body(int N, const int* __restrict__ counts, char* ws, gnms_ws_layout L, const int b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    size_t oa, ol, oc, op;
    leaders_lds_layout(L.NB, &oa, &ol, &oc, &op);
    u64* Xs = reinterpret_cast<u64*>(smem);                      // [kSBPairs][64]
    u64* accAll = reinterpret_cast<u64*>(smem + oa);             // [NB]
    u64* lmask = reinterpret_cast<u64*>(smem + ol);              // [NB]
    int* pair_b = reinterpret_cast<int*>(smem + op);             // [kSBPairs]
    int* pair_bp = pair_b + kSBPairs;
    int* llist = reinterpret_cast<int*>(smem + oc);              // [2][kSB * 64] ranks of the leaders of super-block sb (buffer sb & 1)
    int* lcount = llist + 2 * kSB * 64;                          // [2]
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = (n + 63) >> 6;
    const int nsb = (nb + kSB - 1) / kSB;
    GNMS_T0();
    if (tid < kSB) for (int bp = tid; bp < kSB; ++bp) { pair_b[tri_index(tid, bp)] = tid; pair_bp[tri_index(tid, bp)] = bp; }
    for (int i = tid; i < nb; i += 1024) { accAll[i] = 0ull; lmask[i] = 0ull; }
    __syncthreads();

    // table prefetch of super-block `sb` into registers, by the threads [first, 1024)
    u64 tw[kTabPer];
    auto table_load = [&](int sb, int first, int nthr) {
        const int kb0 = sb * kSB;
        const int nblk = min(kSB, nb - kb0);
#pragma unroll
        for (int u = 0; u < kTabPer; ++u) {
            const int e = (tid - first) + u * nthr;
            tw[u] = 0ull;
            if (tid >= first && e < kSBPairs * 64) {
                const int pr = e >> 6;
                const int bb = pair_b[pr], bp = pair_bp[pr];
                if (bp < nblk) {
                    // row block = the target block bp, column = candidate rank (bb, lane): contiguous 512 B per (b, b') row
                    const int k = (kb0 + bb) * 64 + (e & 63);
                    if (k < n) tw[u] = I.W[(size_t)(kb0 + bp) * L.NC + k];
                }
            }
        }
    };
    auto table_store = [&](int first, int nthr) {
#pragma unroll
        for (int u = 0; u < kTabPer; ++u) {
            const int e = (tid - first) + u * nthr;
            if (tid >= first && e < kSBPairs * 64) Xs[e] = tw[u];
        }
    };
    // prologue: table of super-block 0 (waves 1..15 so the register footprint is the same as in the loop)
    table_load(0, 64, 960);
    table_store(64, 960);
    __syncthreads();
    GNMS_TACC(0);

    // push: OR the words of super-block `src`'s leaders into the removed-words of the blocks [first, last); the calling waves are
    // numbered w of nw and take every nw-th block, four blocks at a time.  Lane j gathers the word of the j-th leader (the resolve
    // appends every leader to a list as it finds it) -- exactly the words that matter, and all of a wave's blocks in ONE memory
    // round trip.
    auto push = [&](int src, int first, int last, int w, int nw) {
        const int* list = llist + (src & 1) * (kSB * 64);
        const int nl = lcount[src & 1];
        for (int base = first + w; base < last; base += 4 * nw) {
            u64 a[4] = {0ull, 0ull, 0ull, 0ull};
            for (int j0 = 0; j0 < nl; j0 += 64) {
                const int j = j0 + lane;
                const int lr = (j < nl) ? list[j] : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kbp = base + u * nw;
                    if (lr >= 0 && kbp < last) a[u] |= I.W[(size_t)kbp * L.NC + lr];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kbp = base + u * nw;
                if (kbp < last) {                                  // wave-uniform
                    const u64 acc = gnms_wave_or(a[u]);
                    if (lane == 0 && acc != 0ull) atomicOr(reinterpret_cast<unsigned long long*>(&accAll[kbp]), (unsigned long long)acc);
                }
            }
        }
    };

    // Per super-block sb:  wave 0 resolves it (registers and LDS only) WHILE waves 1..15 prefetch the table of sb+1 and push the
    // leaders of sb-1 into the blocks from super-block sb+1 on ("far" push: nothing the running resolve reads);  then all 16 waves
    // push the new leaders of sb into the 16 blocks of super-block sb+1 ("near" push, one block per wave), the only part of the
    // push the next resolve has to wait for.
    for (int sb = 0; sb < nsb; ++sb) {
        const int kb0 = sb * kSB;
        const int nblk = min(kSB, nb - kb0);
        if (wave != 0) {
            if (sb + 1 < nsb) table_load(sb + 1, 64, 960);
            if (sb >= 1) push(sb - 1, (sb + 1) * kSB, nb, wave - 1, 15);
        } else {
            // ---- sequential resolve of this super-block: registers and LDS only ----
            u64 myacc = (lane < nblk) ? accAll[kb0 + lane] : 0ull;     // lane b' = removed-word of block kb0+b'
            u64 mylead = 0;                                            // lane b' = leader mask of block kb0+b'
            int* list = llist + (sb & 1) * (kSB * 64);
            int filled = 0;                                            // leaders of this super-block so far (wave-uniform)
            for (int bb = 0; bb < nblk; ++bb) {
                const int k0 = (kb0 + bb) << 6;
                const int nrows = min(64, n - k0);
                u64 cur = readlane64(myacc, bb);
                if (nrows < 64) cur |= ~((1ull << nrows) - 1ull);     // ranks >= n never lead
                if (~cur == 0ull) continue;                            // everything in this block is already removed
                const u64 d = Xs[tri_index(bb, bb) * 64 + lane];       // what rank k0+lane removes inside its own block
                // lane b' (> bb) also ORs the table row of every new leader into its removed-word; the LDS read of
                // leader i is consumed one trip later, so its latency hides behind the scalar chain of leader i+1
                const bool tgt = lane > bb && lane < nblk;
                const u64* row = Xs + (size_t)tri_index(bb, tgt ? lane : bb) * 64;
                u64 leaders = 0, pend = 0;
                do {                                                   // scalar loop: one trip per leader
                    const int p = __builtin_ctzll(~cur);
                    leaders |= 1ull << p;
                    const u64 nxt = row[p];
                    cur |= readlane64(d, p) | (1ull << p);             // a leader always leaves `remaining`
                    myacc |= tgt ? pend : 0ull;
                    pend = nxt;
                } while (~cur != 0ull);
                myacc |= tgt ? pend : 0ull;
                if (lane == bb) mylead = leaders;
                // the block's leaders join the list the pushes gather by (`leaders` is wave-uniform: one masked LDS store)
                if ((leaders >> lane) & 1ull) list[filled + __builtin_popcountll(leaders & ((1ull << lane) - 1ull))] = k0 + lane;
                filled += __builtin_popcountll(leaders);
            }
            if (lane < nblk) lmask[kb0 + lane] = mylead;
            if (lane == 0) lcount[sb & 1] = filled;
        }
        GNMS_TACC(1);
        __syncthreads();                                               // (A) wave 0 is done with Xs; the far pushes have landed
        GNMS_TACC(2);
        push(sb, kb0 + nblk, min(kb0 + nblk + kSB, nb), wave, 16);
        if (sb + 1 < nsb) table_store(64, 960);                        // land the prefetched table for the next super-block
        GNMS_TACC(3);
        __syncthreads();                                               // (B)
        GNMS_TACC(4);
    }
    leaders_epilogue(I, lmask, nb, reinterpret_cast<int*>(Xs));
}

// One CHAIN workgroup of a launch that runs the leader scan of B images with `spw` workgroups per image (spw = ceil(NB / 16) when the
// launch may meet symmetric images -- leaders_chain_wgs --, else 1); chain index c = (workgroup of the image) * B + image, so that the
// workgroups of the first super-blocks, which wait for nobody, are dispatched first.  sym_arg: 0 general, 1 symmetric (W holds full
// rows), 2 as wsym_check_kernel found this image's matrix.  Returns true, with *image set, in the ONE workgroup per image that goes
// on with K4..K6.
// workgroups per image: one per super-block, at most `cap` (0: no cap).  The launches that run beside a matrix write of their own
// (large images) cap it so that the chain leaves the writers their CUs.
__host__ __device__ inline int leaders_chain_wgs(int N, int sym_arg, int cap = 0) {
    const int nsb = max(1, ((N + 63) / 64 + kSB - 1) / kSB);
    const int w = sym_arg ? nsb : 1;
    return (cap > 0 && w > cap) ? cap : w;
}

// Returns 0 in the workgroups that are done; in the ONE workgroup per image that goes on with K4..K6: 1 (symmetric scan; with STAGE the
// fast tail's values are stored, see leaders_sb_body), 2 (the same, but some leader is outside its own group), 3 (general scan).
template <int STAGE = kNoStage>
__device__ __forceinline__ int leaders_chain(int N, const int* __restrict__ counts, char* ws, gnms_ws_layout L, const int B, const int spw,
                                             const int c, const int sym_arg, int* image, const float* __restrict__ stage_src = nullptr,
                                             long stage_ld = 0, const float stage_thr = 0.0f, const float stage_temp = 0.0f,
                                             const int stage_prune = 0, const int Ppow2 = 0, const u64* __restrict__ ext0 = nullptr) {
    const int jw = c / B, b = c - jw * B;
    *image = b;
    // sym_arg: 0 general, 1 symmetric, 2 as wsym_check_kernel found, 3 symmetric on trust (the check runs in this launch: wsym_check_in_launch)
    const int sym = sym_arg == 2 ? (img_ptrs(ws, L, b).misc[3] == 0 ? 1 : 0) : (sym_arg == 3 ? (img_ptrs(ws, L, b).misc[2] == 0 ? 1 : 0) : sym_arg);
    if (!sym) {
        if (jw != spw - 1) return 0;
        leaders_body(N, counts, ws, L, b);
        return 3;
    }
    // symmetric: the image's own super-blocks (ragged counts: fewer than the launch provides for) in contiguous ranges of q
    const int n = gnms_count(counts, b, N);
    const int nsb = max(1, (((n + 63) >> 6) + kSB - 1) / kSB);
    const int q = (nsb + spw - 1) / spw;
    const int j0 = jw * q, j1 = min(nsb, j0 + q);
    if (j0 >= j1) return 0;                                        // (workgroup-uniform)
    return leaders_sb_body<STAGE>(N, counts, ws, L, b, j0, j1, jw, stage_src, stage_ld, stage_thr, stage_temp, stage_prune, Ppow2, 0u, 0, 0, ext0);
}

__global__ __launch_bounds__(1024) void leaders_kernel(int N, const int* __restrict__ counts, char* ws, gnms_ws_layout L, int sym, int B, int spw,
                                                       const u64* __restrict__ ext0 = nullptr) {
    int b;
    leaders_chain(N, counts, ws, L, B, spw, (int)blockIdx.x, sym, &b, nullptr, 0, 0.0f, 0.0f, 0, 0, ext0);
}

// ------------------------------------------------------------------------------------------------
// K4: attribution.  One wave per rank block: walk the leaders with rank < 64(kb+1) in order, 64 per step;
// an exclusive OR-scan across lanes tells each leader which bits it is the FIRST to claim.
//   rem[k] = rank of the leader that removed rank k (k itself for a leader), gpos[k] = that leader's ordinal.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 slab_col(const ImgPtrs& I, const gnms_ws_layout& L, int bb, int k) { return I.W[(size_t)bb * L.NC + k]; }

// one wave: rank block kb of image b.  Besides the attribution it evaluates, in parallel over all rank blocks, the overlap of
// every rank with the leader that removed it (plead[k], groups_kernel's membership test) -- as a prologue of groups_kernel these
// N dependent gathers ran on ONE CU (50 us of its 180 at N=16384).  `src`: see kFromMatrix / kFromBoxes / kFromRecords.
template <int SRC>
__device__ __forceinline__ void attribute_body(const float* __restrict__ src, long ld, int N, const int* __restrict__ counts, float thr, char* ws,
                                               gnms_ws_layout L, const int b, const int kb, const int lane, const int sym_arg = 0) {
    __shared__ int att_lead[16][64];                   // per wave: ordinal of the leader that claimed rank k0 + i
    int* my_lead = att_lead[(threadIdx.x >> 6) & 15];
    const int n = gnms_count(counts, b, N);
    const int k0 = kb << 6;
    if (k0 >= n) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    const int sym = sym_arg == 2 ? (I.misc[3] == 0 ? 1 : 0) : sym_arg;  // (2: wsym_check_kernel's verdict)
    const int nrows = min(64, n - k0);
    const u64 want = (nrows >= 64) ? ~0ull : ((1ull << nrows) - 1ull);
    const u64* slab = I.W + (size_t)kb * L.NC;
    const int nl = I.leadpfx[kb + 1];                 // leaders with rank < k0 + 64
    my_lead[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    u64 acc = 0;
    if (sym) {
        // the leader scan has attributed already (leaders_body, sym): rem[k] is final, what is left is the leader's ordinal and the
        // overlap with it
        if (lane < nrows) {
            const int k = k0 + lane;
            const int lr = rem_load(I.rem, k);
            const int g = I.leadpfx[lr >> 6] + __builtin_popcountll(I.leadw[lr >> 6] & ((1ull << (lr & 63)) - 1ull));
            I.gpos[k] = g;
            const float* m = overlap_src<SRC>(src, I, b, N, ld);
            I.plead[k] = overlap_at<SRC>(m, ld, I.order[k], I.leadc[g], thr);
        }
        return;
    }
    for (int base = 0; base < nl && (acc & want) != want; base += 64) {
        const int t = base + lane;
        u64 w = 0;
        int lr = -1;
        if (t < nl) {
            lr = I.leadr[t];
            w = slab[lr];
            if (lr >= k0) w |= 1ull << (lr - k0);     // the leader's own slot
            w &= want;
        }
        const u64 ex = wave_or_exclusive_scan(w, lane);
        u64 mine = w & ~(acc | ex);
        while (mine) {
            const int bit = __builtin_ctzll(mine);
            I.rem[k0 + bit] = lr;
            my_lead[bit] = t;
            mine &= mine - 1;
        }
        acc |= gnms_wave_or(w);
    }
    __builtin_amdgcn_wave_barrier();                  // LDS is in order within a wave: every claimed slot is visible below
    if (lane < nrows) {
        const int k = k0 + lane;
        const int g = my_lead[lane];                  // every rank is claimed: by an earlier leader, or by itself
        I.gpos[k] = g;                                // ordinal of the leader (groups_kernel's sort key; it overwrites gpos afterwards)
        const float* m = overlap_src<SRC>(src, I, b, N, ld);
        I.plead[k] = overlap_at<SRC>(m, ld, I.order[k], I.leadc[g], thr);    // likewise overwritten by groups_kernel's own plead
    }
}

// K4 for one image by its WHOLE workgroup (1024 threads; the chain kernels below).  sym: the scan has attributed (rem[] final); what
// is left per rank is three levels of dependent gathers (leader ordinal, order / leadc, the two boxes) whose latency beside the write
// stream is microseconds each: four ranks of a thread side by side, every load of a level issued before the first use, stores last.
template <int SRC>
__device__ __forceinline__ void attribute_image(const float* __restrict__ src, long ld, int N, const int* __restrict__ counts, float thr, char* ws,
                                                gnms_ws_layout L, const int b, const int sym_arg) {
    const int tid = threadIdx.x;
    const int sym = sym_arg == 2 ? (img_ptrs(ws, L, b).misc[3] == 0 ? 1 : 0) : sym_arg;
    if (!sym) {
        for (int kb = tid >> 6; kb < L.NB; kb += 16) attribute_body<SRC>(src, ld, N, counts, thr, ws, L, b, kb, tid & 63, 0);
        return;
    }
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const float* m = overlap_src<SRC>(src, I, b, N, ld);
    for (int k0 = 0; k0 < n; k0 += 4096) {
        int g4[4], ca[4], cb[4], lr4[4];
        float pl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + tid + e * 1024;
            const bool ok = k < n;
            lr4[e] = ok ? rem_load(I.rem, k) : 0;
            ca[e] = ok ? I.order[k] : 0;
        }
        // (the leader's input index is order[its rank]: the same value as leadc[its ordinal], one dependent load earlier)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int lr = lr4[e];
            g4[e] = I.leadpfx[lr >> 6] + __builtin_popcountll(I.leadw[lr >> 6] & ((1ull << (lr & 63)) - 1ull));
            cb[e] = (k0 + tid + e * 1024 < n) ? I.order[lr] : 0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pl[e] = (k0 + tid + e * 1024 < n) ? overlap_at<SRC>(m, ld, ca[e], cb[e], thr) : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + tid + e * 1024;
            if (k < n) { I.gpos[k] = g4[e]; I.plead[k] = pl[e]; }
        }
    }
}

template <int SRC>
__global__ __launch_bounds__(64) void attribute_kernel(const float* __restrict__ src, long ld, int N, const int* __restrict__ counts, float thr,
                                                       char* ws, gnms_ws_layout L, int sym) {
    attribute_body<SRC>(src, ld, N, counts, thr, ws, L, (int)blockIdx.y, (int)blockIdx.x, (int)threadIdx.x, sym);
}

// ------------------------------------------------------------------------------------------------
// K5: groups.  member(k) = iou[order[k]][order[rem[k]]] > thr (strict, :249); sort (leader, rank) in LDS;
// runs of equal leader are the groups, their first group_size+1 entries survive (:253-255), the first
// entry is the column the mask keeps (:99).  MASKED: the default rescoring (:95-105,:111) is fused:
//   pre_k = s_k - prune(iou[k][head]) * s_head      (I - P restricted to the head column)
// Arrays head/gpos/gstart/glen/gsorted/plead are indexed by rank; pre is indexed by NMS position q
// (q = rank for hard sort, q = input index when presorted).
// ------------------------------------------------------------------------------------------------
// FUSE (E <= 4, the chain kernels after a scan that has attributed: leaders_body with sym): K4's work -- the leader's ordinal and the
// overlap with it -- is done HERE, for the very ranks phase 1 and phase 3 of this thread own (k = e * 1024 + t), and what K4 would have
// parked in global memory for them (gpos, plead) or they would have loaded again (order, sscore, rem, the leader's score) stays in
// registers across the sort.  Beside the matrix writers every dependent global load of the chain costs ~2 us: K4 -> barrier -> K5 was
// eight levels of them, this is four (rem / order / sscore -> the leader's order, score, ordinal -> the two boxes -> stores).
constexpr int kBigGroupList = 16;     // groups above this size are also listed from the end of hlist (groups_body, solve_groups_kernel)

template <int E, int SRC, bool FUSE = false>
__device__ __forceinline__ void groups_body(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                            gnms_params P, char* ws, gnms_ws_layout L, int Ppow2, const int b) {
    static_assert(!FUSE || E <= 4, "the fused attribution keeps five values per owned rank in registers");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* keys = reinterpret_cast<unsigned*>(smem);          // (leader ordinal << 14) | rank : 28 bits (N <= 16384)
    unsigned* info = keys + Ppow2;                               // per rank: head | pos << 14, or ~0 (in no group)
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const float* m = overlap_src<SRC>(iou, I, b, N, ld);                      // kFromBoxes: `iou` holds the boxes [B][N][4]
    const float thr = P.nms_threshold;
    // ---- phase 1: membership key of every rank, lane-contiguous ranks (k = e * blockDim + t: coalesced loads), straight into LDS ----
    GNMS_T0();
    // key = (ordinal of the leader << 14) | rank: the ordinal (attribute_kernel left it in gpos, with the overlap against that
    // leader in plead) needs only log2(#leaders) bits, i.e. ONE 7-bit radix pass for up to 127 groups
    int f_lr[FUSE ? E : 1], f_ck[FUSE ? E : 1];                       // FUSE: rem[k], order[k], sscore[k], overlap with and score of the leader
    float f_sk[FUSE ? E : 1], f_pl[FUSE ? E : 1], f_sl[FUSE ? E : 1];
    if constexpr (FUSE) {
        int g[E], cb[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = e * (int)blockDim.x + (int)threadIdx.x;
            const bool ok = k < n;
            f_lr[e] = ok ? rem_load(I.rem, k) : 0;
            f_ck[e] = ok ? I.order[k] : 0;
            f_sk[e] = ok ? I.sscore[k] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int lr = f_lr[e];
            g[e] = I.leadpfx[lr >> 6] + __builtin_popcountll(I.leadw[lr >> 6] & ((1ull << (lr & 63)) - 1ull));
            cb[e] = I.order[lr];                                        // (= leadc[g]: the leader's input index)
            f_sl[e] = I.sscore[lr];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = e * (int)blockDim.x + (int)threadIdx.x;
            f_pl[e] = (k < n) ? overlap_at<SRC>(m, ld, f_ck[e], cb[e], thr) : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = e * (int)blockDim.x + (int)threadIdx.x;
            keys[k] = (k < n && f_pl[e] > thr) ? (((unsigned)g[e] << 14) | (unsigned)k) : ~0u;   // strict > (:249)
            info[k] = ~0u;
        }
    } else {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = e * (int)blockDim.x + (int)threadIdx.x;
        unsigned key = ~0u;
        if (k < n && I.plead[k] > thr) key = ((unsigned)I.gpos[k] << 14) | (unsigned)k;     // strict > (:249)
        keys[k] = key;
        info[k] = ~0u;
    }
    }
    GNMS_TACC(8);
    // group by leader: the keys start in rank order, so a STABLE sort on the leader bits alone yields (leader, rank) order
    __shared__ unsigned radix_hist[128 * 16 + 16];
    __syncthreads();
    {
        const int G = I.misc[0];                                           // number of leaders; the all-ones digit is reserved for the padding keys
        block_radix_pass7<E>(keys, 14, radix_hist);
        if (G > 127) block_radix_pass7<E>(keys, 21, radix_hist);
        if (G > 16383) block_radix_pass7<E>(keys, 28, radix_hist);
    }
    GNMS_TACC(9);
    // ---- phase 2: runs of equal leader are the groups; cap, head, position.  Thread t owns sorted positions
    //      t*E .. t*E+E-1; the start of each position's run comes from a max-scan of the run-start flags
    //      (registers -> wave shuffles -> 16 per-wave totals in LDS): no per-element search. ----
    const long long cap = (long long)P.group_size + 1;
    __shared__ int wave_last_start[16];
    {
        const int t = threadIdx.x, ln = t & 63, wv = t >> 6;
        unsigned ky[E];
        int st[E];                                                     // run start of position i, or -1 if it lies in an earlier thread
        int last = -1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = t * E + e;
            ky[e] = keys[i];
            const bool valid = ky[e] != ~0u;
            const bool first = valid && (i == 0 || (keys[i - 1] >> 14) != (ky[e] >> 14));
            if (first) last = i;
            st[e] = last;
        }
        // inclusive max-scan of `last` over the wave, then exclusive value for this thread
        const int inc = gnms_max_scan32(last);                         // DPP running maximum
        int excl = __builtin_amdgcn_update_dpp(-1, inc, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 keeps -1)
        if (ln == 0) excl = -1;
        if (ln == 63) wave_last_start[wv] = inc;
        __syncthreads();
        int carry = -1;
        for (int w = 0; w < wv; ++w) carry = max(carry, wave_last_start[w]);
        const int before = max(excl, carry);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = t * E + e;
            bool big_head = false;
            int hk = 0;
            if (i < n) {
                if (ky[e] != ~0u) {
                    const int k = (int)(ky[e] & 0x3fffu);
                    const int start = st[e] >= 0 ? st[e] : before;
                    const long long pos = i - start;
                    const unsigned hd = keys[start] & 0x3fffu;
                    if (pos < cap) info[k] = hd | ((unsigned)pos << 14);
                    const bool lastofrun = (i + 1 >= n) || (keys[i + 1] == ~0u) || ((keys[i + 1] >> 14) != (ky[e] >> 14));
                    if (lastofrun) {                                  // the run's last element publishes the extent for the head
                        const long long len = pos + 1;
                        I.gstart[hd] = start;
                        I.glen[hd] = (int)(len < cap ? len : cap);
                        big_head = len > 1;
                        hk = (int)hd;
                        // groups of more than kBigGroupList members are listed a second time, from the END of hlist (misc[4] of them): the
                        // unmasked solves hand those to whole workgroups and everything smaller to single waves (nms_solve_kernels.h)
                        if ((len < cap ? len : cap) > kBigGroupList) I.hlist[N - 1 - atomicAdd(&I.misc[4], 1)] = hk;
                    }
                }
            }
            // heads of multi-member groups go to hlist (one atomic per wave)
            const unsigned long long bm = __ballot(big_head);
            if (bm) {
                int base = 0;
                if (ln == __builtin_ctzll(bm)) base = atomicAdd(&I.misc[1], __builtin_popcountll(bm));
                base = __builtin_amdgcn_readlane(base, __builtin_ctzll(bm));
                if (big_head) I.hlist[base + __builtin_popcountll(bm & ((1ull << ln) - 1ull))] = hk;
            }
        }
    }
    // members in (group, rank) order, written lane-contiguously straight from the sorted keys
    for (int i = threadIdx.x; i < n; i += blockDim.x) I.gsorted[i] = (keys[i] == ~0u) ? -1 : (int)(keys[i] & 0x3fffu);
    __syncthreads();
    GNMS_TACC(10);
    // ---- phase 3 (lane-contiguous ranks k = e * blockDim + t: every load and store below is coalesced; with the thread-contiguous
    //      ownership of phase 1 the stores of a wave were E * 4 bytes apart -- a different 64-byte line per lane at E = 16):
    //      group fields and, for MASKED groups, the default rescoring
    //      pre_k = s_k - prune(iou[k][head]) * s_head  (I - P restricted to the head column, :95-105,:111) ----
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = e * (int)blockDim.x + (int)threadIdx.x;
        if (k >= N) continue;
        const unsigned inf = info[k];
        const int h = (k < n && inf != ~0u) ? (int)(inf & 0x3fffu) : -1;
        I.head[k] = h;
        I.gpos[k] = (h >= 0) ? (int)(inf >> 14) : -1;
        if (h != k) I.glen[k] = 0;                                   // only heads carry an extent
        if (!P.mask_group_boxes) continue;
        float pre = 0.0f, pl = 0.0f;
        int ck;
        float sk;
        if constexpr (FUSE) { ck = f_ck[e]; sk = f_sk[e]; }
        else { ck = (k < n) ? I.order[k] : 0; sk = (k < n) ? I.sscore[k] : 0.0f; }
        const int q = P.presorted ? ck : k;
        if (h == k) {
            pre = sk;
        } else if (h >= 0) {
            int lr;
            if constexpr (FUSE) lr = f_lr[e]; else lr = rem_load(I.rem, k);
            float v, sh;
            int ch;
            if (h != lr) {                                           // the leader itself is not a member (NaN / <= thr diagonal)
                ch = I.order[h];
                v = overlap_at<SRC>(m, ld, ck, ch, thr);
                sh = I.sscore[h];
            } else {
                ch = -1;
                if constexpr (FUSE) { v = f_pl[e]; sh = f_sl[e]; }
                else {
                    v = I.plead[k];                                  // overlap with the leader, left there by attribute_kernel
                    sh = I.sscore[lr];
                }
            }
            bool tril = true;                                        // torch.tril in NMS order (:72): always true for hard sort
            if (P.presorted) { if (ch < 0) ch = I.order[h]; tril = ch < ck; }
            if (tril) pl = gnms_prune(v, thr, P.temperature, P.pruning_method);
            pre = sk - pl * sh;
        }
        I.plead[k] = pl;
        I.pre[(k < n) ? q : k] = pre;
        if constexpr (FUSE) {
            // K6 starts from the clamped value by NMS position and ends with the input indices of the listed boxes: both are at hand
            // here and wait for it in LDS (beyond keys / info, both dead for this thread's rank by now) -- finalize_body<E, true> then
            // begins without a global load and the barrier in front of it need not wait for the stores above
            if (k < n) {
                const float r2 = pre < 0.0f ? 0.0f : (pre > 1.0f ? 1.0f : pre);      // torch.clamp keeps NaN
                I.r2[q] = r2;
                reinterpret_cast<float*>(smem)[q] = r2;                              // finalize_body's `stage` (the key region)
                reinterpret_cast<int*>(smem + (size_t)Ppow2 * 8)[k] = ck;            // order[] by rank
                reinterpret_cast<float*>(smem + (size_t)Ppow2 * 12)[q] = r2;         // a copy the key sort does not overwrite
            }
        }
    }
    GNMS_TACC(11);
}

template <int E, int SRC>
__global__ __launch_bounds__(1024) void groups_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                      gnms_params P, char* ws, gnms_ws_layout L, int Ppow2) {
    groups_body<E, SRC>(iou, N, ld, counts, P, ws, L, Ppow2, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// K6: finalize (lib/groomed_nms.py:111-129).  r2 = clamp(pre,0,1); r = r2 with (< valid_thr) zeroed;
// the reference then sorts r descending (:116-121).  After thresholding every invalid box is an exact 0,
// so a STABLE descending sort is: [NaN boxes by position][valid boxes by (r desc, position)][invalid boxes
// by position].  Only the valid subset needs a sort; the other two are stream compactions (one packed
// block scan).  valid / invalid are INPUT indices padded with -1; prob is written in the order the
// reference returns it.
// ------------------------------------------------------------------------------------------------
// STAGED (behind groups_body<E, SRC, true>, masked groups): the clamped values already sit in `stage` and the input index of every
// rank in LDS behind the key region (ordL); r2 is in global memory as well.
template <int E, bool STAGED = false>
__device__ __forceinline__ void finalize_body(int N, const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L,
                                              int Ppow2, float* __restrict__ prob, long long* __restrict__ valid,
                                              long long* __restrict__ invalid, int* __restrict__ nvalid, int* __restrict__ ninvalid,
                                              const int b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);                     // [Ppow2]
    __shared__ u64 wave_tot[16];
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const float vthr = P.valid_box_prob_threshold;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, T = blockDim.x;
    GNMS_T0();
    // clamp: global loads and stores with lane-contiguous positions (coalesced), staged in LDS for the owner threads below,
    // which need thread-contiguous positions so that the compaction keeps the order
    float* stage = reinterpret_cast<float*>(smem);                 // [Ppow2] floats inside the key region (not yet in use)
    const int* ordL = reinterpret_cast<const int*>(smem + (size_t)Ppow2 * 8);   // STAGED: order[] by rank, behind the key region
    if constexpr (!STAGED) {
        for (int q = t; q < n; q += T) {
            const float pre = I.pre[q];
            const float r2 = pre < 0.0f ? 0.0f : (pre > 1.0f ? 1.0f : pre);          // torch.clamp keeps NaN
            I.r2[q] = r2;
            stage[q] = r2;
        }
        __syncthreads();
    }
    // classify
    u64 key[E];
    int cls[E];                                                    // 0 nan, 1 valid, 2 invalid, 3 padding
    u64 packed = 0;                                                // nan | valid << 16 | invalid << 32 (each count <= N <= 16384)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int q = t * E + e;
        cls[e] = 3;
        key[e] = ~0ull;
        if (q < n) {
            const float r2 = stage[q];
            const float rr = (r2 < vthr) ? 0.0f : r2;                           // :115
            cls[e] = (rr != rr) ? 0 : ((rr >= vthr) ? 1 : ((rr < vthr) ? 2 : 0)); // vthr NaN: neither list (:118-123)
            key[e] = ((u64)gnms_desc_key(rr) << 32) | (unsigned)q;
            packed += (cls[e] == 0) ? 1ull : (cls[e] == 1 ? (1ull << 16) : (1ull << 32));
        }
    }
    // exclusive block scan of the packed counters
    // (two 32-bit DPP prefix sums: the low word carries nan | valid << 16 without overflow between the fields, the high word invalid)
    const u64 inc = (u64)gnms_add_scan32((unsigned)(packed & 0xffffffffu)) | ((u64)gnms_add_scan32((unsigned)(packed >> 32)) << 32);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();                                               // every owner has read its staged values
    u64 base = 0, total = 0;
    const int nwaves = T >> 6;
    for (int w = 0; w < nwaves; ++w) { const u64 v = wave_tot[w]; if (w < wave) base += v; total += v; }
    u64 run = base + inc - packed;
    const int n_nan = (int)(total & 0xffff), nv = (int)((total >> 16) & 0xffff), ni = (int)(total >> 32);
    const int n_ge = n_nan + nv;
    if (!valid && !invalid && !P.return_sorted_prob) {
        // the caller wants the probabilities only (training reads just the third return value, lib/loss/rpn_3d.py:791):
        // no compaction, no sort of the valid boxes -- only the counts and prob
        float* pbq = prob + (size_t)b * N;
        for (int j = t; j < N; j += T) {
            float out = 0.0f;
            // (STAGED: no full barrier since groups_body stored r2 -- the staged copy is the one every thread can see)
            if (j < n) { const float r2j = STAGED ? stage[j] : I.r2[j]; out = P.group_boxes ? r2j : ((r2j < vthr) ? 0.0f : r2j); }   // :124-127
            pbq[j] = out;
            I.sidx[j] = j;
        }
        if (t == 0) {
            if (nvalid) nvalid[b] = nv;
            if (ninvalid) ninvalid[b] = ni;
        }
        return;
    }
    for (int i = t; i < Ppow2; i += T) keys[i] = ~0ull;           // (the staged values end here)
    __syncthreads();
    // sidx layout: [0,n_nan) NaN by position, [n_nan, n_ge) valid (sorted below), [n_ge, n_ge+ni) invalid by position
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int q = t * E + e;
        if (cls[e] == 0) { I.sidx[(int)(run & 0xffff)] = q; run += 1ull; }
        else if (cls[e] == 1) { keys[(int)((run >> 16) & 0xffff)] = key[e]; run += 1ull << 16; }
        else if (cls[e] == 2) {
            const int j = (int)(run >> 32);
            I.sidx[n_ge + j] = q;
            if (invalid) invalid[(size_t)b * N + j] = P.presorted ? q : (STAGED ? ordL[q] : I.order[q]);
            run += 1ull << 32;
        }
    }
    __syncthreads();
    GNMS_TACC(12);
    // sort the valid keys: they sit in keys[0..nv), padded with ~0
    if (nv > 1) {
        if (nv <= 256 && 2 * nv <= Ppow2) {
            // a few valid boxes (typically ~100): rank of each key = number of smaller keys (they are distinct), read as LDS broadcasts
            u64* sorted = keys + Ppow2 / 2;
            u64 mine = 0;
            int rank = 0;
            if (t < nv) {
                mine = keys[t];
                for (int j = 0; j < nv; ++j) rank += (keys[j] < mine) ? 1 : 0;
            }
            __syncthreads();
            if (t < nv) sorted[rank] = mine;
            __syncthreads();
            if (t < nv) keys[t] = sorted[t];
            __syncthreads();
        } else if (nv <= T) {
            u64 r1[1] = {keys[t]};
            __syncthreads();
            int pe = 64;                                                   // typically ~100 valid boxes: 128 keys, one merge pass instead of four
            while (pe < nv) pe <<= 1;
            block_sort<1, u64>(r1, keys, pe);
        } else {
            // more valid boxes than threads: sort next_pow2(nv) keys, not all Ppow2 (the valid keys are compacted at the front)
            int pe = 2 * T;
            while (pe < nv) pe <<= 1;
            bool done = false;
            if constexpr (E >= 4) {
                if (pe == 2 * T) {
                    u64 r[2];
                    r[0] = keys[t * 2]; r[1] = keys[t * 2 + 1];
                    __syncthreads();
                    block_sort<2, u64>(r, keys, pe);
                    done = true;
                }
            }
            if constexpr (E >= 8) {
                if (!done && pe == 4 * T) {
                    u64 r[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = keys[t * 4 + e];
                    __syncthreads();
                    block_sort<4, u64>(r, keys, pe);
                    done = true;
                }
            }
            if constexpr (E >= 16) {
                if (!done && pe == 8 * T) {
                    u64 r[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = keys[t * 8 + e];
                    __syncthreads();
                    block_sort<8, u64>(r, keys, pe);
                    done = true;
                }
            }
            if (!done) {
                u64 r[E];
#pragma unroll
                for (int e = 0; e < E; ++e) r[e] = keys[t * E + e];
                __syncthreads();
                block_sort<E, u64>(r, keys, Ppow2);
            }
        }
    }
    GNMS_TACC(13);
    float* pb = prob + (size_t)b * N;
    for (int j = t; j < N; j += T) {
        if (j < nv) {
            const int q = (int)(keys[j] & 0xffffffffu);
            I.sidx[n_nan + j] = q;
            if (valid) valid[(size_t)b * N + j] = P.presorted ? q : (STAGED ? ordL[q] : I.order[q]);
        } else if (valid) valid[(size_t)b * N + j] = -1;
        if (j >= ni && invalid) invalid[(size_t)b * N + j] = -1;
        if (j >= n) I.sidx[j] = j;
    }
    // (STAGED, unsorted output: every value of the last loop comes from LDS -- no barrier for sidx, no global load)
    const float* r2L = reinterpret_cast<const float*>(smem + (size_t)Ppow2 * 12);
    if (!STAGED || P.return_sorted_prob) __syncthreads();
    for (int j = t; j < N; j += T) {
        float out = 0.0f;
        if (j < n) {
            if (P.return_sorted_prob) { const int sj = I.sidx[j]; const float r2q = STAGED ? r2L[sj] : I.r2[sj]; out = (r2q < vthr) ? 0.0f : r2q; }   // :117
            else { const float r2j = STAGED ? r2L[j] : I.r2[j]; out = P.group_boxes ? r2j : ((r2j < vthr) ? 0.0f : r2j); }       // :124-127
        }
        pb[j] = out;
    }
    GNMS_TACC(14);
    if (t == 0) {
        if (nvalid) nvalid[b] = nv;
        if (ninvalid) ninvalid[b] = ni;
    }
}

template <int E>
__global__ __launch_bounds__(1024) void finalize_kernel(int N, const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L,
                                                        int Ppow2, float* __restrict__ prob, long long* __restrict__ valid,
                                                        long long* __restrict__ invalid, int* __restrict__ nvalid,
                                                        int* __restrict__ ninvalid) {
    finalize_body<E>(N, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// The FAST TAIL (round 5): masked groups, hard sort, symmetric scan, N <= 4096.
//   leaders_sb_body<STAGE>   every scan workgroup: head / plead / pre / r2 of its own ranks right behind its resolve (see there)
//   fast_final_body          the image's last scan workgroup: verdict (no leader outside its own group, no group above the cap), then K6
//                            from LDS -- or, verdict "slow", K5 proper (groups_body) as before
//   csr_build_body           ONE MORE workgroup per image, beside K6: the groups as contiguous runs (gsorted / gstart / glen / gpos / hlist),
//                            which only the backward reads -- a counting sort on the leader's RANK (LDS atomics), members ranked inside
//                            their run by comparison, so that the result does not depend on the order the atomics were served in
// What it takes off the path to the probabilities: the attribution gathers (four dependent levels on one CU -> one level on nsb CUs),
// both radix passes, the run scan and K5's stores: scan + groups + finalize 63 -> ~40 us at B = 8, N = 4096 (profiles/r05*).
// ------------------------------------------------------------------------------------------------
// dynamic LDS: the scan's structures, then cnt[N] (fast_final_body); csr_build_body: 4 arrays of Ppow2 ints
__host__ __device__ __forceinline__ size_t fast_tail_cnt_offset(int NB) { return (leaders_lds_size(NB) + 15) & ~(size_t)15; }
__host__ __device__ __forceinline__ size_t fast_tail_lds_size(int N, int Ppow2) {
    const size_t a = fast_tail_cnt_offset((N + 63) / 64) + (size_t)N * 4, c = (size_t)Ppow2 * 32;     // (finalize_fast_body: 3 arrays + 2 key lists + the bucket segments)
    return a > c ? a : c;
}

// K6 of the fast tail.  In: r2 / head / input index by rank in LDS (fast_final_body).  Same outputs as finalize_body, bit for bit; what
// differs is how the valid boxes get into descending order.  A valid HEAD carries r = clamp(its own score), and the ranks ARE the scores in
// descending order (clamp is monotone, ties keep their positions): the valid heads, compacted in rank order, are a sorted run as they
// stand (list A; checked, one compare per element -- a violation sends everything through the sort).  Only the other valid boxes (members
// whose rescored value stayed above the threshold: list B, typically a few dozen) are sorted, and the two lists are merged by rank
// (position in the own list + binary search in the other).  For ~1 900 valid boxes of 4 096 that replaces a 2 048-key merge sort (9 us on
// one CU) by a compaction, a small sort and 11 LDS probes per box.  No global load and no wait for a global store anywhere.
constexpr int kDirectMergeMax = 256;     // boxes of list B up to which K6 of the fast tail places them by counting (finalize_fast_body)
template <int E>
__device__ __forceinline__ void finalize_fast_body(int N, const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L,
                                                   int Ppow2, float* __restrict__ prob, long long* __restrict__ valid,
                                                   long long* __restrict__ invalid, int* __restrict__ nvalid, int* __restrict__ ninvalid,
                                                   const int b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float* r2L = reinterpret_cast<const float*>(smem);
    const int* hdL = reinterpret_cast<const int*>(smem + (size_t)Ppow2 * 4);
    const int* ordL = reinterpret_cast<const int*>(smem + (size_t)Ppow2 * 8);
    u64* keyA = reinterpret_cast<u64*>(smem + (size_t)Ppow2 * 12);                   // [Ppow2] valid heads, in rank order
    u64* keyB = reinterpret_cast<u64*>(smem + (size_t)Ppow2 * 20);                   // [Ppow2] the other valid boxes
    __shared__ unsigned ff_tot[2][16];
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const float vthr = P.valid_box_prob_threshold;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float* pb = prob + (size_t)b * N;
    GNMS_T0();
    // classify, thread-contiguous positions (the compaction keeps the order)
    u64 key[E];
    int cls[E];                                                    // 0 nan, 1 valid head, 2 valid other, 3 invalid, 4 padding
    unsigned w0 = 0u, w1 = 0u;                                     // nan | heads << 16,  others | invalid << 16  (each count <= 4096)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int q = t * E + e;
        cls[e] = 4;
        key[e] = ~0ull;
        if (q < n) {
            const float r2 = r2L[q];
            const float rr = (r2 < vthr) ? 0.0f : r2;                              // :115
            cls[e] = (rr != rr) ? 0 : ((rr >= vthr) ? (hdL[q] == q ? 1 : 2) : ((rr < vthr) ? 3 : 0));   // vthr NaN: neither list (:118-123)
            key[e] = ((u64)gnms_desc_key(rr) << 32) | (unsigned)q;
            w0 += (cls[e] == 0) ? 1u : (cls[e] == 1 ? (1u << 16) : 0u);
            w1 += (cls[e] == 2) ? 1u : (cls[e] == 3 ? (1u << 16) : 0u);
        }
    }
    const unsigned inc0 = gnms_add_scan32(w0), inc1 = gnms_add_scan32(w1);
    if (lane == 63) { ff_tot[0][wave] = inc0; ff_tot[1][wave] = inc1; }
    lds_barrier();
    unsigned base0 = 0u, base1 = 0u, tot0 = 0u, tot1 = 0u;
    for (int w = 0; w < 16; ++w) {
        const unsigned a = ff_tot[0][w], c = ff_tot[1][w];
        if (w < wave) { base0 += a; base1 += c; }
        tot0 += a; tot1 += c;
    }
    const int n_nan = (int)(tot0 & 0xffffu), ni = (int)(tot1 >> 16);
    int nA = (int)(tot0 >> 16), nB = (int)(tot1 & 0xffffu);
    const int nv = nA + nB, n_ge = n_nan + nv;
    // the two counts leave as soon as they are known: where they go to pinned host memory (gnms_host_counts_slot: the reference's index
    // tensors have a data-dependent length) the host takes them ~3 us before the lists below are complete -- which it does not read; whatever
    // it enqueues next is behind this launch on the stream
    if (t == 0) { if (nvalid) nvalid[b] = nv; if (ninvalid) ninvalid[b] = ni; }
    if (!valid && !invalid && !P.return_sorted_prob) {
        // the caller wants the probabilities only (training reads just the third return value, lib/loss/rpn_3d.py:791)
        for (int j = t; j < N; j += 1024) pb[j] = (j < n) ? r2L[j] : 0.0f;                    // (grouped mode: the un-thresholded clone, :124-125)
        return;
    }
    const bool sorted_out = P.return_sorted_prob != 0;
    // (the bucket counters of the merge below live where the heads were: every thread has read its heads in front of the barrier above)
    for (int i = t; i < Ppow2; i += 1024) reinterpret_cast<int*>(smem + (size_t)Ppow2 * 4)[i] = 0;
    // compaction.  sidx layout: [0, n_nan) NaN by position, [n_nan, n_ge) valid (merged below), [n_ge, n_ge + ni) invalid by position
    // (sidx is the backward's map for the SORTED output only, bwd_gx_kernel: written only then)
    {
        unsigned run0 = base0 + inc0 - w0, run1 = base1 + inc1 - w1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int q = t * E + e;
            if (cls[e] == 0) { const int p = (int)(run0 & 0xffffu); if (sorted_out) { I.sidx[p] = q; pb[p] = r2L[q]; } run0 += 1u; }
            else if (cls[e] == 1) { keyA[run0 >> 16] = key[e]; run0 += 1u << 16; }
            else if (cls[e] == 2) { keyB[run1 & 0xffffu] = key[e]; run1 += 1u; }
            else if (cls[e] == 3) {
                const int j = (int)(run1 >> 16);
                if (invalid) invalid[(size_t)b * N + j] = ordL[q];
                if (sorted_out) { I.sidx[n_ge + j] = q; pb[n_ge + j] = 0.0f; }     // (thresholded, :115-117)
                run1 += 1u << 16;
            }
        }
    }
    lds_barrier();
    GNMS_TACC(12);
    // the heads must be a sorted run (they are, whenever the scores were finite and the ranks their descending order)
    int viol = 0;
    for (int i = t + 1; i < nA; i += 1024) viol |= keyA[i] <= keyA[i - 1];
    if (t < kDirectMergeMax) reinterpret_cast<int*>(smem + (size_t)Ppow2 * 28)[t] = 0;      // (the counting merge's ranks; nobody else is in that region yet)
    viol = __syncthreads_or(viol);
    auto emit = [&](const u64 k, const int p) {
        const int q = (int)(k & 0xffffffffu);
        if (valid) valid[(size_t)b * N + p] = ordL[q];
        if (sorted_out) { I.sidx[n_nan + p] = q; pb[n_nan + p] = r2L[q]; }   // (valid: >= the threshold, returned as it is)
    };
    if (viol) {                                                    // never seen; kept exact: everything through the sort
        for (int i = t; i < nA; i += 1024) keyB[nB + i] = keyA[i];
        nB += nA;
        u64 r[E];
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = (t * E + e < nB) ? keyB[t * E + e] : ~0ull;
        lds_barrier();
        block_sort<E, u64>(r, keyB, E * 1024);
        for (int i = t; i < nB; i += 1024) emit(keyB[i], i);
    } else if (nB == 0) {
        for (int i = t; i < nA; i += 1024) emit(keyA[i], i);
    } else if (nB <= kDirectMergeMax) {
        // FEW other valid boxes (up to 256; ~100-200 of 4 096 uniform boxes): no bucket counters, no prefix over the heads, no scatter by
        // atomics.  A box of B ends up at (heads in front of it: a binary search) + (boxes of B in front of it: its rank in B, counted --
        // nB^2 compares spread over all 1024 threads, thread (box, segment), the reads of a wave at ONE LDS address); the bucket numbers g
        // laid out by that rank are a sorted list, and head i ends up at i + (boxes of B whose bucket is <= i: a binary search in it).
        // Two barriers, two binary searches and nB^2 / 1024 compares per thread where the bucket merge has five barriers, two rounds of
        // LDS atomics and a scan.  Same positions -- the keys are distinct, the order is total.
        int* cntB = reinterpret_cast<int*>(smem + (size_t)Ppow2 * 28);             // [kDirectMergeMax] rank of each box of B (zeroed in front of the barrier above)
        int* gS = cntB + kDirectMergeMax;                                          // [nB] bucket numbers in B's sorted order
        {
            int lgP = 6;
            while ((1 << lgP) < nB) ++lgP;                                         // P = 64 .. 256 boxes side by side, S = 1024 / P segments
            const int j = t & ((1 << lgP) - 1), sg = t >> lgP, S = 1024 >> lgP, per = (nB + S - 1) / S;
            const int x0 = sg * per, x1 = min(nB, x0 + per);
            if (j < nB && x0 < x1) {
                const u64 k = keyB[j];
                int c = 0, x = x0;
                for (; x + 4 <= x1; x += 4) c += (keyB[x] < k ? 1 : 0) + (keyB[x + 1] < k ? 1 : 0) + (keyB[x + 2] < k ? 1 : 0) + (keyB[x + 3] < k ? 1 : 0);
                for (; x < x1; ++x) c += keyB[x] < k ? 1 : 0;
                if (c) atomicAdd(&cntB[j], c);
            }
        }
        lds_barrier();
        if (t < nB) {
            const u64 k = keyB[t];
            const int r = cntB[t], g = lower_bound_lds<u64>(keyA, nA, k);
            gS[r] = g;
            emit(k, g + r);
        }
        lds_barrier();
        for (int i = t; i < nA; i += 1024) {
            int lo = 0, hi = nB;                                                   // boxes of B with bucket <= i
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (gS[mid] <= i) lo = mid + 1; else hi = mid; }
            emit(keyA[i], i + lo);
        }
    } else {
        // MERGE BY BUCKETS.  The sorted heads split the key space into nA + 1 buckets; a box of B lies in bucket g = number of heads in front
        // of it (a binary search), and its final position is g + (boxes of B in earlier buckets) + (its rank among its bucket mates); head
        // i ends up at i + (boxes of B in buckets <= i).  So B is not sorted as long as the buckets are small (a box or two on detector-like
        // scores): count per bucket (LDS atomics), prefix over the buckets, members scattered into their bucket's segment (the scatter's
        // atomics turn the exclusive prefix into the inclusive one), rank inside the segment by comparison.  A bucket of more than
        // kBucketMax boxes -- scores that nearly tie put every rescored member behind the last head: the C3 harness's random-init network
        // does exactly that, 2 281 boxes in one bucket, and ranking them by comparison took 0.5 ms -- sends B through the sort instead.
        // (nB >= 1, hence nA + 1 <= n <= Ppow2 counters: the head array, dead by now.)
        constexpr int kBucketMax = 32;
        int* cntG = reinterpret_cast<int*>(smem + (size_t)Ppow2 * 4);              // [nA + 1] (zeroed during the compaction above)
        int* seg = reinterpret_cast<int*>(smem + (size_t)Ppow2 * 28);              // [nB] indices into keyB, bucket by bucket
        int gB[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = t + e * 1024;
            gB[e] = 0;
            if (i < nB) { gB[e] = lower_bound_lds<u64>(keyA, nA, keyB[i]); atomicAdd(&cntG[gB[e]], 1); }
        }
        lds_barrier();
        int big;
        {   // exclusive prefix over the nA + 1 buckets, in place (thread t owns buckets t * E .. t * E + E - 1); the largest bucket on the way
            int c[E], sum = 0, mx = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) { const int g = t * E + e; c[e] = (g <= nA) ? cntG[g] : 0; sum += c[e]; mx = max(mx, c[e]); }
            const unsigned inc = gnms_add_scan32((unsigned)sum);
            if (lane == 63) ff_tot[0][wave] = inc;
            big = __syncthreads_or(mx > kBucketMax);
            unsigned base = 0u;
            for (int w = 0; w < 16; ++w) if (w < wave) base += ff_tot[0][w];
            int run = (int)(base + inc) - sum;
#pragma unroll
            for (int e = 0; e < E; ++e) { const int g = t * E + e; if (g <= nA) cntG[g] = run; run += c[e]; }
        }
        lds_barrier();
        if (!big) {
#pragma unroll
            for (int e = 0; e < E; ++e) { const int i = t + e * 1024; if (i < nB) seg[atomicAdd(&cntG[gB[e]], 1)] = i; }
            lds_barrier();                                         // cntG[g] is now the INCLUSIVE prefix: boxes of B in buckets <= g
            for (int i = t; i < nA; i += 1024) emit(keyA[i], i + cntG[i]);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = t + e * 1024;
                if (i < nB) {
                    const int g = gB[e], s0 = g > 0 ? cntG[g - 1] : 0, s1 = cntG[g];
                    const u64 k = keyB[i];
                    int rank = 0;
                    for (int j = s0; j < s1; ++j) rank += keyB[seg[j]] < k ? 1 : 0;
                    emit(k, g + s0 + rank);
                }
            }
        } else {
            // B through the sort; a head's position still comes from the bucket prefix (cntG[i + 1]: boxes of B in buckets <= i)
            for (int i = t; i < nA; i += 1024) emit(keyA[i], i + cntG[i + 1]);
            if (nB <= 1024) {
                u64 r1[1] = {t < nB ? keyB[t] : ~0ull};
                lds_barrier();
                int pe = 64;
                while (pe < nB) pe <<= 1;
                block_sort<1, u64>(r1, keyB, pe);
            } else {
                u64 r[E];
#pragma unroll
                for (int e = 0; e < E; ++e) r[e] = (t * E + e < nB) ? keyB[t * E + e] : ~0ull;
                lds_barrier();
                block_sort<E, u64>(r, keyB, E * 1024);
            }
            for (int j = t; j < nB; j += 1024) { const u64 k = keyB[j]; emit(k, j + lower_bound_lds<u64>(keyA, nA, k)); }
        }
    }
    GNMS_TACC(13);
    for (int j = t; j < N; j += 1024) {
        if (j >= nv && valid) valid[(size_t)b * N + j] = -1;
        if (j >= ni && invalid) invalid[(size_t)b * N + j] = -1;
        if (sorted_out && j >= n) I.sidx[j] = j;
        if (!sorted_out) pb[j] = (j < n) ? r2L[j] : 0.0f;          // (grouped mode: the un-thresholded clone, :124-125)
        else if (j >= n) pb[j] = 0.0f;
    }
    GNMS_TACC(14);
}

template <int E, int SRC, bool FUSED = false>
__device__ __forceinline__ void fast_final_body(const float* __restrict__ src, int N, long ld, const int* __restrict__ counts, gnms_params P,
                                                char* ws, gnms_ws_layout L, int Ppow2, float* __restrict__ prob, long long* __restrict__ valid,
                                                long long* __restrict__ invalid, int* __restrict__ nvalid, int* __restrict__ ninvalid,
                                                const int b, const int last, const unsigned tag = 0u) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* r2L = reinterpret_cast<float*>(smem);                                     // r2 by rank (= position)
    int* hdL = reinterpret_cast<int*>(smem + (size_t)Ppow2 * 4);                     // head by rank
    int* ordL = reinterpret_cast<int*>(smem + (size_t)Ppow2 * 8);                    // input index by rank
    int* cnt = reinterpret_cast<int*>(smem + fast_tail_cnt_offset(L.NB));            // members per leader rank (dead once the verdict is in)
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int tid = threadIdx.x;
    const int nb = (n + 63) >> 6;
    const int nsb = max(1, (nb + kSB - 1) / kSB);
    const int own0 = (nsb - 1) * kSB * 64;                                           // the parked super-block: [own0, n)
    const u64 epoch = FUSED ? ((u64)tag << 32) : ((u64)(unsigned)I.misc[8] << 32);
    const long long cap = (long long)P.group_size + 1;
    int slow = last == 2;
    GNMS_T0();
    if (!slow) {
        // the other super-blocks' ranks: what their workgroups stored (write-through, in front of their granules)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = e * 1024 + tid;
            if (k < own0) {
                const int hd = rem_load(I.head, k);
                const float r2 = __hip_atomic_load(I.r2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int ck = I.order[k];
                r2L[k] = r2; hdL[k] = hd; ordL[k] = ck;
            }
        }
        // (a group above kFastGroupMax members also takes K5 proper: csr_build_body ranks the members of a run by comparison, quadratic in its length)
        const long long lim = cap < kFastGroupMax ? cap : kFastGroupMax;
        if (lim < (long long)n) {                                                    // (else no group can be longer than the limit)
            for (int i = tid; i < n; i += 1024) cnt[i] = 0;
            lds_barrier();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int k = e * 1024 + tid;
                const int hd = (k < n) ? hdL[k] : -1;
                if (hd >= 0) atomicAdd(&cnt[hd], 1);
            }
            lds_barrier();
            int over = 0;
            for (int i = tid; i < n; i += 1024) over |= (long long)cnt[i] > lim;
            slow = __syncthreads_or(over);
        }
    }
    // this workgroup's own write-through stores (leaders_sb_body) are acknowledged before the verdict -- csr_build_body's go-ahead -- leaves
    GNMS_TACC(8);
    __builtin_amdgcn_s_waitcnt(0x0f70);                                              // vmcnt(0)
    __syncthreads();
    GNMS_TACC(9);
    if (tid == 0) gran_store(I.gran + (size_t)16 * 32 + kGranVerdict, FUSED ? strong_gran(tag, kSlotVerdict + (slow ? 2u : 1u)) : (epoch | (slow ? 2ull : 1ull)));
    if (slow) {
        size_t oa, ol, oc, op;
        leaders_lds_layout(L.NB, &oa, &ol, &oc, &op);
        leaders_epilogue(I, reinterpret_cast<const u64*>(smem + ol), nb, reinterpret_cast<int*>(smem));
        __syncthreads();
        groups_body<E, SRC, true>(src, N, ld, counts, P, ws, L, Ppow2, b);
        lds_barrier();
        finalize_body<E, true>(N, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, b);
        return;
    }
    finalize_fast_body<E>(N, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, b);
}

template <int E, bool FUSED = false>
__device__ __forceinline__ void csr_build_body(int N, const int* __restrict__ counts, char* ws, gnms_ws_layout L, const int b, const unsigned tag = 0u) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* cnt = reinterpret_cast<int*>(smem);                                         // members per leader rank
    int* fill = cnt + E * 1024;
    int* seg = fill + E * 1024;                                                      // the runs, members in arrival order
    int* startL = seg + E * 1024;
    __shared__ int s_cnt[4];                                                         // [0] verdict, [1] multi-member heads, [2] big heads
    __shared__ int wave_tot[16];
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {
        const u64* g = I.gran + (size_t)16 * 32 + kGranVerdict;
        u64 v = gran_load(g);
        int verdict;
        if constexpr (FUSED) {                                                       // (strong granules: the launch zeroed nothing)
            const u64 v1 = strong_gran(tag, kSlotVerdict + 1u), v2 = strong_gran(tag, kSlotVerdict + 2u);
            while (v != v1 && v != v2) { __builtin_amdgcn_s_sleep(8); v = gran_load(g); }
            verdict = v == v1 ? 1 : 2;
        } else {
            const u64 epoch = (u64)(unsigned)I.misc[8] << 32;
            while ((v & 0xffffffff00000000ull) != epoch) { __builtin_amdgcn_s_sleep(8); v = gran_load(g); }
            verdict = (int)(v & 3ull);
        }
        if (lane == 0) { s_cnt[0] = verdict; s_cnt[1] = 0; s_cnt[2] = 0; }
    }
    for (int i = tid; i < E * 1024; i += 1024) { cnt[i] = 0; fill[i] = 0; }
    __syncthreads();
    if (s_cnt[0] != 1) return;                                                       // K5 proper ran (or runs): it builds the runs itself
    int hd[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = e * 1024 + tid;
        hd[e] = (k < n) ? rem_load(I.head, k) : -1;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) if (hd[e] >= 0) atomicAdd(&cnt[hd[e]], 1);
    __syncthreads();
    // run starts: exclusive scan of the counts over the ranks (thread t owns ranks t * E .. t * E + E - 1)
    int c[E], sum = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) { c[e] = cnt[tid * E + e]; sum += c[e]; }
    const int inc = (int)gnms_add_scan32((unsigned)sum);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) { const int v = wave_tot[w]; if (w < wave) base += v; total += v; }
    int run = base + inc - sum;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        startL[i] = run;
        if (i < n) { I.gstart[i] = run; I.glen[i] = c[e]; }
        // heads of multi-member groups -> hlist (any order), the big ones once more from the end (bwd_masked_*, solve_groups_kernel)
        const bool multi = i < n && c[e] > 1;
        const unsigned long long bm = __ballot(multi);
        if (bm) {
            int hb = 0;
            if (lane == __builtin_ctzll(bm)) hb = atomicAdd(&s_cnt[1], __builtin_popcountll(bm));
            hb = __builtin_amdgcn_readlane(hb, __builtin_ctzll(bm));
            if (multi) I.hlist[hb + __builtin_popcountll(bm & ((1ull << lane) - 1ull))] = i;
        }
        if (i < n && c[e] > kBigGroupList) I.hlist[N - 1 - atomicAdd(&s_cnt[2], 1)] = i;
        run += c[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) if (hd[e] >= 0) seg[startL[hd[e]] + atomicAdd(&fill[hd[e]], 1)] = e * 1024 + tid;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = e * 1024 + tid;
        if (k >= N) continue;
        int pos = -1;
        if (hd[e] >= 0) {
            const int st = startL[hd[e]], len = cnt[hd[e]];
            pos = 0;
            for (int i = 0; i < len; ++i) pos += seg[st + i] < k ? 1 : 0;            // rank order inside the run, whatever order the atomics ran in
            I.gsorted[st + pos] = k;
        }
        I.gpos[k] = pos;
        if (k >= n) { I.head[k] = -1; I.glen[k] = 0; }                               // padding ranks: in no group
    }
    for (int i = total + tid; i < n; i += 1024) I.gsorted[i] = -1;                   // (behind the members, as K5's sort leaves it)
    if (tid == 0) { I.misc[1] = s_cnt[1]; I.misc[4] = s_cnt[2]; }
}

// ------------------------------------------------------------------------------------------------
// K3..K6 in ONE launch (masked groups): leaders -> attribution -> groups + rescoring -> finalize.  Chain workgroups per image: `spw` for
// the scan (the last one goes on with K4..K6), plus -- `fast` -- one for csr_build_body.
// Ppow2 >= 1024 (smaller images pad their keys).  dynamic LDS = max(leaders table (+ cnt), Ppow2 * 16).
// ------------------------------------------------------------------------------------------------
// is the fast tail possible for this launch? (decided on the host; the kernels get it as `fast`)
__host__ __device__ inline bool fast_tail_ok(int N, const gnms_params& P, int sym_arg, int chain_cap = 0) {
    return N <= 4096 && P.group_boxes && P.mask_group_boxes && !P.presorted && sym_arg != 0 && chain_cap == 0;
}

template <int E, int SRC>
__global__ __launch_bounds__(1024) void tail_kernel(const float* __restrict__ src, int N, long ld, const int* __restrict__ counts, gnms_params P,
                                                    char* ws, gnms_ws_layout L, int Ppow2, float* __restrict__ prob,
                                                    long long* __restrict__ valid, long long* __restrict__ invalid, int* __restrict__ nvalid,
                                                    int* __restrict__ ninvalid, int sym_arg, int B, int spw, int fast, int nchk) {
    int b;
    GNMS_TINIT();
    if constexpr (E <= 4) {
        if (fast) {
            // grid: [nchk symmetry checkers (sym_arg 3)] [B * spw scan workgroups] [B CSR workgroups]
            const int bx = (int)blockIdx.x - nchk;
            if (bx < 0) { wsym_check_in_launch(N, counts, ws, L, B, (int)blockIdx.x, nchk); return; }
            if (bx >= B * spw) {                                         // the image's CSR workgroup
                b = bx - B * spw;
                const int* misc = img_ptrs(ws, L, b).misc;
                const int symb = sym_arg == 2 ? (misc[3] == 0 ? 1 : 0) : (sym_arg == 3 ? (misc[2] == 0 ? 1 : 0) : sym_arg);
                if (symb) csr_build_body<E>(N, counts, ws, L, b);
                return;
            }
            int last = leaders_chain<SRC>(N, counts, ws, L, B, spw, bx, sym_arg, &b, src, ld, P.nms_threshold, P.temperature, P.pruning_method, Ppow2);
            if (!last) { GNMS_TFLUSH(ws, L, b); return; }
            if (last != 3 && sym_arg == 3) {                             // the scan ran on trust: the checkers' verdict before anything is final
                ImgPtrs I = img_ptrs(ws, L, b);
                int asym = 0;
                if (threadIdx.x < 64) {
                    while (__hip_atomic_load(I.misc + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nchk) __builtin_amdgcn_s_sleep(4);
                    asym = __hip_atomic_load(I.misc + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                }
                if (__syncthreads_or(asym)) {                            // not symmetric after all: the general scan, K4..K6 proper; nothing for the CSR workgroup
                    if (threadIdx.x == 0) gran_store(I.gran + (size_t)16 * 32 + kGranVerdict, ((u64)(unsigned)I.misc[8] << 32) | 2ull);
                    leaders_body(N, counts, ws, L, b);
                    last = 3;
                }
            }
            if (last != 3) { fast_final_body<E, SRC>(src, N, ld, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, b, last); GNMS_TFLUSH(ws, L, b); return; }
            __syncthreads();                                             // general scan: K4..K6 as before
            attribute_image<SRC>(src, ld, N, counts, P.nms_threshold, ws, L, b, 0);
            __syncthreads();
            groups_body<E, SRC>(src, N, ld, counts, P, ws, L, Ppow2, b);
            __syncthreads();
            finalize_body<E>(N, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, b);
            return;
        }
    }
    if (!leaders_chain(N, counts, ws, L, B, spw, (int)blockIdx.x, sym_arg, &b)) return;   // (the image's other scan workgroups)
    const int sym = sym_arg == 2 ? (img_ptrs(ws, L, b).misc[3] == 0 ? 1 : 0) : sym_arg;   // (2: wsym_check_kernel's verdict for this image)
    __syncthreads();
    if (E <= 4 && sym && P.mask_group_boxes) {                       // the scan has attributed: K4's rest rides in K5, K6 starts from LDS
        if constexpr (E <= 4) {
            groups_body<E, SRC, true>(src, N, ld, counts, P, ws, L, Ppow2, b);
            lds_barrier();
            finalize_body<E, true>(N, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, b);
        }
        return;
    }
    attribute_image<SRC>(src, ld, N, counts, P.nms_threshold, ws, L, b, sym);
    __syncthreads();
    groups_body<E, SRC>(src, N, ld, counts, P, ws, L, Ppow2, b);
    __syncthreads();
    finalize_body<E>(N, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, b);
}

}  // namespace
}  // namespace gnms
