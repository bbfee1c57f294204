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
#include "gnms_common.h"

namespace gnms {
namespace {   // internal linkage: the header is included by several translation units

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------------------
// in-LDS bitonic sort of P (power of two) 64-bit keys, ascending, by the whole workgroup
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bitonic_sort_lds(u64* keys, int P) {
    const int T = blockDim.x;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const u64 a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int lower_bound_lds(const u64* keys, int n, u64 v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct ImgPtrs {
    int* order; float* sscore; int* rem; int* head; int* gpos; int* gsorted; int* gstart; int* glen;
    float* plead; float* pre; float* r2; int* sidx; float* xsol; float* gx; int* leadc; int* leadr; u64* leadw; int* leadpfx;
    int* misc; u64* W;
};

__device__ __host__ __forceinline__ ImgPtrs img_ptrs(char* ws, const gnms_ws_layout& L, int b) {
    char* p = ws + (size_t)b * L.per_image;
    ImgPtrs I;
    I.order = (int*)(p + L.off_order); I.sscore = (float*)(p + L.off_sscore); I.rem = (int*)(p + L.off_rem);
    I.head = (int*)(p + L.off_head); I.gpos = (int*)(p + L.off_gpos); I.gsorted = (int*)(p + L.off_gsorted);
    I.gstart = (int*)(p + L.off_gstart); I.glen = (int*)(p + L.off_glen); I.plead = (float*)(p + L.off_plead);
    I.pre = (float*)(p + L.off_pre); I.r2 = (float*)(p + L.off_r2); I.sidx = (int*)(p + L.off_sidx);
    I.xsol = (float*)(p + L.off_xsol); I.gx = (float*)(p + L.off_gx); I.leadc = (int*)(p + L.off_leadc); I.leadr = (int*)(p + L.off_leadr);
    I.leadw = (u64*)(p + L.off_leadw); I.leadpfx = (int*)(p + L.off_leadpfx); I.misc = (int*)(p + L.off_misc);
    I.W = (u64*)(p + L.off_W);
    return I;
}

// ------------------------------------------------------------------------------------------------
// K1: stable descending argsort of the scores (lib/groomed_nms.py:41; get_groups :213)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sort_scores_kernel(const float* __restrict__ scores, int N, const int* __restrict__ counts,
                                                           char* ws, gnms_ws_layout L, int P, long long* __restrict__ order_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    const int b = blockIdx.x;
    const int n = counts ? counts[b] : N;
    const float* s = scores + (size_t)b * N;
    ImgPtrs I = img_ptrs(ws, L, b);
    for (int i = threadIdx.x; i < P; i += blockDim.x)
        keys[i] = (i < n) ? (((u64)gnms_desc_key(s[i]) << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    bitonic_sort_lds(keys, P);
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        int idx = k;                  // padding ranks map to themselves
        float v = 0.0f;
        if (k < n) { idx = (int)(keys[k] & 0xffffffffu); v = s[idx]; }
        I.order[k] = idx;
        I.sscore[k] = v;
        if (order_out) order_out[(size_t)b * N + k] = idx;
    }
    if (threadIdx.x < 8) I.misc[threadIdx.x] = 0;
}

// ------------------------------------------------------------------------------------------------
// K2: threshold bit matrix.  One wave = 64 rank-rows x 256 input columns.  Rows order[64*kb + r] are
// contiguous 4N-byte streams whatever the permutation, so the row gather is free; the column side
// is never permuted: lane t accumulates, for each of its 4 columns c, the 64-bit word
//     W[kb][c] = sum_r  !(iou[order[64 kb + r]][c] <= thr) << r
// i.e. column c of the thresholded matrix with its bits already in RANK space.  16 x 1-KiB loads are
// kept in flight per wave.
// ------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void bitmask_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                      float thr, char* ws, gnms_ws_layout L) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int kb = blockIdx.y;
    const int n = counts ? counts[b] : N;
    const int k0 = kb * 64;
    const int c0 = (blockIdx.x * 4 + wave) * 256;
    if (k0 >= n || c0 >= n) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    const float* m = iou + (size_t)b * N * ld;

    const int myrank = k0 + lane;
    const int myrow = (myrank < n) ? I.order[myrank] : I.order[k0];      // clamp to a valid row; bits masked below
    const int nrows = min(64, n - k0);
    const u64 rowmask = (nrows >= 64) ? ~0ull : ((1ull << nrows) - 1ull);

    int col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) col[j] = VEC ? (c0 + 4 * lane + j) : (c0 + lane + 64 * j);
    // VEC reads 16 B at col[0]; legal while col[0] < ld (ld % 4 == 0).  Columns >= n produce words nobody reads.
    const bool active = VEC ? (col[0] < L.NC && col[0] + 3 < ld) : true;

    unsigned lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int rb = 0; rb < 64; rb += 16) {
        float v[16][4];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int row = __builtin_amdgcn_readlane(myrow, rb + u);
            const float* p = m + (size_t)row * ld;
            if (VEC) {
                float4 t = active ? *reinterpret_cast<const float4*>(p + col[0]) : make_float4(0.f, 0.f, 0.f, 0.f);
                v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[u][j] = (col[j] < n) ? p[col[j]] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = rb + u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool notlow = !(v[u][j] <= thr);                       // lib/groomed_nms.py:250 (NaN -> removed)
                if (r < 32) lo[j] |= notlow ? (1u << r) : 0u; else hi[j] |= notlow ? (1u << (r - 32)) : 0u;
            }
        }
    }
    u64* Wk = I.W + (size_t)kb * L.NC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (col[j] < L.NC) Wk[col[j]] = (((u64)hi[j] << 32) | lo[j]) & rowmask;
    }
}

// ------------------------------------------------------------------------------------------------
// K3: leaders (= the boxes classical greedy NMS keeps).  For rank block kb the word
//   removed = OR over all earlier leaders L of W[kb][order[L]]
// is pulled cooperatively (leaders' input indices sit in LDS), then wave 0 resolves the 64 ranks of
// the block against each other on the diagonal words, visiting only the leaders (s_ff1 on ~removed).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 uniform64(u64 v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void leaders_kernel(int N, const int* __restrict__ counts, char* ws, gnms_ws_layout L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* leadc_s = reinterpret_cast<int*>(smem);                 // [N]
    __shared__ u64 part[4];
    __shared__ int nlead_s;
    const int b = blockIdx.x;
    const int n = counts ? counts[b] : N;
    ImgPtrs I = img_ptrs(ws, L, b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = (n + 63) >> 6;
    if (threadIdx.x == 0) { nlead_s = 0; I.leadpfx[0] = 0; }
    __syncthreads();
    for (int kb = 0; kb < nb; ++kb) {
        const u64* slab = I.W + (size_t)kb * L.NC;
        const int nlead = nlead_s;
        u64 acc = 0;
        for (int t = threadIdx.x; t < nlead; t += 256) acc |= slab[leadc_s[t]];
        acc = gnms_wave_or(acc);
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (wave == 0) {
            const int k0 = kb << 6;
            const int nrows = min(64, n - k0);
            const int myc = (lane < nrows) ? I.order[k0 + lane] : 0;
            const u64 d = (lane < nrows) ? slab[myc] : 0ull;      // who rank k0+lane would remove inside this block
            u64 cur = part[0] | part[1] | part[2] | part[3];
            if (nrows < 64) cur |= ~((1ull << nrows) - 1ull);      // ranks >= n never lead
            cur = uniform64(cur);                                  // wave-uniform: keep the resolve loop on the scalar unit
            u64 leaders = 0;
            while (~cur != 0ull) {
                const int p = __builtin_ctzll(~cur);
                const unsigned dl = __builtin_amdgcn_readlane((unsigned)(d & 0xffffffffu), p);
                const unsigned dh = __builtin_amdgcn_readlane((unsigned)(d >> 32), p);
                leaders |= 1ull << p;
                cur |= (((u64)dh << 32) | dl) | (1ull << p);       // a leader always leaves `remaining`
            }
            const int before = __builtin_popcountll(leaders & ((1ull << lane) - 1ull));
            if ((leaders >> lane) & 1ull) {
                leadc_s[nlead + before] = myc;
                I.leadc[nlead + before] = myc;
                I.leadr[nlead + before] = k0 + lane;
            }
            if (lane == 0) {
                I.leadw[kb] = leaders;
                const int tot = nlead + __builtin_popcountll(leaders);
                I.leadpfx[kb + 1] = tot;
                nlead_s = tot;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) I.misc[0] = nlead_s;
}

// ------------------------------------------------------------------------------------------------
// K4: attribution.  One wave per rank block: walk the leaders with rank < 64(kb+1) in order, 64 per step;
// an exclusive OR-scan across lanes tells each leader which bits it is the FIRST to claim.
//   rem[k] = rank of the leader that removed rank k (k itself for a leader).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 wave_or_exclusive_scan(u64 v, int lane) {
    u64 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u64 t = __shfl_up(inc, off, 64);
        if (lane >= off) inc |= t;
    }
    u64 ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 0ull : ex;
}

__global__ __launch_bounds__(64) void attribute_kernel(int N, const int* __restrict__ counts, char* ws, gnms_ws_layout L) {
    const int b = blockIdx.y, kb = blockIdx.x;
    const int n = counts ? counts[b] : N;
    const int k0 = kb << 6;
    if (k0 >= n) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    const int lane = threadIdx.x;
    const int nrows = min(64, n - k0);
    const u64 want = (nrows >= 64) ? ~0ull : ((1ull << nrows) - 1ull);
    const u64* slab = I.W + (size_t)kb * L.NC;
    const int nl = I.leadpfx[kb + 1];                 // leaders with rank < k0 + 64
    u64 acc = 0;
    for (int base = 0; base < nl && (acc & want) != want; base += 64) {
        const int t = base + lane;
        u64 w = 0;
        int lr = -1;
        if (t < nl) {
            lr = I.leadr[t];
            w = slab[I.leadc[t]];
            if (lr >= k0) w |= 1ull << (lr - k0);     // the leader's own slot
            w &= want;
        }
        const u64 ex = wave_or_exclusive_scan(w, lane);
        u64 mine = w & ~(acc | ex);
        while (mine) {
            const int bit = __builtin_ctzll(mine);
            I.rem[k0 + bit] = lr;
            mine &= mine - 1;
        }
        acc |= gnms_wave_or(w);
    }
}

// ------------------------------------------------------------------------------------------------
// K5: groups.  member(k) = iou[order[k]][order[rem[k]]] > thr (strict, :249); sort (leader, rank) in LDS;
// runs of equal leader are the groups, their first group_size+1 entries survive (:253-255), the first
// entry is the column the mask keeps (:99).  MASKED: the default rescoring (:95-105,:111) is fused:
//   pre_k = s_k - prune(iou[k][head]) * s_head      (I - P restricted to the head column)
// Arrays head/gpos/gstart/glen/gsorted/plead are indexed by rank; pre is indexed by NMS position q
// (q = rank for hard sort, q = input index when presorted).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void groups_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                      gnms_params P, char* ws, gnms_ws_layout L, int Ppow2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    const int b = blockIdx.x;
    const int n = counts ? counts[b] : N;
    ImgPtrs I = img_ptrs(ws, L, b);
    const float* m = iou + (size_t)b * N * ld;
    const float thr = P.nms_threshold;
    for (int k = threadIdx.x; k < Ppow2; k += blockDim.x) {
        u64 key = ~0ull;
        if (k < n) {
            const int lr = I.rem[k];
            const float v = m[(size_t)I.order[k] * ld + I.order[lr]];
            if (v > thr) key = ((u64)(unsigned)lr << 32) | (unsigned)k;
        }
        keys[k] = key;
        if (k < N) { I.head[k] = -1; I.gpos[k] = -1; I.glen[k] = 0; I.gstart[k] = 0; I.plead[k] = 0.0f; }
    }
    __syncthreads();
    bitonic_sort_lds(keys, Ppow2);
    const long long cap = (long long)P.group_size + 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const u64 key = keys[i];
        if (key == ~0ull) { I.gsorted[i] = -1; continue; }
        const unsigned lr = (unsigned)(key >> 32);
        const int k = (int)(key & 0xffffffffu);
        const int start = lower_bound_lds(keys, n, (u64)lr << 32);
        const int end = lower_bound_lds(keys, n, ((u64)lr + 1ull) << 32);
        const long long pos = i - start;
        I.gsorted[i] = k;
        if (pos < cap) {
            I.head[k] = (int)(keys[start] & 0xffffffffu);
            I.gpos[k] = (int)pos;
            I.gstart[k] = start;
            const long long len = end - start;
            I.glen[k] = (int)(len < cap ? len : cap);
        }
        if (pos == 0) atomicAdd(&I.misc[1], 1);
    }
    __syncthreads();
    if (!P.mask_group_boxes) return;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        float pre = 0.0f, pl = 0.0f;
        const int q = P.presorted ? I.order[k] : k;
        if (k < n) {
            const int h = I.head[k];
            if (h == k) {
                pre = I.sscore[k];
            } else if (h >= 0) {
                const int ck = I.order[k], ch = I.order[h];
                const bool tril = P.presorted ? (ch < ck) : true;       // torch.tril in NMS order (:72)
                if (tril) pl = gnms_prune(m[(size_t)ck * ld + ch], thr, P.temperature, P.pruning_method);
                pre = I.sscore[k] - pl * I.sscore[h];
            }
        }
        I.plead[k] = pl;
        I.pre[q] = pre;
    }
}

// ------------------------------------------------------------------------------------------------
// K6: finalize (lib/groomed_nms.py:111-129).  r2 = clamp(pre,0,1); r = r2 with (< valid_thr) zeroed;
// stable descending sort of r; valid / invalid lists of INPUT indices; the prob vector in the order
// the reference returns it.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void finalize_kernel(int N, const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L,
                                                        int Ppow2, float* __restrict__ prob, long long* __restrict__ valid,
                                                        long long* __restrict__ invalid, int* __restrict__ nvalid,
                                                        int* __restrict__ ninvalid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    const int b = blockIdx.x;
    const int n = counts ? counts[b] : N;
    ImgPtrs I = img_ptrs(ws, L, b);
    const float vthr = P.valid_box_prob_threshold;
    for (int q = threadIdx.x; q < Ppow2; q += blockDim.x) {
        u64 key = ~0ull;
        if (q < n) {
            const float pre = I.pre[q];
            const float r2 = pre < 0.0f ? 0.0f : (pre > 1.0f ? 1.0f : pre);      // torch.clamp keeps NaN
            const float r = (r2 < vthr) ? 0.0f : r2;                            // :115
            I.r2[q] = r2;
            key = ((u64)gnms_desc_key(r) << 32) | (unsigned)q;
        }
        keys[q] = key;
    }
    __syncthreads();
    bitonic_sort_lds(keys, Ppow2);
    // NaN keys are 0; valid  <=> r >= vthr <=> key32 <= desc_key(vthr)
    const int n_nan = lower_bound_lds(keys, n, 1ull << 32);
    int n_ge = (vthr != vthr) ? n_nan : lower_bound_lds(keys, n, ((u64)gnms_desc_key(vthr) + 1ull) << 32);
    if (n_ge < n_nan) n_ge = n_nan;
    const int nv = n_ge - n_nan;
    const int ni = (vthr != vthr) ? 0 : (n - n_ge);
    float* pb = prob + (size_t)b * N;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        if (j >= n) {
            pb[j] = 0.0f;
            I.sidx[j] = j;
            continue;
        }
        const int q = (int)(keys[j] & 0xffffffffu);
        I.sidx[j] = q;
        const int inp = P.presorted ? q : I.order[q];
        if (j >= n_nan && j < n_ge) { if (valid) valid[(size_t)b * N + (j - n_nan)] = inp; }
        else if (j >= n_ge && ni > 0) { if (invalid) invalid[(size_t)b * N + (j - n_ge)] = inp; }
        const float r2q = I.r2[q];
        if (P.return_sorted_prob) pb[j] = (r2q < vthr) ? 0.0f : r2q;                       // :117
        const float r2j = I.r2[j];
        if (!P.return_sorted_prob) pb[j] = P.group_boxes ? r2j : ((r2j < vthr) ? 0.0f : r2j);  // :124-127
    }
    if (threadIdx.x == 0) {
        if (nvalid) nvalid[b] = nv;
        if (ninvalid) ninvalid[b] = ni;
    }
}

}  // namespace
}  // namespace gnms
