// nms_solve_kernels.h -- the non-default rescoring modes of GrooMeD-NMS (gfx950).
//
//   unmasked groups  lib/groomed_nms.py:107  T = inverse(I_g + P_g),  g <= group_size+1
//   ungrouped        lib/groomed_nms.py:110  inverse(I + P), N x N
// I + P is UNIT LOWER TRIANGULAR in NMS order (P = tril(prune(iou)) with a zero diagonal, :71-73), so
// "multiply by the inverse" is a forward substitution  x_i = s_i - sum_{j<i} P_ij x_j  and its backward
// a substitution with the transpose; nothing is inverted, no N x N matrix is built.
//   dL/ds = y,   (I+P)^T y = gx        dL/dP_ij = -y_i x_j  (j < i)
// NMS order: position q <-> input index cq(q) = presorted ? q : order[q].
#pragma once
#include "nms_backward_kernels.h"

namespace gnms {

constexpr int kGroupTileCap = 128;    // groups up to this size keep P_g in LDS (128*129*4 = 66 KB)
constexpr int kGroupMaxMembers = 2048;

// ------------------------------------------------------------------------------------------------
// unmasked groups: one wave per group (launched per rank; only heads work).
// dynamic LDS: int sc[G], int sq[G], float acc[G], float Pl[tile*(tile+1)]   with G = kGroupMaxMembers
// ------------------------------------------------------------------------------------------------
template <bool BWD, bool BOXES>
__global__ __launch_bounds__(64) void solve_groups_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                          gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ grad_scores,
                                                          float* __restrict__ grad_iou) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sc = reinterpret_cast<int*>(smem);
    int* sq = sc + kGroupMaxMembers;
    float* acc = reinterpret_cast<float*>(sq + kGroupMaxMembers);
    float* Pl = acc + kGroupMaxMembers;
    const int b = blockIdx.y, k = blockIdx.x;
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int lane = threadIdx.x;
    if (k >= n) {
        if (lane == 0) { if (BWD) grad_scores[(size_t)b * N + k] = 0.0f; else I.pre[k] = 0.0f; }
        return;
    }
    const int h = I.head[k];
    if (h < 0) {   // in no group: zero row of M
        if (lane == 0) {
            const int c = I.order[k];
            if (BWD) grad_scores[(size_t)b * N + c] = 0.0f; else I.pre[P.presorted ? c : k] = 0.0f;
        }
        return;
    }
    if (h != k) return;
    const float* m = iou + (BOXES ? (size_t)b * N * 4 : (size_t)b * N * ld);   // BOXES: `iou` holds the boxes [B][N][4]
    float* gi = (BWD && grad_iou && !BOXES) ? grad_iou + (size_t)b * N * ld : nullptr;
    const int g = I.glen[k], start = I.gstart[k];
    const bool tiled = g <= kGroupTileCap;
    const int ts = g + 1;   // padded tile stride

    for (int t = lane; t < g; t += 64) {
        const int mk = I.gsorted[start + t];
        const int c = I.order[mk];
        const int q = P.presorted ? c : mk;
        int slot = t;
        if (P.presorted) {      // NMS order inside the group = ascending input index
            slot = 0;
            for (int u = 0; u < g; ++u) slot += (I.order[I.gsorted[start + u]] < c) ? 1 : 0;
        }
        sc[slot] = c; sq[slot] = q;
        acc[slot] = BWD ? I.gx[q] : I.sscore[mk];
    }
    __syncthreads();
    if (tiled) {
        for (int e = lane; e < g * g; e += 64) {
            const int a = e / g, bb = e - a * g;
            Pl[a * ts + bb] = (bb < a) ? gnms_prune(overlap_at<BOXES>(m, ld, sc[a], sc[bb]), P.nms_threshold, P.temperature, P.pruning_method) : 0.0f;
        }
        __syncthreads();
    }
    auto Pab = [&](int a, int bb) -> float {
        return tiled ? Pl[a * ts + bb] : gnms_prune(overlap_at<BOXES>(m, ld, sc[a], sc[bb]), P.nms_threshold, P.temperature, P.pruning_method);
    };
    if (!BWD) {
        for (int bb = 0; bb < g - 1; ++bb) {
            const float xb = acc[bb];
            for (int a = bb + 1 + lane; a < g; a += 64) acc[a] -= Pab(a, bb) * xb;
            __syncthreads();
        }
        for (int t = lane; t < g; t += 64) I.pre[sq[t]] = acc[t];
    } else {
        for (int bb = g - 1; bb > 0; --bb) {
            const float yb = acc[bb];
            for (int a = lane; a < bb; a += 64) acc[a] -= Pab(bb, a) * yb;
            __syncthreads();
        }
        for (int t = lane; t < g; t += 64) grad_scores[(size_t)b * N + sc[t]] = acc[t];
        if (gi) {
            for (int e = lane; e < g * g; e += 64) {
                const int a = e / g, bb = e - a * g;
                if (bb >= a) continue;
                const size_t off = (size_t)sc[a] * ld + sc[bb];
                const float d = gnms_prune_grad(m[off], P.nms_threshold, P.temperature, P.pruning_method);
                gi[off] = (-(acc[a] * I.pre[sq[bb]])) * d;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ungrouped: one workgroup (1024 threads) per image, blocked substitution, ONE pass over the matrix.
// dynamic LDS: float xin[N] (x or the running transpose accumulator, by input column), float T[64*65],
//              float xb[64]
// `rem` (unused by this mode) carries pos_of[input index] = NMS position.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ungrouped_prepare_kernel(int N, const int* __restrict__ counts, gnms_params P, char* ws,
                                                                 gnms_ws_layout L) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    I.rem[P.presorted ? k : I.order[k]] = k;       // order is a permutation of [0,N) (identity on the padding)
}

__global__ __launch_bounds__(1024) void ungrouped_forward_kernel(const float* __restrict__ iou, const float* __restrict__ scores, int N,
                                                                 long ld, const int* __restrict__ counts, gnms_params P, char* ws,
                                                                 gnms_ws_layout L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xin = reinterpret_cast<float*>(smem);
    float* T = xin + ((N + 3) & ~3);
    float* xb = T + 64 * 65;
    const int b = blockIdx.x;
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const float* m = iou + (size_t)b * N * ld;
    const float* s = scores + (size_t)b * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < N; c += blockDim.x) { xin[c] = 0.0f; if (c >= n) I.pre[c] = 0.0f; }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int rows = min(64, n - i0);
        // (1) contributions of every solved column (xin is still 0 for unsolved ones)
        for (int a = wave; a < rows; a += 16) {
            const int ci = P.presorted ? (i0 + a) : I.order[i0 + a];
            const float* row = m + (size_t)ci * ld;
            float sum = 0.0f;
            for (int c = lane; c < n; c += 64) {
                const float x = xin[c];
                if (x != 0.0f) sum += gnms_prune(row[c], P.nms_threshold, P.temperature, P.pruning_method) * x;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
            if (lane == 0) xb[a] = s[ci] - sum;
        }
        // (2) diagonal tile T[a][bb], bb < a
        for (int e = threadIdx.x; e < rows * rows; e += blockDim.x) {
            const int a = e / rows, bb = e - a * rows;
            if (bb < a) {
                const int ca = P.presorted ? (i0 + a) : I.order[i0 + a];
                const int cb = P.presorted ? (i0 + bb) : I.order[i0 + bb];
                T[a * 65 + bb] = gnms_prune(m[(size_t)ca * ld + cb], P.nms_threshold, P.temperature, P.pruning_method);
            }
        }
        __syncthreads();
        if (wave == 0) {
            float x = (lane < rows) ? xb[lane] : 0.0f;
            for (int bb = 0; bb < rows - 1; ++bb) {
                const float xbb = __shfl(x, bb, 64);
                if (lane > bb && lane < rows) x -= T[lane * 65 + bb] * xbb;
            }
            if (lane < rows) {
                const int ci = P.presorted ? (i0 + lane) : I.order[i0 + lane];
                xin[ci] = x;
                I.pre[i0 + lane] = x;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void ungrouped_backward_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                                  gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ grad_scores,
                                                                  float* __restrict__ grad_iou) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ain = reinterpret_cast<float*>(smem);        // sum_{solved i} P(i, c) y_i, by input column c
    float* T = ain + ((N + 3) & ~3);
    float* yb = T + 64 * 65;
    const int b = blockIdx.x;
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const float* m = iou + (size_t)b * N * ld;
    float* gi = grad_iou ? grad_iou + (size_t)b * N * ld : nullptr;
    float* gs = grad_scores + (size_t)b * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < N; c += blockDim.x) { ain[c] = 0.0f; if (c >= n) gs[c] = 0.0f; }
    __syncthreads();
    const int nblk = (n + 63) >> 6;
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int i0 = blk << 6;
        const int rows = min(64, n - i0);
        for (int e = threadIdx.x; e < rows * rows; e += blockDim.x) {
            const int a = e / rows, bb = e - a * rows;
            if (bb < a) {
                const int ca = P.presorted ? (i0 + a) : I.order[i0 + a];
                const int cb = P.presorted ? (i0 + bb) : I.order[i0 + bb];
                T[a * 65 + bb] = gnms_prune(m[(size_t)ca * ld + cb], P.nms_threshold, P.temperature, P.pruning_method);
            }
        }
        __syncthreads();
        if (wave == 0) {
            int ci = 0;
            float y = 0.0f;
            if (lane < rows) {
                ci = P.presorted ? (i0 + lane) : I.order[i0 + lane];
                y = I.gx[i0 + lane] - ain[ci];
            }
            for (int a = rows - 1; a > 0; --a) {
                const float ya = __shfl(y, a, 64);
                if (lane < a) y -= T[a * 65 + lane] * ya;
            }
            if (lane < rows) { yb[lane] = y; gs[ci] = y; }
        }
        __syncthreads();
        // right-looking update: every column owned by one thread; rows of the block stream through
        for (int c = threadIdx.x; c < n; c += blockDim.x) {
            const int pc = I.rem[c];                        // NMS position of input column c
            const float xc = I.pre[pc];
            float a_c = ain[c];
            for (int a = 0; a < rows; ++a) {
                const int ci = P.presorted ? (i0 + a) : I.order[i0 + a];
                const size_t off = (size_t)ci * ld + c;
                const float v = m[off];
                const float y = yb[a];
                const bool live = pc < i0 + a;              // strictly lower triangle in NMS order
                if (live && pc < i0) a_c += gnms_prune(v, P.nms_threshold, P.temperature, P.pruning_method) * y;
                if (gi) gi[off] = live ? (-(y * xc)) * gnms_prune_grad(v, P.nms_threshold, P.temperature, P.pruning_method) : 0.0f;
            }
            ain[c] = a_c;
        }
        __syncthreads();
    }
}

}  // namespace gnms
