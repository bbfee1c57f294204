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
// unmasked groups: grid (solve_groups_wgs(B), B) x 1024 threads, one workgroup per CU.  Every workgroup first settles its share of the
// trivial boxes (padding, boxes in no group, groups of one); then
//   * SMALL groups (<= 32 members from the boxes, <= 16 from the matrix, where every step is a dependent load): ONE WAVE per group,
//     walking the head list hlist with stride (workgroups x 16).  Lane a holds member a; step bb of the substitution broadcasts x_bb
//     (and member bb's box) with v_readlane, P_ab of step bb + 1 is computed while step bb's update is in flight: no LDS, no barrier.
//     NMS inputs have ~1 000 groups of 2-4 members per image (uniform boxes, N = 4096): with one 256-thread workgroup per group, a
//     barrier per step and five dependent loads in front of each group this kernel took 53 us forward and 53 backward (rounds 1-4a);
//     now 9 / 8.
//   * BIG groups (the second list groups_body keeps at the end of hlist): one workgroup per group -- all 1024 threads fill the
//     triangular tile P_g in LDS (up to 128 members: the default cap is 101), then wave 0 alone substitutes out of LDS, two members per
//     lane, the next step's entries read ahead; beyond 128 members entry by entry with a barrier per step, as before.
// (History: the first version launched one 64-thread workgroup per BOX with 90 KB of LDS each: 32768 workgroups, ~200 us of dispatch.)
// Same operations in the same order everywhere: x_a -= P_ab * x_b for ascending b (forward), y_a -= P_ba * y_b for descending b (backward).
// dynamic LDS: int sc[G], int sq[G], float acc[G], float Pl[tile*(tile+1)]   with G = kGroupMaxMembers
// ------------------------------------------------------------------------------------------------
constexpr size_t kSolveGroupsLds = (size_t)kGroupMaxMembers * 12 + (size_t)kGroupTileCap * (kGroupTileCap + 1) * 4;
static inline int solve_groups_wgs(int B, int cus) { const int w = cus / (B > 0 ? B : 1); return w < 16 ? 16 : (w > 256 ? 256 : w); }

template <bool BWD, bool BOXES>
__global__ __launch_bounds__(1024) void solve_groups_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                            gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ grad_scores,
                                                            float* __restrict__ grad_iou) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sc = reinterpret_cast<int*>(smem);
    int* sq = sc + kGroupMaxMembers;
    float* acc = reinterpret_cast<float*>(sq + kGroupMaxMembers);
    float* Pl = acc + kGroupMaxMembers;
    const int b = blockIdx.y;
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = T >> 6;
    float* gs = BWD ? grad_scores + (size_t)b * N : nullptr;
    // ---- trivial boxes ----
    for (int k = blockIdx.x * T + tid; k < N; k += gridDim.x * T) {
        if (k >= n) { if (BWD) gs[k] = 0.0f; else I.pre[k] = 0.0f; continue; }
        const int h = I.head[k];
        const int c = I.order[k];
        const int q = P.presorted ? c : k;
        if (h < 0) { if (BWD) gs[c] = 0.0f; else I.pre[q] = 0.0f; }                       // in no group: zero row of M
        else if (h == k && I.glen[k] == 1) { if (BWD) gs[c] = I.gx[q]; else I.pre[q] = I.sscore[k]; }
    }
    // ---- multi-member groups ----
    const float* m = iou + (BOXES ? (size_t)b * N * 4 : (size_t)b * N * ld);   // BOXES: `iou` holds the boxes [B][N][4]
    float* gi = (BWD && grad_iou && !BOXES) ? grad_iou + (size_t)b * N * ld : nullptr;
    const int nheads = I.misc[1];
    constexpr int kWaveGroup = BOXES ? 32 : 16;                     // (>= kBigGroupList: every group above it is in the second list)
    static_assert(kWaveGroup >= kBigGroupList, "groups between the two thresholds would be solved by nobody");
    const bool wave_path = !P.presorted;                            // (pre-sorted scores re-order the members by input index: workgroup path)
    auto bcf = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    if (wave_path) {
        const float4* bx4 = reinterpret_cast<const float4*>(m);
        for (int hi = blockIdx.x * nwv + wave; hi < nheads; hi += gridDim.x * nwv) {
            const int k = __builtin_amdgcn_readfirstlane(I.hlist[hi]);
            const int g = __builtin_amdgcn_readfirstlane(I.glen[k]), start = __builtin_amdgcn_readfirstlane(I.gstart[k]);
            if (g > kWaveGroup) continue;                               // (wave-uniform)
            const bool on = lane < g;
            const int mk = on ? I.gsorted[start + lane] : 0;            // NMS position of member `lane` (= its rank: scores not pre-sorted)
            const int c = on ? I.order[mk] : 0;
            float x = on ? (BWD ? I.gx[mk] : I.sscore[mk]) : 0.0f;
            float4 ba = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (BOXES && on) ba = bx4[c];
            // P between member `lane` and member bb (fwd: row = lane, column = bb; backward: the transpose)
            auto p_of = [&](int bb, bool fwd) -> float {
                float ov;
                if (BOXES) {
                    const float4 bq = make_float4(bcf(ba.x, bb), bcf(ba.y, bb), bcf(ba.z, bb), bcf(ba.w, bb));
                    ov = fwd ? pair_iou(ba, bq) : pair_iou(bq, ba);
                } else {
                    const int cb = __builtin_amdgcn_readlane(c, bb);
                    ov = fwd ? m[(size_t)c * ld + cb] : m[(size_t)cb * ld + c];
                }
                return gnms_prune(ov, P.nms_threshold, P.temperature, P.pruning_method);
            };
            if (!BWD) {
                float pc = (g > 1 && on && lane > 0) ? p_of(0, true) : 0.0f;
                for (int bb = 0; bb < g - 1; ++bb) {
                    const float xb = bcf(x, bb);
                    const float pn = (bb + 1 < g - 1 && on && lane > bb + 1) ? p_of(bb + 1, true) : 0.0f;
                    if (on && lane > bb) x -= pc * xb;
                    pc = pn;
                }
                if (on) I.pre[mk] = x;
            } else {
                float pc = (g > 1 && lane < g - 1) ? p_of(g - 1, false) : 0.0f;
                for (int bb = g - 1; bb > 0; --bb) {
                    const float yb = bcf(x, bb);
                    const float pn = (bb - 1 > 0 && lane < bb - 1) ? p_of(bb - 1, false) : 0.0f;
                    if (lane < bb) x -= pc * yb;
                    pc = pn;
                }
                if (on) gs[c] = x;
                if (gi) {
                    const float prev = on ? I.pre[mk] : 0.0f;
                    for (int bb = 0; bb < g - 1; ++bb) {
                        const int cb = __builtin_amdgcn_readlane(c, bb);
                        const float pb = bcf(prev, bb);
                        if (on && lane > bb) {
                            const size_t off = (size_t)c * ld + cb;
                            gi[off] = (-(x * pb)) * gnms_prune_grad(m[off], P.nms_threshold, P.temperature, P.pruning_method);
                        }
                    }
                }
            }
        }
    }
    // ---- big groups (every multi-member group when the scores came pre-sorted): one workgroup each ----
    const int nlist = wave_path ? I.misc[4] : nheads;
    for (int hi = blockIdx.x; hi < nlist; hi += gridDim.x) {
        const int k = wave_path ? I.hlist[N - 1 - hi] : I.hlist[hi];
        const int g = I.glen[k], start = I.gstart[k];
        if (wave_path && g <= kWaveGroup) continue;                     // (done above; workgroup-uniform)
        const bool tiled = g <= kGroupTileCap;
        const int ts = g + 1;   // padded tile stride
        __syncthreads();        // the previous group's LDS is consumed
        for (int t = tid; t < g; t += T) {
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
            for (int e = tid; e < g * g; e += T) {
                const int a = e / g, bb = e - a * g;
                Pl[a * ts + bb] = (bb < a) ? gnms_prune(overlap_at<BOXES>(m, ld, sc[a], sc[bb], P.nms_threshold), P.nms_threshold, P.temperature, P.pruning_method) : 0.0f;
            }
            __syncthreads();
            // the substitution out of LDS on ONE wave (members lane and lane + 64), no barrier per step; the next step's entries read ahead
            if (wave == 0) {
                const int a0 = lane, a1 = lane + 64;
                const bool on0 = a0 < g, on1 = a1 < g;
                float x0 = on0 ? acc[a0] : 0.0f, x1 = on1 ? acc[a1] : 0.0f;
                auto x_of = [&](int bb) { return bb >= 64 ? bcf(x1, bb - 64) : bcf(x0, bb); };
                if (!BWD) {
                    float p0 = (g > 1 && on0) ? Pl[a0 * ts] : 0.0f, p1 = (g > 1 && on1) ? Pl[a1 * ts] : 0.0f;
                    for (int bb = 0; bb < g - 1; ++bb) {
                        const float xb = x_of(bb);
                        const bool more = bb + 1 < g - 1;
                        const float n0 = (more && on0) ? Pl[a0 * ts + bb + 1] : 0.0f, n1 = (more && on1) ? Pl[a1 * ts + bb + 1] : 0.0f;
                        if (on0 && a0 > bb) x0 -= p0 * xb;
                        if (on1 && a1 > bb) x1 -= p1 * xb;
                        p0 = n0; p1 = n1;
                    }
                } else {
                    float p0 = (g > 1 && on0) ? Pl[(g - 1) * ts + a0] : 0.0f, p1 = (g > 1 && on1) ? Pl[(g - 1) * ts + a1] : 0.0f;
                    for (int bb = g - 1; bb > 0; --bb) {
                        const float yb = x_of(bb);
                        const bool more = bb - 1 > 0;
                        const float n0 = (more && on0) ? Pl[(bb - 1) * ts + a0] : 0.0f, n1 = (more && on1) ? Pl[(bb - 1) * ts + a1] : 0.0f;
                        if (a0 < bb) x0 -= p0 * yb;
                        if (a1 < bb) x1 -= p1 * yb;
                        p0 = n0; p1 = n1;
                    }
                }
                if (on0) acc[a0] = x0;
                if (on1) acc[a1] = x1;
            }
            __syncthreads();
        }
        auto Pab = [&](int a, int bb) -> float {
            return tiled ? Pl[a * ts + bb] : gnms_prune(overlap_at<BOXES>(m, ld, sc[a], sc[bb], P.nms_threshold), P.nms_threshold, P.temperature, P.pruning_method);
        };
        if (!BWD) {
            if (!tiled) {
                for (int bb = 0; bb < g - 1; ++bb) {
                    const float xb = acc[bb];
                    for (int a = bb + 1 + tid; a < g; a += T) acc[a] -= Pab(a, bb) * xb;
                    __syncthreads();
                }
            }
            for (int t = tid; t < g; t += T) I.pre[sq[t]] = acc[t];
        } else {
            if (!tiled) {
                for (int bb = g - 1; bb > 0; --bb) {
                    const float yb = acc[bb];
                    for (int a = tid; a < bb; a += T) acc[a] -= Pab(bb, a) * yb;
                    __syncthreads();
                }
            }
            for (int t = tid; t < g; t += T) gs[sc[t]] = acc[t];
            if (gi) {
                for (int e = tid; e < g * g; e += T) {
                    const int a = e / g, bb = e - a * g;
                    if (bb >= a) continue;
                    const size_t off = (size_t)sc[a] * ld + sc[bb];
                    const float d = gnms_prune_grad(m[off], P.nms_threshold, P.temperature, P.pruning_method);
                    gi[off] = (-(acc[a] * I.pre[sq[bb]])) * d;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ungrouped mode (group_boxes=False, lib/groomed_nms.py:72-73, 111): x = (I + P)^-1 s with P = tril(f(iou_sorted), -1) is a
// forward substitution over all N boxes, the backward a substitution with the transpose.  Three steps:
//   U1  ungrouped_permute_kernel   Ps[i][j] = f(iou[order[i]][order[j]]) for j < i, into a scratch matrix in NMS-position
//       space (the reference makes the same copy, :48).  One workgroup per row: the input row is read coalesced into LDS,
//       the column permutation is an LDS gather, the stores are coalesced.  4N^2 bytes in, 2N^2 out.
//   U2  ungrouped_solve_forward_kernel   one workgroup per block of 64 positions, ALL blocks in flight: block b streams
//       its 64 x 64b strip of Ps tile by tile (next tile prefetched into registers), and consumes x of block c as soon as
//       block c publishes it (mailbox in global memory, decoupled look-back: a block only waits for lower-numbered blocks,
//       which are dispatched first).  The sequential part is 64 dependent steps inside a 64 x 64 diagonal tile.
//   U3  ungrouped_solve_backward_kernel  the same with the transpose, blocks published from the last to the first.
// `rem` (unused by this mode) carries pos_of[input index] = NMS position; the bit-matrix region `W` holds the two mailboxes;
// `xsol` the backward solution by position.  (The first version did all of this in ONE workgroup per image, row by
// row: 17.4 ms per step at B=8, N=4096.)
// ------------------------------------------------------------------------------------------------
// Mailbox between the blocks of one image: one 64-bit word per position = (1 << 32 | float bits), written and polled with
// single 8-byte agent-scope atomics -- the value validates itself, so a consumer needs ONE memory round trip per block
// (flag + fence + data would be two and an L2 write-back).  Zeroed (= empty) by the prepare kernels.  The words live in the
// bit-matrix region W, which this mode does not use: [0, N) forward, [N, 2N) backward (only touched when there are >= 2 blocks).
__device__ __forceinline__ void mail_put(u64* slot, float v) {
    __hip_atomic_store(slot, (1ull << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float mail_get(const u64* slot) {
    u64 w;
    while (((w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) == 0) __builtin_amdgcn_s_sleep(1);
    return __uint_as_float((unsigned)(w & 0xffffffffu));
}
// row pitch of the scratch matrix: whole 64-column tiles, because the solves load the diagonal tile of the last row block in full
// (columns i0 .. i0+63; the entries past N are never used, but they must lie inside the allocation)
__host__ __device__ inline size_t ungrouped_ld(int N) { return (size_t)((N + 63) & ~63); }
__host__ __device__ inline size_t ungrouped_scratch_bytes(int B, int N) { return (size_t)B * N * ungrouped_ld(N) * sizeof(float); }

__global__ __launch_bounds__(1024) void ungrouped_prepare_kernel(int N, const int* __restrict__ counts, gnms_params P, char* ws,
                                                                 gnms_ws_layout L) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    I.rem[P.presorted ? k : I.order[k]] = k;       // order is a permutation of [0,N) (identity on the padding)
    if (k >= n) I.pre[k] = 0.0f;
    if (L.NB >= 2) I.W[k] = 0ull;                  // forward mailbox
}

__global__ __launch_bounds__(256) void ungrouped_permute_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                                gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ Ps_all) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* rowbuf = reinterpret_cast<float*>(smem);
    const int b = blockIdx.y, i = blockIdx.x;
    const int n = gnms_count(counts, b, N);
    if (i >= n || i == 0) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    const int ri = P.presorted ? i : I.order[i];
    const float* row = iou + ((size_t)b * N + ri) * ld;
    for (int c = threadIdx.x; c < n; c += blockDim.x) rowbuf[c] = row[c];
    __syncthreads();
    float* out = Ps_all + ((size_t)b * N + i) * ungrouped_ld(N);
    for (int j = threadIdx.x; j < i; j += blockDim.x) {
        const int cj = P.presorted ? j : I.order[j];
        out[j] = gnms_prune(rowbuf[cj], P.nms_threshold, P.temperature, P.pruning_method);
    }
}

// U1 straight from the boxes (gnms_forward_with_iou2d, round 4b): Ps[i][j] = f(iou(box at position i, box at position j)), j < i, with the
// matrix kernel's own arithmetic (pair_iou: bit-identical entries) from the boxes in rank order the score sort leaves in rbox -- no read
// of the 4 N^2-byte matrix (it is written beside, by gnms_iou2d's writers), 2 N^2 bytes out in 16-byte stores.  Hard-sorted scores only.
// B = 8, N = 4096: 258 -> 95 us (one workgroup per row: 161 -- 32 768 workgroups of at most four vectors per thread).
constexpr int kPermuteRows = 16;      // rows per workgroup of ungrouped_permute_boxes_kernel: the column boxes are loaded once for all of them
__global__ __launch_bounds__(256) void ungrouped_permute_boxes_kernel(int N, const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L,
                                                                      float* __restrict__ Ps_all) {
    const int b = blockIdx.y, i0 = blockIdx.x * kPermuteRows;
    const int n = gnms_count(counts, b, N);
    if (i0 >= n) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    const int iend = min(i0 + kPermuteRows, n);                      // rows [i0, iend); row i has the columns j < i
    const size_t ldp = ungrouped_ld(N);
    float* out0 = Ps_all + ((size_t)b * N + i0) * ldp;
    for (int j = threadIdx.x * 4; j < iend - 1; j += 1024) {        // (entries on / above the diagonal inside a row's last vector: never read)
        float4 cb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cb[u] = I.rbox[min(j + u, n - 1)];
        for (int i = max(i0, j + 1); i < iend; ++i) {
            const float4 a = I.rbox[i];                                // (workgroup-uniform address)
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = gnms_prune(pair_iou(a, cb[u]), P.nms_threshold, P.temperature, P.pruning_method);
            *reinterpret_cast<float4*>(out0 + (size_t)(i - i0) * ldp + j) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// 256 threads: thread t owns row t>>2 of the block and 16 consecutive columns (t&3)*16.. of every 64-column tile
__device__ __forceinline__ void ungrouped_load_tile(const float* __restrict__ p, bool live, float4 (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = live ? reinterpret_cast<const float4*>(p)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void ungrouped_solve_forward_kernel(const float* __restrict__ scores, int N, const int* __restrict__ counts,
                                                                      gnms_params P, char* ws, gnms_ws_layout L,
                                                                      const float* __restrict__ Ps_all) {
    __shared__ float xc[64];
    __shared__ float tb[64];
    const int b = blockIdx.y, blk = blockIdx.x;
    const int n = gnms_count(counts, b, N);
    const int i0 = blk * 64;
    if (i0 >= n) return;                                       // nobody waits for a block past the end
    const int rows = min(64, n - i0);
    const int nblk = (n + 63) >> 6;
    ImgPtrs I = img_ptrs(ws, L, b);
    const size_t ldp = ungrouped_ld(N);
    const float* Ps = Ps_all + (size_t)b * N * ldp;
    const float* s = scores + (size_t)b * N;
    u64* mail = I.W;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = t >> 2, cg = (t & 3) * 16;
    const bool live = r < rows;
    const float* strip = Ps + (size_t)(i0 + r) * ldp + cg;
    // wave 0: row `lane` of the diagonal tile in registers (independent of every x); entries on/above the diagonal are not used
    float4 trow[16];
    if (wave == 0) {
        const float4* p = reinterpret_cast<const float4*>(Ps + (size_t)(i0 + min(lane, rows - 1)) * ldp + i0);
#pragma unroll
        for (int q = 0; q < 16; ++q) trow[q] = p[q];
    }
    float acc = 0.0f;
    float4 cur[4], nxt[4];
    if (blk > 0) ungrouped_load_tile(strip, live, cur);
    for (int c = 0; c < blk; ++c) {
        if (c + 1 < blk) ungrouped_load_tile(strip + (size_t)(c + 1) * 64, live, nxt);
        __syncthreads();                                       // xc of the previous tile is consumed
        if (t < 64) xc[t] = mail_get(mail + c * 64 + t);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc += cur[q].x * xc[cg + 4 * q] + cur[q].y * xc[cg + 4 * q + 1] + cur[q].z * xc[cg + 4 * q + 2] + cur[q].w * xc[cg + 4 * q + 3];
            cur[q] = nxt[q];
        }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if ((t & 3) == 0 && live) tb[r] = s[P.presorted ? (i0 + r) : I.order[i0 + r]] - acc;
    __syncthreads();
    if (wave == 0) {
        float x = (lane < rows) ? tb[lane] : 0.0f;
        const float* tr = reinterpret_cast<const float*>(trow);
#pragma unroll
        for (int bb = 0; bb < 63; ++bb) {                       // 63 dependent steps: v_readlane + multiply + subtract, no LDS
            const float xbb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), bb));
            const float upd = x - tr[bb] * xbb;
            x = (lane > bb && lane < rows) ? upd : x;
        }
        if (lane < rows) {
            I.pre[i0 + lane] = x;
            if (blk + 1 < nblk) mail_put(mail + i0 + lane, x);
        }
    }
}

__global__ __launch_bounds__(1024) void ungrouped_backward_prepare_kernel(int N, char* ws, gnms_ws_layout L) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N && L.NB >= 2) img_ptrs(ws, L, blockIdx.y).W[N + k] = 0ull;       // backward mailbox
}

__global__ __launch_bounds__(256) void ungrouped_solve_backward_kernel(int N, const int* __restrict__ counts, gnms_params P, char* ws,
                                                                       gnms_ws_layout L, const float* __restrict__ Ps_all,
                                                                       float* __restrict__ grad_scores) {
    __shared__ float red[64 * 65];
    __shared__ float yc[64];
    const int b = blockIdx.y;
    const int n = gnms_count(counts, b, N);
    const int nblk = (n + 63) >> 6;
    ImgPtrs I = img_ptrs(ws, L, b);
    float* gs = grad_scores + (size_t)b * N;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if ((int)blockIdx.x >= nblk) {                              // padding positions get no gradient
        for (int k = (int)blockIdx.x * 64 + t; k < min(N, ((int)blockIdx.x + 1) * 64); k += blockDim.x)
            if (k >= n) gs[P.presorted ? k : I.order[k]] = 0.0f;
        return;
    }
    const int blk = nblk - 1 - (int)blockIdx.x;                 // the LAST block has no dependency: it is dispatched first
    const int i0 = blk * 64;
    const int rows = min(64, n - i0);
    const size_t ldp = ungrouped_ld(N);
    const float* Ps = Ps_all + (size_t)b * N * ldp;
    u64* mail = I.W + N;
    const int r = t >> 2, cg = (t & 3) * 16;
    // wave 0: column `lane` of the diagonal tile in registers (tcol[a] = T[a][lane], used for a > lane)
    float tcol[64];
    if (wave == 0) {
#pragma unroll
        for (int a = 0; a < 64; ++a) tcol[a] = Ps[(size_t)(i0 + min(a, rows - 1)) * ldp + i0 + lane];
    }
    float part[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) part[q] = 0.0f;
    float4 cur[4], nxt[4];
    // tile (c, blk): rows 64c + r, columns i0 + cg ..
    const float* colstrip = Ps + (size_t)r * ldp + i0 + cg;
    if (nblk - 1 > blk) ungrouped_load_tile(colstrip + (size_t)(nblk - 1) * 64 * ldp, (nblk - 1) * 64 + r < n, cur);
    for (int c = nblk - 1; c > blk; --c) {
        if (c - 1 > blk) ungrouped_load_tile(colstrip + (size_t)(c - 1) * 64 * ldp, true, nxt);
        __syncthreads();
        if (t < 64) yc[t] = (c * 64 + t < n) ? mail_get(mail + c * 64 + t) : 0.0f;
        __syncthreads();
        const float y = yc[r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            part[4 * q] += cur[q].x * y; part[4 * q + 1] += cur[q].y * y; part[4 * q + 2] += cur[q].z * y; part[4 * q + 3] += cur[q].w * y;
            cur[q] = nxt[q];
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) red[r * 65 + cg + q] = part[q];
    __syncthreads();
    if (wave == 0) {
        float sum = 0.0f;
        for (int rr = 0; rr < 64; ++rr) sum += red[rr * 65 + lane];
        int ci = 0;
        float y = 0.0f;
        if (lane < rows) {
            ci = P.presorted ? (i0 + lane) : I.order[i0 + lane];
            y = I.gx[i0 + lane] - sum;
        }
#pragma unroll
        for (int a = 63; a > 0; --a) {
            const float ya = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), a));
            const float upd = y - tcol[a] * ya;
            y = (lane < a && a < rows) ? upd : y;
        }
        if (lane < rows) {
            I.xsol[i0 + lane] = y;
            gs[ci] = y;
            if (blk > 0) mail_put(mail + i0 + lane, y);
        } else if (i0 + lane < N) {
            gs[i0 + lane] = 0.0f;                               // padding behind a ragged image (order is the identity there)
        }
    }
}

// dL/diou[r][c] = -(y_r x_c) f'(iou[r][c]) where column c precedes row r in NMS order, else 0 (input index space, coalesced)
__global__ __launch_bounds__(256) void ungrouped_grad_iou_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                                 gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ grad_iou) {
    const int b = blockIdx.z, rr = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = gnms_count(counts, b, N);
    if (c >= N) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    const size_t off = ((size_t)b * N + rr) * ld + c;
    float g = 0.0f;
    if (rr < n && c < n) {
        const int pr = I.rem[rr], pc = I.rem[c];
        if (pc < pr) g = (-(I.xsol[pr] * I.pre[pc])) * gnms_prune_grad(iou[off], P.nms_threshold, P.temperature, P.pruning_method);
    }
    grad_iou[off] = g;
}

}  // namespace gnms
