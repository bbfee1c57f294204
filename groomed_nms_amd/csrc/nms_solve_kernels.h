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
//   U2  ungrouped_solve_forward_kernel   one workgroup per block of 128 positions, ALL blocks in flight: block b streams
//       its 128 x 128b strip of Ps tile by tile (next tile prefetched into registers), and consumes x of block c as soon as
//       block c publishes it (mailbox in global memory, decoupled look-back: a block only waits for lower-numbered blocks,
//       which are dispatched first).  The diagonal tile is inverted ahead of the chain, so a block's own part is a product.
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
// two consecutive slots by one lane; a slot at or past `n` counts as present with the value 0
__device__ __forceinline__ u64 mail_peek(const u64* mail, int p, int n) {
    return p < n ? __hip_atomic_load(mail + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1ull << 32);
}
__device__ __forceinline__ float2 mail_finish2(const u64* mail, int p, int n, u64 w0, u64 w1) {   // w0, w1: a first attempt already made
    while ((w0 >> 32) == 0 || (w1 >> 32) == 0) {
        __builtin_amdgcn_s_sleep(1);
        w0 = mail_peek(mail, p, n);
        w1 = mail_peek(mail, p + 1, n);
    }
    return make_float2(__uint_as_float((unsigned)(w0 & 0xffffffffu)), __uint_as_float((unsigned)(w1 & 0xffffffffu)));
}
__device__ __forceinline__ float2 mail_get2(const u64* mail, int p, int n) {
    const u64 w0 = mail_peek(mail, p, n), w1 = mail_peek(mail, p + 1, n);
    return mail_finish2(mail, p, n, w0, w1);
}
// row pitch of the scratch matrix: whole 128-column tiles, because the solves load the diagonal tile of the last row block in full
// (columns i0 .. i0+127; the entries past N are never used, but they must lie inside the allocation)
__host__ __device__ inline size_t ungrouped_ld(int N) { return (size_t)((N + 127) & ~127); }
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
// B = 8, N = 4096: 258 -> 95 us (one workgroup per row: 161 -- 32 768 workgroups of at most four vectors per thread); round 5, the packed
// plain row body: 77 us (every row of the workgroup for every vector, so that the row's box is a hoistable scalar load: 96).
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
    // Boxes that "divide plainly" (iou_tile.h: finite, no negative zero, x2 >= x1, y2 >= y1, magnitudes inside 2^-13 .. 2^20 -- pixel boxes
    // always are) take the matrix writers' packed row body: the intersection as med3(min3(..)), the division as rcp + six packed fma steps,
    // bit for bit the quotient pair_iou's IEEE division gives.  18 + 30 VALU slots per row of four entries instead of 40 + 44.
    bool rows_plain = true;
    for (int i = i0; i < iend; ++i) rows_plain &= gnms_iou::box_divides_plainly(I.rbox[i]);   // (workgroup-uniform)
    for (int j = threadIdx.x * 4; j < iend - 1; j += 1024) {        // (entries on / above the diagonal inside a row's last vector: never read)
        float4 cb[4];
        gnms_iou::ColPairs cp;
        bool cok = true;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            cb[u] = I.rbox[min(j + u, n - 1)];
            gnms_iou::colpairs_set(cp, u, cb[u]);
            cok &= gnms_iou::box_divides_plainly(cb[u]);
        }
        const bool plain = rows_plain && __all(cok);                  // (wave-uniform)
        for (int i = max(i0, j + 1); i < iend; ++i) {
            const float4 a = I.rbox[i];                                // (uniform except in the vectors that cross the diagonal: no SGPR operands)
            float v[4];
            if (plain) {
                using namespace gnms_iou;
                const float aw = a.z - a.x, ah = a.w - a.y;
                const float aarea = aw * ah;
                const gnms_f2 sx1 = {a.x, a.x}, sy1 = {a.y, a.y}, sx2 = {a.z, a.z}, sy2 = {a.w, a.w}, sa = {aarea, aarea};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const gnms_f2 dx1 = sx2 - cp.x1[p], dx2 = cp.x2[p] - sx1;
                    const gnms_f2 dy1 = sy2 - cp.y1[p], dy2 = cp.y2[p] - sy1;
                    const gnms_f2 w = {__builtin_amdgcn_fmed3f(hw_min3(dx1.x, dx2.x, cp.w[p].x), 0.0f, aw), __builtin_amdgcn_fmed3f(hw_min3(dx1.y, dx2.y, cp.w[p].y), 0.0f, aw)};
                    const gnms_f2 h = {__builtin_amdgcn_fmed3f(hw_min3(dy1.x, dy2.x, cp.h[p].x), 0.0f, ah), __builtin_amdgcn_fmed3f(hw_min3(dy1.y, dy2.y, cp.h[p].y), 0.0f, ah)};
                    const gnms_f2 inter = w * h;
                    const gnms_f2 uni = (sa + cp.area[p]) - inter;
                    const gnms_f2 q = div2_plain(inter, uni);
                    v[2 * p] = gnms_prune(q.x, P.nms_threshold, P.temperature, P.pruning_method);
                    v[2 * p + 1] = gnms_prune(q.y, P.nms_threshold, P.temperature, P.pruning_method);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = gnms_prune(pair_iou(a, cb[u]), P.nms_threshold, P.temperature, P.pruning_method);
            }
            *reinterpret_cast<float4*>(out0 + (size_t)(i - i0) * ldp + j) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ---- the two solves on blocks of 128 positions (round 5) -----------------------------------------------------------------
// A block's workgroup (512 threads; thread t owns row t>>2 and the 32 consecutive columns (t&3)*32.. of every 128-column tile)
// inverts its own diagonal tile while the lower blocks are still solving: D = (I + T)^-1 of the unit lower triangular 128 x 128
// tile from the inverses of its two 64 x 64 diagonal sub-tiles (one wave each, lane = column, forward substitution on the identity:
// no cross-lane step) and D21 = -D22 T21 D11 (two 64^3 products out of LDS).  The block's solve is then a product,
// x = D (s - sum_c P[blk][c] x_c): behind the LAST source block there are two 128 x 128 matrix-vector products of register-resident
// tiles (32 fused multiply-adds per thread and a quad reduction each) where the 64-position version had a tile product and 63
// dependent cross-lane steps, and there are half as many hand-offs.  D^T goes into the (otherwise unused) upper triangle of the
// diagonal tile of Ps for the backward solve -- written after the block has published x.  Round 4: 200 / 221 us at B = 8, N = 4096.
constexpr int kUB = 128;                  // positions per block
constexpr int kUP = 132;                  // LDS row pitch in floats: 16-byte aligned rows, consecutive rows four banks apart
// eight waves compute; a ninth polls the mailbox (a wave's loads return in order: a poll issued behind the prefetch of a 64-KiB tile
// could not complete before the tile had landed)
constexpr int kUThreads = 576;
constexpr size_t kUngroupedFwdLds = (size_t)(2 * kUB * kUP + 3 * kUB) * sizeof(float);
constexpr size_t kUngroupedBwdLds = (size_t)(kUB * kUP + 3 * kUB) * sizeof(float);

// eight 16-byte loads, unconditional: the callers clamp the row to one that exists and never use what a dead row produced, so that
// the compiler sees straight-line loads and counts them (a load under a branch made it wait for ALL outstanding loads, vmcnt(0))
__device__ __forceinline__ void ungrouped_load_tile(const float* __restrict__ p, float4 (&dst)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = reinterpret_cast<const float4*>(p)[q];
}
// sum over the four lanes of a quad, in every lane (two DPP quad_perm moves; __shfl_xor compiles to two ds_bpermute round trips)
__device__ __forceinline__ float ungrouped_quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}
// 64 x 64 product out of LDS, 512 threads, two rows x four columns per thread: C[i][j] = sign * sum_k A[i][k] B[k][j]
__device__ __forceinline__ void ungrouped_mm64(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C, float sign, int t) {
    const int ti = (t >> 4) * 2, tj = (t & 15) * 4;
    float c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
        const float4 a0 = *reinterpret_cast<const float4*>(A + ti * kUP + k);
        const float4 a1 = *reinterpret_cast<const float4*>(A + (ti + 1) * kUP + k);
        const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float4 bv = *reinterpret_cast<const float4*>(Bm + (k + kk) * kUP + tj);
            c0[0] = __builtin_fmaf(a0v[kk], bv.x, c0[0]); c0[1] = __builtin_fmaf(a0v[kk], bv.y, c0[1]);
            c0[2] = __builtin_fmaf(a0v[kk], bv.z, c0[2]); c0[3] = __builtin_fmaf(a0v[kk], bv.w, c0[3]);
            c1[0] = __builtin_fmaf(a1v[kk], bv.x, c1[0]); c1[1] = __builtin_fmaf(a1v[kk], bv.y, c1[1]);
            c1[2] = __builtin_fmaf(a1v[kk], bv.z, c1[2]); c1[3] = __builtin_fmaf(a1v[kk], bv.w, c1[3]);
        }
    }
    *reinterpret_cast<float4*>(C + ti * kUP + tj) = make_float4(sign * c0[0], sign * c0[1], sign * c0[2], sign * c0[3]);
    *reinterpret_cast<float4*>(C + (ti + 1) * kUP + tj) = make_float4(sign * c1[0], sign * c1[1], sign * c1[2], sign * c1[3]);
}
// dot product of 32 register values with 32 consecutive LDS floats, four partial sums
__device__ __forceinline__ float ungrouped_dot32(const float (&a)[32], const float* __restrict__ v) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 bv = *reinterpret_cast<const float4*>(v + 4 * q);
        s0 = __builtin_fmaf(a[4 * q], bv.x, s0); s1 = __builtin_fmaf(a[4 * q + 1], bv.y, s1);
        s2 = __builtin_fmaf(a[4 * q + 2], bv.z, s2); s3 = __builtin_fmaf(a[4 * q + 3], bv.w, s3);
    }
    return (s0 + s1) + (s2 + s3);
}

#ifndef GNMS_UTICK
#define GNMS_UTICK(i) do {} while (0)                          // (tools/usolve_ticks.hip: phase ticks of the inversion)
#endif
#ifndef GNMS_UHOP
#define GNMS_UHOP(blk, slot) do {} while (0)                   // (tools/usolve_ticks.hip: when a block of image 0 sees / publishes what)
#endif
// The inverse of the unit lower triangular 128 x 128 tile I + T (T in Tl, strictly lower, zero elsewhere) into Dl (zero on entry), 512
// threads, kUInvertBarriers barriers.  Bottom-up: the four 32 x 32 diagonal sub-tiles (one wave each, lane = column, forward substitution
// on the identity: no cross-lane step; the next row of T is fetched while the current one is used -- with one wait per LDS read the
// 64 x 64 version of this step was 12.4 us of a 17.5-us inversion), then D21 = -D22 T21 D11 at 32 and at 64.
constexpr int kUInvertBarriers = 5;
__device__ __forceinline__ void ungrouped_invert_diag32(const float* __restrict__ Tl, float* __restrict__ Dl, int t) {
    const int lane = t & 63, wave = t >> 6;
    if (wave < 4 && lane < 32) {
        const int h0 = wave * 32;
        float X[32];
        float4 tr[8], tn[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) tr[q] = tn[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i + 1 < 32) {                                  // row i + 1 (wave-uniform address: LDS broadcast)
                const float* Tn = Tl + (h0 + i + 1) * kUP + h0;
#pragma unroll
                for (int q = 0; q < 8; ++q) if (4 * q < i + 1) tn[q] = *reinterpret_cast<const float4*>(Tn + 4 * q);
            }
            float s0 = (i == lane) ? 1.0f : 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (4 * q < i) s0 = __builtin_fmaf(-tr[q].x, X[4 * q], s0);
                if (4 * q + 1 < i) s1 = __builtin_fmaf(-tr[q].y, X[4 * q + 1], s1);
                if (4 * q + 2 < i) s2 = __builtin_fmaf(-tr[q].z, X[4 * q + 2], s2);
                if (4 * q + 3 < i) s3 = __builtin_fmaf(-tr[q].w, X[4 * q + 3], s3);
            }
            const float v = (s0 + s1) + (s2 + s3);
            X[i] = v;
            Dl[(h0 + i) * kUP + h0 + lane] = v;
#pragma unroll
            for (int q = 0; q < 8; ++q) tr[q] = tn[q];
        }
    }
    GNMS_UTICK(1);
    __syncthreads();
    GNMS_UTICK(2);
}
// 32 x 32 products out of LDS for the two 64 x 64 diagonal sub-tiles at once (pair p = t >> 8), one row x four columns per thread:
// C_p[i][j] = sign * sum_k A_p[i][k] B_p[k][j]; the operands of pair p start 64 rows and 64 columns behind those of pair 0
__device__ __forceinline__ void ungrouped_mm32x2(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C, float sign, int t) {
    const int off = (t >> 8) * (64 * kUP + 64);
    const int ti = (t & 255) >> 3, tj = (t & 7) * 4;
    A += off; Bm += off; C += off;
    float c0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
        const float4 a0 = *reinterpret_cast<const float4*>(A + ti * kUP + k);
        const float a0v[4] = {a0.x, a0.y, a0.z, a0.w};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float4 bv = *reinterpret_cast<const float4*>(Bm + (k + kk) * kUP + tj);
            c0[0] = __builtin_fmaf(a0v[kk], bv.x, c0[0]); c0[1] = __builtin_fmaf(a0v[kk], bv.y, c0[1]);
            c0[2] = __builtin_fmaf(a0v[kk], bv.z, c0[2]); c0[3] = __builtin_fmaf(a0v[kk], bv.w, c0[3]);
        }
    }
    *reinterpret_cast<float4*>(C + ti * kUP + tj) = make_float4(sign * c0[0], sign * c0[1], sign * c0[2], sign * c0[3]);
}
// D21 = -D22 T21 D11 at 32 and at 64 (the first product of each level goes through an unused upper right part of Tl)
__device__ __forceinline__ void ungrouped_invert_offdiag(float* __restrict__ Tl, float* __restrict__ Dl, int t) {
    // 32 -> 64, both halves: T21 D11 goes through the (zero, unused) upper right 32 x 32 of each 64 x 64 diagonal sub-tile of Tl
    ungrouped_mm32x2(Tl + 32 * kUP, Dl, Tl + 32, 1.0f, t);
    __syncthreads();
    ungrouped_mm32x2(Dl + 32 * kUP + 32, Tl + 32, Dl + 32 * kUP, -1.0f, t);
    GNMS_UTICK(6);
    __syncthreads();
    ungrouped_mm64(Tl + 64 * kUP, Dl, Tl + 64, 1.0f, t);                       // T21 D11 -> Tl[0..63][64..127]
    GNMS_UTICK(3);
    __syncthreads();
    ungrouped_mm64(Dl + 64 * kUP + 64, Tl + 64, Dl + 64 * kUP, -1.0f, t);      // D21 = -D22 (T21 D11)
    GNMS_UTICK(4);
    __syncthreads();
    GNMS_UTICK(5);
}

// The strip's tiles go through three register buffers, two loads ahead of the use; the stage count is padded to a multiple of three
// with null stages IN FRONT (x = 0 from the poller, the tile index clamped).  No load is issued that is not used: the registers of a
// redundant prefetch are the ones the code behind the loop reuses, and it then waits a tile fetch for them on the path of the hand-off.
__global__ __launch_bounds__(kUThreads) void ungrouped_solve_forward_kernel(const float* __restrict__ scores, int N, const int* __restrict__ counts,
                                                                      gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ Ps_all) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Tl = reinterpret_cast<float*>(smem);          // the diagonal tile T (strictly lower); its upper right quadrant: T21 D11
    float* Dl = Tl + kUB * kUP;                            // D = (I + T)^-1
    float* xc = Dl + kUB * kUP;                            // two buffers: the poller fills one while the other is read
    float* rb = xc + 2 * kUB;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int n = gnms_count(counts, b, N);
    const int i0 = blk * kUB;
    if (i0 >= n) return;                                       // nobody waits for a block past the end
    const int rows = min(kUB, n - i0);
    const int nblk = (n + kUB - 1) / kUB;
    ImgPtrs I = img_ptrs(ws, L, b);
    const size_t ldp = ungrouped_ld(N);
    float* Ps = Ps_all + (size_t)b * N * ldp;
    u64* mail = I.W;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int pad = (3 - blk % 3) % 3;
    if (wave == 8) {   // the poller
        for (int k = 0; k < 1 + kUInvertBarriers; ++k) __syncthreads();            // the diagonal tile and its inverse
        for (int c = -pad; c < blk; ++c) {
            const float2 v = c < 0 ? make_float2(0.f, 0.f) : mail_get2(mail, c * kUB + 2 * lane, n);
            *reinterpret_cast<float2*>(xc + ((c + 4) & 1) * kUB + 2 * lane) = v;
            if (c == blk - 1) GNMS_UHOP(blk, 0);
            __syncthreads();
        }
        __syncthreads();                                       // rb
        return;
    }
    const int r = t >> 2, cg = (t & 3) * 32;
    const bool live = r < rows;
    const float* strip = Ps + (size_t)min(i0 + r, n - 1) * ldp + cg;   // (a dead row reads the image's last row: finite, never used)
    {   // the diagonal tile: strictly lower part of the live rows, zero elsewhere
        float4 d[8];
        ungrouped_load_tile(strip + i0, d);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = cg + 4 * q;
            *reinterpret_cast<float4*>(Tl + r * kUP + c) = make_float4((live && c < r) ? d[q].x : 0.f, (live && c + 1 < r) ? d[q].y : 0.f,
                                                                       (live && c + 2 < r) ? d[q].z : 0.f, (live && c + 3 < r) ? d[q].w : 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(Dl + r * kUP + cg + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);   // (D is zero right of its diagonal)
    }
    __syncthreads();
    ungrouped_invert_diag32(Tl, Dl, t);
    // the row's score IN FRONT of the tile loads: a wave's loads return in order, and behind the two (redundant) prefetches of the last
    // stages the wait for it was a tile fetch long -- on the path of every hand-off
    const float sr = scores[(size_t)b * N + (P.presorted ? min(i0 + r, n - 1) : I.order[min(i0 + r, n - 1)])];
    const int last_tile = max(blk - 1, 0);
    float4 ta[8], tb[8], tc[8];
    ungrouped_load_tile(strip + (size_t)min(max(-pad, 0), last_tile) * kUB, ta);
    ungrouped_load_tile(strip + (size_t)min(max(-pad + 1, 0), last_tile) * kUB, tb);
    ungrouped_invert_offdiag(Tl, Dl, t);
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
#define GNMS_U_STAGE(C, USE, FILL, LOAD)                                                                                   \
    {                                                                                                                      \
        if (LOAD) ungrouped_load_tile(strip + (size_t)max((C) + 2, 0) * kUB, FILL);                                        \
        __syncthreads();                                       /* x of block C (zeros for C < 0) is in xc[(C + 4) & 1] */  \
        const float* xs = xc + (((C) + 4) & 1) * kUB + cg;                                                                 \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                                    \
            const float4 xv = *reinterpret_cast<const float4*>(xs + 4 * q);                                                \
            acc0 = __builtin_fmaf(USE[q].x, xv.x, acc0); acc1 = __builtin_fmaf(USE[q].y, xv.y, acc1);                      \
            acc2 = __builtin_fmaf(USE[q].z, xv.z, acc2); acc3 = __builtin_fmaf(USE[q].w, xv.w, acc3);                      \
        }                                                                                                                  \
    }
    if (blk > 0) {   // pad + blk stages, a multiple of three; the last two issue no load (LOAD is a literal: no branch)
        int c = -pad;
        for (; c + 3 < blk; c += 3) {
            GNMS_U_STAGE(c, ta, tc, true)
            GNMS_U_STAGE(c + 1, tb, ta, true)
            GNMS_U_STAGE(c + 2, tc, tb, true)
        }
        GNMS_U_STAGE(c, ta, tc, true)
        GNMS_U_STAGE(c + 1, tb, ta, false)
        GNMS_U_STAGE(c + 2, tc, tb, false)
    }
#undef GNMS_U_STAGE
    // row r of D out of LDS (D is zero right of its diagonal).  Read here, not in front of the loop: held across the loop the 32 values
    // were spilled to scratch and came back behind the last barrier
    float dv[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(Dl + r * kUP + cg + 4 * q);
        dv[4 * q] = v.x; dv[4 * q + 1] = v.y; dv[4 * q + 2] = v.z; dv[4 * q + 3] = v.w;
    }
    const float acc = ungrouped_quad_sum((acc0 + acc1) + (acc2 + acc3));
    GNMS_UHOP(blk, 1);
    if ((t & 3) == 0) rb[r] = live ? sr - acc : 0.0f;
    __syncthreads();
    GNMS_UHOP(blk, 2);
    const float x = ungrouped_quad_sum(ungrouped_dot32(dv, rb + cg));
    if ((t & 3) == 0 && live) {
        if (blk + 1 < nblk) mail_put(mail + i0 + r, x);
        I.pre[i0 + r] = x;
    }
    GNMS_UHOP(blk, 3);
    if (live) {   // D^T over the diagonal tile of Ps (row r: 0 left of the diagonal, 1 on it, D[c][r] right of it), for the backward solve
        float* drow = Ps + (size_t)(i0 + r) * ldp + i0 + cg;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = cg + 4 * q + u;
                v[u] = c > r ? Dl[c * kUP + r] : (c == r ? 1.0f : 0.0f);
            }
            *reinterpret_cast<float4*>(drow + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

__global__ __launch_bounds__(1024) void ungrouped_backward_prepare_kernel(int N, char* ws, gnms_ws_layout L) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N && L.NB >= 2) img_ptrs(ws, L, blockIdx.y).W[N + k] = 0ull;       // backward mailbox
}

// y = D^T (g - sum_{c > blk} P[c][blk]^T y_c), blocks published from the last to the first.  The sources c >= blk + 2 are accumulated per
// thread (four rows of the source tile x eight columns) and reduced over the rows through LDS once, BEFORE the nearest source arrives;
// the nearest source's tile waits transposed in registers (thread = column, 32 interleaved rows), so that behind y_{blk+1} there are, as
// in the forward solve, two register-resident matrix-vector products.
__global__ __launch_bounds__(kUThreads) void ungrouped_solve_backward_kernel(int N, const int* __restrict__ counts, gnms_params P, char* ws,
                                                                       gnms_ws_layout L, const float* __restrict__ Ps_all,
                                                                       float* __restrict__ grad_scores) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    float* yc = red + kUB * kUP;                           // two buffers
    float* rb = yc + 2 * kUB;
    const int b = blockIdx.y;
    const int n = gnms_count(counts, b, N);
    const int nblk = (n + kUB - 1) / kUB;
    ImgPtrs I = img_ptrs(ws, L, b);
    float* gs = grad_scores + (size_t)b * N;
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= nblk) {                              // padding positions get no gradient
        for (int k = (int)blockIdx.x * kUB + t; k < min(N, ((int)blockIdx.x + 1) * kUB); k += blockDim.x)
            if (k >= n) gs[P.presorted ? k : I.order[k]] = 0.0f;
        return;
    }
    const int blk = nblk - 1 - (int)blockIdx.x;                 // the LAST block has no dependency: it is dispatched first
    const int i0 = blk * kUB;
    const int rows = min(kUB, n - i0);
    const size_t ldp = ungrouped_ld(N);
    const float* Ps = Ps_all + (size_t)b * N * ldp;
    u64* mail = I.W + N;
    const bool has_near = blk + 1 < nblk, has_far = blk + 2 < nblk;
    if (t >= 512) {   // the poller
        const int lane = t & 63;
        if (has_near) __syncthreads();
        for (int c = nblk - 1; c >= blk + 1; --c) {
            const int p0 = c * kUB + 2 * lane;
            const u64 w0 = mail_peek(mail, p0, n), w1 = mail_peek(mail, p0 + 1, n);
            if (c == blk + 1 && has_far) { __syncthreads(); __syncthreads(); }   // the reduction of the far part (the first attempt is in flight)
            const float2 v = mail_finish2(mail, p0, n, w0, w1);
            *reinterpret_cast<float2*>(yc + (c & 1) * kUB + 2 * lane) = v;
            __syncthreads();
        }
        __syncthreads();                                       // rb
        return;
    }
    const int r = t >> 2, qd = t & 3, cg = qd * 32;
    const bool live = r < rows;
    float dv[32];                                               // row r of D^T (the forward solve left it over the diagonal tile)
    {
        float4 d[8];
        ungrouped_load_tile(Ps + (size_t)min(i0 + r, n - 1) * ldp + i0 + cg, d);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = cg + 4 * q;
            dv[4 * q] = (live && c >= r && c < rows) ? d[q].x : 0.f; dv[4 * q + 1] = (live && c + 1 >= r && c + 1 < rows) ? d[q].y : 0.f;
            dv[4 * q + 2] = (live && c + 2 >= r && c + 2 < rows) ? d[q].z : 0.f; dv[4 * q + 3] = (live && c + 3 >= r && c + 3 < rows) ? d[q].w : 0.f;
        }
    }
    float nt[32];                                               // nt[k] = P[(blk+1) * 128 + qd + 4k][i0 + r]
    if (has_near) {
        float4 d[8];
        ungrouped_load_tile(Ps + (size_t)min((blk + 1) * kUB + r, n - 1) * ldp + i0 + cg, d);   // (a dead row: finite values against y = 0)
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(red + r * kUP + cg + 4 * q) = d[q];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) nt[k] = red[(qd + 4 * k) * kUP + r];
    }
    // far tiles: thread = rows 4 * (t >> 4) .. + 3 of the source tile, columns 8 * (t & 15) .. + 7
    const int rg = (t >> 4) * 4, cq = (t & 15) * 8;
    float part[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) part[k] = 0.0f;
    if (has_far) {
        // (rows past the image read its last row against y = 0)
        float4 cur[8], nxt[8];
#define GNMS_U_FAR_LOAD(C, DST)                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
            const float* p = Ps + (size_t)min((C) * kUB + rg + j, n - 1) * ldp + i0 + cq;                                  \
            DST[2 * j] = reinterpret_cast<const float4*>(p)[0]; DST[2 * j + 1] = reinterpret_cast<const float4*>(p)[1];    \
        }
        GNMS_U_FAR_LOAD(nblk - 1, cur)
        for (int c = nblk - 1; c >= blk + 2; --c) {
            if (c - 1 >= blk + 2) { GNMS_U_FAR_LOAD(c - 1, nxt) }   // (a redundant load here would be waited for by the copy below)
            __syncthreads();                                   // y of block c is in yc[c & 1]
            const float4 yv = *reinterpret_cast<const float4*>(yc + (c & 1) * kUB + rg);
            const float yj[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part[0] = __builtin_fmaf(cur[2 * j].x, yj[j], part[0]); part[1] = __builtin_fmaf(cur[2 * j].y, yj[j], part[1]);
                part[2] = __builtin_fmaf(cur[2 * j].z, yj[j], part[2]); part[3] = __builtin_fmaf(cur[2 * j].w, yj[j], part[3]);
                part[4] = __builtin_fmaf(cur[2 * j + 1].x, yj[j], part[4]); part[5] = __builtin_fmaf(cur[2 * j + 1].y, yj[j], part[5]);
                part[6] = __builtin_fmaf(cur[2 * j + 1].z, yj[j], part[6]); part[7] = __builtin_fmaf(cur[2 * j + 1].w, yj[j], part[7]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) cur[q] = nxt[q];
        }
#undef GNMS_U_FAR_LOAD
    }
    float far = 0.0f;
    if (has_far) {
        __syncthreads();                                       // (the near tile's transposed reads of red are done)
        *reinterpret_cast<float4*>(red + (t >> 4) * kUP + cq) = make_float4(part[0], part[1], part[2], part[3]);
        *reinterpret_cast<float4*>(red + (t >> 4) * kUP + cq + 4) = make_float4(part[4], part[5], part[6], part[7]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) far += red[(qd + 4 * k) * kUP + r];    // 32 row groups: this lane takes eight, the quad all of them
        far = ungrouped_quad_sum(far);
    }
    const float gxr = I.gx[min(i0 + r, n - 1)];
    const int ci = P.presorted ? min(i0 + r, n - 1) : I.order[min(i0 + r, n - 1)];
    float near = 0.0f;
    if (has_near) {
        __syncthreads();                                       // y of block blk + 1
        const float* yn = yc + ((blk + 1) & 1) * kUB + qd;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            s0 = __builtin_fmaf(nt[k], yn[4 * k], s0); s1 = __builtin_fmaf(nt[k + 1], yn[4 * k + 4], s1);
            s2 = __builtin_fmaf(nt[k + 2], yn[4 * k + 8], s2); s3 = __builtin_fmaf(nt[k + 3], yn[4 * k + 12], s3);
        }
        near = ungrouped_quad_sum((s0 + s1) + (s2 + s3));
    }
    if (qd == 0) rb[r] = live ? (gxr - far) - near : 0.0f;
    __syncthreads();
    const float y = ungrouped_quad_sum(ungrouped_dot32(dv, rb + cg));
    if (qd == 0) {
        if (live) {
            if (blk > 0) mail_put(mail + i0 + r, y);
            I.xsol[i0 + r] = y;
            gs[ci] = y;
        } else if (i0 + r < N) {
            gs[i0 + r] = 0.0f;                                  // padding behind a ragged image (order is the identity there)
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
