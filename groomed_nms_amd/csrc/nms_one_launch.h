// nms_one_launch.h -- a SMALL image's whole forward pass as ONE launch (round 6).
//
// The reference only ever hands the layer <= 500 boxes of one image (lib/loss/rpn_3d.py:732,791; lib/rpn_util.py:1293-1319), and at that
// size the three launches of the layer -- sort, threshold bits, chain (K3..K6) -- are 24 us of kernels behind ~12 us of launch boundaries
// and ~12 us of host enqueue.  Here the three are ROLES of one grid, handing over through flags in the workspace:
//
//   [B * nsort sort workgroups] [B * NB * split table workgroups] [B chain workgroups] [B CSR workgroups]
//
//   sort    sort_count_body<KPW, FUSED>: the score sort by counting, exactly the workgroup of sort_count_kernel; outputs through agent-scope
//           stores, then its flag.
//   table   waits for the image's sort flags, then thresholds rows 64 tb + r of the matrix IN RANK ORDER and leaves -- not W, which nobody
//           behind the fast tail reads -- the words of the scan's triangular table themselves: word (source block bb <= tb, target row r) =
//           bits s of !(iou[order[64 tb + r]][order[64 bb + s]] <= thr), lib/groomed_nms.py:250.  A set entry is scattered to its bit by
//           one LDS atomic (the thresholded matrix of an NMS input is sparse: a few entries per row), the columns' ranks come from an LDS
//           copy of rankof; columns no row of the block can see (rank >= 64 (tb + 1)) are not even loaded.  The table has the reference's
//           own orientation (row = the later box, column = the leader), so nothing is assumed about the matrix's symmetry and the
//           symmetry check of the three-launch path has no counterpart here.
//   chain   leaders_sb_body<SRC, FUSED> (one super-block) -> fast_final_body: as in tail_kernel, the table copied from the image above.
//   CSR     csr_build_body, and the launch's last act: the workspace's call counter moves on.
//
// Flags.  Nothing can be zeroed ahead of a launch that has no launch in front of it, so a flag is a STRONG granule: all 64 bits are a
// function of (tag, slot), tag = the call counter's successor -- read by every workgroup at its start, stored by the image's CSR workgroup
// at the very end, when every other workgroup of the image has long read it (each of them is upstream of the CSR workgroup's wait).  Stale
// flags of earlier calls or graph replays carry other tags; the bytes of a recycled allocation would have to match all 64 bits.
// A workgroup only waits for workgroups with a LOWER block index.
#pragma once
#include "nms_kernels.h"

namespace gnms {
namespace {

// wave-level: lanes [0, count) poll one granule each until all of them carry the launch's flag (count <= 64)
__device__ __forceinline__ void one_launch_wait(const u64* g, const int count, const unsigned tag, const unsigned slot0) {
    const int lane = threadIdx.x & 63, i = lane < count ? lane : 0;
    const u64 want = strong_gran(tag, slot0 + (unsigned)i);
    while (__ballot(gran_load(g + i) != want) != 0ull) __builtin_amdgcn_s_sleep(1);
}

constexpr int kOneLaunchMaxN = 1024;     // one super-block

// table workgroup w of image b (w = tb * split + part: rows [part * 64 / split, (part + 1) * 64 / split) of rank block tb); split in {1, 2, 4}
__device__ __forceinline__ void one_launch_bits_from_matrix(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                            const float thr, char* ws, gnms_ws_layout L, const int b, const int w,
                                                            const int split, const unsigned tag, const int nsort) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NP = (N + 63) & ~63;
    int* rankL = reinterpret_cast<int*>(smem);                                   // [NP] rank of input column c
    u64* tab = reinterpret_cast<u64*>(smem + (size_t)NP * 4);                    // [tb + 1][64]
    __shared__ int rowL[64];                                                     // input row of rank 64 tb + r
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = w / split, part = w - tb * split;
    const int k0 = tb * 64;
    if (wave == 0) one_launch_wait(I.gran + 32, nsort, tag, kSlotSort);
    __syncthreads();
    if (k0 < n) {                                                                // (workgroup-uniform)
        const int rpg = 64 / split, rpw = 4 / split;                             // rows per workgroup / per wave
        const int rows0 = part * rpg;
        const int lim = k0 + 64;                                                 // columns of rank >= lim: no row of the block looks at them
        for (int i = tid; i < n; i += 1024) rankL[i] = coh_load(I.rankof + i);
        if (tid < 64) rowL[tid] = (k0 + tid < n) ? coh_load(I.order + k0 + tid) : 0;
        for (int i = tid; i < (tb + 1) * 64; i += 1024) tab[i] = 0ull;
        __syncthreads();
        const float* m = iou + (size_t)b * N * ld;
        const int nchunks = (n + 255) >> 8;                                      // <= 4 chunks of 256 columns, lane l: columns 256 c + 4 l ..
        int rk[4][4];
        bool need[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col0 = c * 256 + 4 * lane;
            int lo = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rk[c][j] = (c < nchunks && col0 + j < n) ? rankL[col0 + j] : 0x7fffffff;
                lo = min(lo, rk[c][j]);
            }
            need[c] = lo < lim;                                                  // (col0 < n <= ld and both multiples of 4: the 16 bytes are inside the row)
        }
        float4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rows0 + wave * rpw + u;
            const bool row_ok = u < rpw && k0 + r < n;
            const float* p = m + (size_t)rowL[row_ok ? r : 0] * ld + 4 * lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row_ok && need[c]) v[u][c] = load_nt_f4(p + c * 256);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rows0 + wave * rpw + u;
            const bool row_ok = u < rpw && k0 + r < n;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (!(row_ok && need[c])) continue;
                const float e[4] = {v[u][c].x, v[u][c].y, v[u][c].z, v[u][c].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (rk[c][j] < lim && !(e[j] <= thr))                        // lib/groomed_nms.py:250 (NaN -> removed)
                        atomicOr(&tab[(rk[c][j] >> 6) * 64 + r], 1ull << (rk[c][j] & 63));
            }
        }
        __syncthreads();
        u64* img = I.W + (size_t)(tb * (tb + 1) / 2) * 64;                       // the table's words of target block tb, source blocks 0 .. tb
        for (int i = tid; i < (tb + 1) * rpg; i += 1024) {
            const int bb = i / rpg, r = rows0 + (i - bb * rpg);
            coh_store(img + bb * 64 + r, tab[bb * 64 + r]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);                                          // vmcnt(0)
    __syncthreads();
    if (tid == 0) coh_store(I.gran + (size_t)3 * 32 + w, strong_gran(tag, kSlotBits + (unsigned)w));
}

// bytes of dynamic LDS the launch needs (every workgroup asks for the chain's)
__host__ __device__ inline size_t one_launch_lds_size(int N) {
    const size_t NP = (size_t)((N + 63) & ~63);
    const size_t chain = fast_tail_lds_size(N, 1024), sort = NP * 8, bits = NP * 4 + (size_t)kSB * 64 * 8;
    size_t m = chain > sort ? chain : sort;
    return m > bits ? m : bits;
}

template <int SRC>
__global__ __launch_bounds__(1024) void one_launch_kernel(const float* __restrict__ scores, const float* __restrict__ src, int N, long ld,
                                                          const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L,
                                                          float* __restrict__ prob, long long* __restrict__ valid, long long* __restrict__ invalid,
                                                          int* __restrict__ nvalid, int* __restrict__ ninvalid, long long* __restrict__ order_out,
                                                          int B, int kpw, int split) {
    GNMS_TINIT();
    const int NP = (N + 63) & ~63, nsort = NP / kpw, nbits = L.NB * split;
    int bx = (int)blockIdx.x, b;
    if (bx < B * nsort) {
        b = bx / nsort;
        const unsigned tag = (unsigned)gnms_next_epoch(coh_load(img_ptrs(ws, L, b).misc + 8));
        if (kpw == 32) sort_count_body<32, true>(scores, nullptr, N, counts, ws, L, order_out, 0, bx - b * nsort, b, 0, tag);
        else sort_count_body<64, true>(scores, nullptr, N, counts, ws, L, order_out, 0, bx - b * nsort, b, 0, tag);
        return;
    }
    bx -= B * nsort;
    if (bx < B * nbits) {
        b = bx / nbits;
        const unsigned tag = (unsigned)gnms_next_epoch(coh_load(img_ptrs(ws, L, b).misc + 8));
        one_launch_bits_from_matrix(src, N, ld, counts, P.nms_threshold, ws, L, b, bx - b * nbits, split, tag, nsort);
        return;
    }
    bx -= B * nbits;
    if (bx < B) {                                                                // the image's chain
        b = bx;
        ImgPtrs I = img_ptrs(ws, L, b);
        const unsigned tag = (unsigned)gnms_next_epoch(coh_load(I.misc + 8));
        if (threadIdx.x < 64) one_launch_wait(I.gran + 32, nsort, tag, kSlotSort);
        __syncthreads();
        const int last = leaders_sb_body<SRC, true>(N, counts, ws, L, b, 0, 1, 0, src, ld, P.nms_threshold, P.temperature, P.pruning_method, 1024,
                                                    tag, nbits);
        fast_final_body<1, SRC, true>(src, N, ld, counts, P, ws, L, 1024, prob, valid, invalid, nvalid, ninvalid, b, last, tag);
        GNMS_TFLUSH(ws, L, b);
        return;
    }
    b = bx - B;                                                                  // the image's CSR workgroup
    ImgPtrs I = img_ptrs(ws, L, b);
    const unsigned tag = (unsigned)gnms_next_epoch(coh_load(I.misc + 8));
    csr_build_body<1, true>(N, counts, ws, L, b, tag);
    __syncthreads();
    if (threadIdx.x == 0) coh_store(I.misc + 8, (int)tag);                       // every other workgroup of the image has read the counter long ago
}

}  // namespace
}  // namespace gnms
