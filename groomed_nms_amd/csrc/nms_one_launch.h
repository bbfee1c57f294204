// nms_one_launch.h -- a SMALL image's whole forward pass as ONE launch (round 6).
//
// The reference only ever hands the layer <= 500 boxes of one image (lib/loss/rpn_3d.py:732,791; lib/rpn_util.py:1293-1319), and at that
// size the three launches of the layer -- sort, threshold bits, chain (K3..K6) -- are 24 us of kernels behind ~12 us of launch boundaries
// and ~12 us of host enqueue.  Here the three are ROLES of one grid, handing over through flags in the workspace:
//
//   [B * nsort sort workgroups] [B * NB * split table workgroups] [B chain workgroups] [B CSR workgroups]
//   (one-call entry, table in x order: [sort] [x sort] [B * NB table workgroups] [chain] [CSR] [matrix writers])
//
//   sort    sort_count_body<KPW, FUSED>: the score sort by counting, exactly the workgroup of sort_count_kernel; outputs through agent-scope
//           stores, then its flag.
//   table   thresholds a few INPUT rows of the matrix while the sort still runs, waits for the image's sort flags, and leaves -- not W, which
//           nobody behind the fast tail reads -- the words of the scan's triangular table themselves: word (source block bb <= tb, target
//           row r) = bits s of !(iou[order[64 tb + r]][order[64 bb + s]] <= thr), lib/groomed_nms.py:250.  A set entry is scattered to its
//           bit by one LDS atomic (the thresholded matrix of an NMS input is sparse: a few entries per row), the ranks come from an LDS
//           copy of rankof.  The table has the reference's own orientation (row = the later box, column = the leader), so nothing is
//           assumed about the matrix's symmetry and the symmetry check of the three-launch path has no counterpart here.
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

// table workgroup w of image b: the 64 / split INPUT rows [w * 64 / split, ...) of the matrix, split in {1, 2, 4} (wave v: rows v * 4 / split + u).
// Which rows those are does not depend on the sort, so the rows are requested and thresholded BEFORE the sort's flags are waited for -- the
// one memory latency of the role that matters hides behind the sort; behind the flags only the ranks are fetched (LDS copy of rankof), the
// set entries scattered to the words (target rank k = rankof[row], source block bb = rankof[column] >> 6) of an LDS table, and row k's
// words for bb <= k >> 6 stored (every word of the table image has exactly one writer: the workgroup that holds its target's row).
__device__ __forceinline__ void one_launch_bits_from_matrix(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                            const float thr, char* ws, gnms_ws_layout L, const int b, const int w,
                                                            const int split, const unsigned tag, const int nsort) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NP = (N + 63) & ~63;
    int* rankL = reinterpret_cast<int*>(smem);                                   // [NP] rank of input column c
    u64* tab = reinterpret_cast<u64*>(smem + (size_t)NP * 4);                    // [rows of the workgroup][kSB source blocks]
    __shared__ int krL[64];                                                      // rank of the workgroup's row
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rpg = 64 / split, rpw = 4 / split;                                 // rows per workgroup / per wave
    const int i0 = w * rpg;                                                      // first input row
    const int nb = (n + 63) >> 6;
    const float* m = iou + (size_t)b * N * ld;
    const int nchunks = (n + 255) >> 8;                                          // <= 4 chunks of 256 columns, lane l: columns 256 c + 4 l ..
    unsigned bits[4] = {0u, 0u, 0u, 0u};                                         // row u: bit 4 c + j = !(iou[row][256 c + 4 l + j] <= thr)
    if (i0 < n) {                                                                // (workgroup-uniform)
        float4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + wave * rpw + u;
            const bool row_ok = u < rpw && i < n;
            const float* p = m + (size_t)(row_ok ? i : i0) * ld + 4 * lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) {                                        // (4 l < n <= ld and both multiples of 4: the 16 bytes are inside the row)
                v[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row_ok && c < nchunks && c * 256 + 4 * lane < n) v[u][c] = load_nt_f4(p + c * 256);
            }
        }
        for (int i = tid; i < rpg * kSB; i += 1024) tab[i] = 0ull;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool row_ok = u < rpw && i0 + wave * rpw + u < n;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int col0 = c * 256 + 4 * lane;
                if (!(row_ok && c < nchunks)) continue;
                const float e[4] = {v[u][c].x, v[u][c].y, v[u][c].z, v[u][c].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col0 + j < n && !(e[j] <= thr)) bits[u] |= 1u << (4 * c + j);   // lib/groomed_nms.py:250 (NaN -> removed)
            }
        }
    }
    if (wave == 0) one_launch_wait(I.gran + 32, nsort, tag, kSlotSort);
    __syncthreads();
    if (i0 < nb * 64) {                                                          // (rows past the image's last rank block have no words in the table)
        for (int i = tid; i < n; i += 1024) rankL[i] = coh_load(I.rankof + i);
        if (tid < rpg) krL[tid] = (i0 + tid < n) ? coh_load(I.rankof + i0 + tid) : i0 + tid;   // (padding rows: rank = index, all-zero words)
        if (i0 >= n) for (int i = tid; i < rpg * kSB; i += 1024) tab[i] = 0ull;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u >= rpw) continue;
            const int rl = wave * rpw + u;
            const int tbk = krL[rl] >> 6;                                        // the row's own rank block: sources beyond it are never looked at
            unsigned todo = bits[u];
            while (todo) {
                const int q = __builtin_ctz(todo);
                todo &= todo - 1u;
                const int rc = rankL[(q >> 2) * 256 + 4 * lane + (q & 3)];
                if ((rc >> 6) <= tbk) atomicOr(&tab[rl * kSB + (rc >> 6)], 1ull << (rc & 63));
            }
        }
        __syncthreads();
        for (int i = tid; i < rpg * kSB; i += 1024) {
            const int rl = i / kSB, bb = i - rl * kSB;
            const int k = krL[rl], tb = k >> 6;
            if (bb <= tb && tb < nb) coh_store(I.W + (size_t)(tb * (tb + 1) / 2 + bb) * 64 + (k & 63), tab[i]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);                                          // vmcnt(0)
    __syncthreads();
    if (tid == 0) coh_store(I.gran + (size_t)3 * 32 + w, strong_gran(tag, kSlotBits + (unsigned)w));
}

// bytes of dynamic LDS the launch needs (every workgroup asks for the chain's)
// The table straight from the BOXES (gnms_forward_with_iou2d's one launch: the matrix is an output there).  In rank space nothing has to be
// permuted: task (tb, bb <= tb) -- its number IS the table's pair index -- holds the 64 target boxes of rank block tb in its lanes, walks
// the sources of block bb (wave-uniform, broadcast with v_readlane) and a lane's bits are its word, stored coalesced.  The decision is
// bitmask_boxes_body's, operation for operation (sign of fma(-thr, uni, inter) outside a guard band of 8 ulp, the IEEE division inside it
// and for boxes without a positive finite area), i.e. bit for bit `!(iou <= thr)` of the matrix the writers of the same launch store.
// Workgroup w of an image: tasks [w * tpw, (w + 1) * tpw), each on 16 / tpw waves that share the 64 sources; tpw in {1, 2, 4, 8, 16}.
__device__ __forceinline__ void one_launch_bits_from_boxes(int N, const int* __restrict__ counts, const float thr, char* ws, gnms_ws_layout L,
                                                           const int b, const int w, const int tpw, const unsigned tag, const int nsort) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* wordsL = reinterpret_cast<u64*>(smem);                                  // [tpw][64]
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpt = 16 / tpw, tl = wave / wpt, part = wave - tl * wpt;           // waves per task, the wave's task, its share of the sources
    const int task = w * tpw + tl;
    int tb = 0;
    while ((tb + 1) * (tb + 2) / 2 <= task) ++tb;
    const int bb = task - tb * (tb + 1) / 2;
    const int nb = (n + 63) >> 6;
    const bool live = tb < nb;                                                   // (wave-uniform; tasks past the image's last block: nothing to write)
    if (wpt > 1) for (int i = tid; i < tpw * 64; i += 1024) wordsL[i] = 0ull;
    if (wave == 0) one_launch_wait(I.gran + 32, nsort, tag, kSlotSort);
    __syncthreads();
    u64 word = 0ull;
    if (live) {
        const int kt = tb * 64 + lane, ks = bb * 64 + lane;
        const float4 tbx = coh_load_f4(I.rbox + (kt < n ? kt : n - 1));
        const float4 sbx = coh_load_f4(I.rbox + (ks < n ? ks : n - 1));
        const float tarea = (tbx.z - tbx.x) * (tbx.w - tbx.y), sarea = (sbx.z - sbx.x) * (sbx.w - sbx.y);
        const bool targets_ok = __all((tarea > 0.0f) && (tarea < INFINITY));
        const u64 sources_ok = __ballot((sarea > 0.0f) && (sarea < INFINITY));
        const float guard = fmaxf(fabsf(thr), 1.0f) * 9.6e-7f;                   // 8 ulp at the threshold's magnitude (bitmask_boxes_body)
        const int spp = 64 / wpt, nsrc = min(64, n - bb * 64);
        const int s1 = min(nsrc, (part + 1) * spp);
        for (int s = part * spp; s < s1; ++s) {
            const float ax1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbx.x), s));
            const float ay1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbx.y), s));
            const float ax2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbx.z), s));
            const float ay2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbx.w), s));
            const float aa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sarea), s));
            const float wd = fmaxf(gnms_iou3d::vmin_s(ax2, tbx.z) - gnms_iou3d::vmax_s(ax1, tbx.x), 0.0f);
            const float ht = fmaxf(gnms_iou3d::vmin_s(ay2, tbx.w) - gnms_iou3d::vmax_s(ay1, tbx.y), 0.0f);
            const float inter = wd * ht;
            const float uni = (aa + tarea) - inter;
            const float d = __builtin_fmaf(-thr, uni, inter);
            const bool unsure = !(fabsf(d) > guard * uni);                       // also true for NaN
            bool bit;
            if (!(targets_ok && ((sources_ok >> s) & 1ull)) || __any(unsure)) bit = !(inter / uni <= thr);
            else bit = d > 0.0f;
            word |= bit ? (1ull << s) : 0ull;
        }
        if (kt >= n) word = 0ull;
        if (wpt > 1 && word != 0ull) atomicOr(&wordsL[tl * 64 + lane], word);
    }
    if (wpt > 1) {
        __syncthreads();
        if (live && part == 0) coh_store(I.W + (size_t)task * 64 + lane, wordsL[tl * 64 + lane]);
    } else if (live) {
        coh_store(I.W + (size_t)task * 64 + lane, word);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);                                          // vmcnt(0)
    __syncthreads();
    if (tid == 0) coh_store(I.gran + (size_t)3 * 32 + w, strong_gran(tag, kSlotBits + (unsigned)w));
}

// The table from the boxes with the SOURCES in x order (larger batches of larger images: B = 8, N = 1024).  In rank space a table task cannot be
// culled -- 136 tasks of 64 x 64 pairs per image at N = 1024, every one a workgroup with a CU to itself -- where bitmask_boxes_body's x-sorted
// columns keep 15-20 % of the rows.  Here workgroup tb of an image holds the 64 TARGET boxes of rank block tb as rows, wave v the 64 sources
// x-rank 64 v .. 64 v + 63 as its lanes (N <= 1024: sixteen waves cover every source), rows that do not reach into the wave's hull are skipped,
// and a set pair -- a few per row -- goes to bit (source rank & 63) of word (target row, source rank >> 6) of an LDS table by one LDS atomic;
// the words of source blocks <= tb then leave as the table image's (tb, bb) entries.  Needs the x sort as a second sort role of the launch
// (sort_count_body role 1: xidx, xbox).  The decision is bitmask_boxes_body's general row, operation for operation.
__device__ __forceinline__ void one_launch_bits_from_boxes_x(int N, const int* __restrict__ counts, const float thr, char* ws, gnms_ws_layout L,
                                                             const int b, const int tb, const unsigned tag, const int nsort) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* tab = reinterpret_cast<u64*>(smem);                                     // [64 target rows][kSB source blocks]
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = (n + 63) >> 6;
    const bool live = tb < nb;                                                   // (workgroup-uniform)
    for (int i = tid; i < 64 * kSB; i += 1024) tab[i] = 0ull;
    if (wave == 0) one_launch_wait(I.gran + 32, nsort, tag, kSlotSort);          // the score sort: rbox, rankof
    if (wave == 1) one_launch_wait(I.gran + 64, nsort, tag, kSlotSort + 32u);    // the x sort: xidx, xbox
    __syncthreads();
    const int k0 = tb * 64;
    if (live && wave * 64 < n) {                                                 // (wave-uniform)
        const int nrows = min(64, n - k0);
        const float4 rb = coh_load_f4(I.rbox + min(k0 + lane, n - 1));
        const int p = wave * 64 + lane, pp = p < n ? p : n - 1;                  // clamped duplicates: harmless in the hull, never set
        const float4 cb = coh_load_f4(I.xbox + pp);
        const int crank = coh_load(I.rankof + coh_load(I.xidx + pp));
        const float carea = (cb.z - cb.x) * (cb.w - cb.y);
        const bool cols_ok = __all((carea > 0.0f) && (carea < INFINITY));
        const float hx0 = wave_min_f(cb.x), hy0 = wave_min_f(cb.y), hx1 = wave_max_f(cb.z), hy1 = wave_max_f(cb.w);
        const bool cull = cols_ok && (thr >= 0.0f);
        const float rarea = (rb.z - rb.x) * (rb.w - rb.y);
        const bool row_fine = (rarea > 0.0f) && (rarea < INFINITY);
        const u64 rows_ok = __ballot(row_fine);
        const bool reaches = (rb.z > hx0) && (rb.x < hx1) && (rb.w > hy0) && (rb.y < hy1);
        u64 todo = __ballot((lane < nrows) && (!(cull && row_fine) || reaches));
        const float guard = fmaxf(fabsf(thr), 1.0f) * 9.6e-7f;                   // 8 ulp at the threshold's magnitude (bitmask_boxes_body)
        const bool mine = (p < n) && ((crank >> 6) <= tb);                       // a source the table holds for this target block
        while (todo) {                                                           // wave-uniform loop over the surviving rows
            const int r = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const float ax1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.x), r));
            const float ay1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.y), r));
            const float ax2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.z), r));
            const float ay2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.w), r));
            const float aa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rarea), r));
            const float w = fmaxf(gnms_iou3d::vmin_s(ax2, cb.z) - gnms_iou3d::vmax_s(ax1, cb.x), 0.0f);
            const float h = fmaxf(gnms_iou3d::vmin_s(ay2, cb.w) - gnms_iou3d::vmax_s(ay1, cb.y), 0.0f);
            const float inter = w * h;
            const float uni = (aa + carea) - inter;
            const float d = __builtin_fmaf(-thr, uni, inter);
            const bool unsure = !(fabsf(d) > guard * uni);                       // also true for NaN
            bool bit;
            if (!(cols_ok && ((rows_ok >> r) & 1ull)) || __any(unsure)) bit = !(inter / uni <= thr);
            else bit = d > 0.0f;
            if (bit && mine) atomicOr(&tab[r * kSB + (crank >> 6)], 1ull << (crank & 63));
        }
    }
    __syncthreads();
    if (live) {
        for (int i = tid; i < 64 * kSB; i += 1024) {
            const int r = i / kSB, bb = i - r * kSB;
            if (bb <= tb) coh_store(I.W + (size_t)(tb * (tb + 1) / 2 + bb) * 64 + r, (k0 + r < n) ? tab[i] : 0ull);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);                                          // vmcnt(0)
    __syncthreads();
    if (tid == 0) coh_store(I.gran + (size_t)3 * 32 + tb, strong_gran(tag, kSlotBits + (unsigned)tb));
}

// the chain's side region (the image's order / scores / boxes by rank, leaders_sb_body<.., FUSED>) lies behind everything else of the chain
__host__ __device__ inline size_t one_launch_lds_side(int N) { return (fast_tail_lds_size(N, 1024) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t one_launch_lds_size(int N, bool boxes) {
    const size_t NP = (size_t)((N + 63) & ~63);
    const size_t chain = one_launch_lds_side(N) + 8192 + (boxes ? 16384 : 0), sort = NP * 8, bits = NP * 4 + (size_t)64 * kSB * 8;
    size_t m = chain > sort ? chain : sort;
    return m > bits ? m : bits;
}

// the image's chain workgroup (c < B) or its CSR workgroup (B <= c < 2 B)
template <int SRC>
__device__ __forceinline__ void one_launch_chain_or_csr(const float* __restrict__ src, int N, long ld, const int* __restrict__ counts, gnms_params P,
                                                        char* ws, gnms_ws_layout L, float* __restrict__ prob, long long* __restrict__ valid,
                                                        long long* __restrict__ invalid, int* __restrict__ nvalid, int* __restrict__ ninvalid,
                                                        const int B, const int c, const int nsort, const int nbits) {
    int b;
    if (c < B) {                                                                 // the image's chain
        b = c;
        ImgPtrs I = img_ptrs(ws, L, b);
        const unsigned tag = (unsigned)gnms_next_epoch(coh_load(I.misc + 8));
        GNMS_T0();
        if (threadIdx.x < 64) one_launch_wait(I.gran + 32, nsort, tag, kSlotSort);
        __syncthreads();
        GNMS_TACC(16);                                                           // (developer build: the chain's wait for the sort)
        const int last = leaders_sb_body<SRC, true>(N, counts, ws, L, b, 0, 1, 0, src, ld, P.nms_threshold, P.temperature, P.pruning_method, 1024,
                                                    tag, nbits, one_launch_lds_side(N));
        fast_final_body<1, SRC, true>(src, N, ld, counts, P, ws, L, 1024, prob, valid, invalid, nvalid, ninvalid, b, last, tag);
        GNMS_TFLUSH(ws, L, b);
        return;
    }
    b = c - B;                                                                   // the image's CSR workgroup
    ImgPtrs I = img_ptrs(ws, L, b);
    const unsigned tag = (unsigned)gnms_next_epoch(coh_load(I.misc + 8));
    csr_build_body<1, true>(N, counts, ws, L, b, tag);
    __syncthreads();
    if (threadIdx.x == 0) coh_store(I.misc + 8, (int)tag);                       // every other workgroup of the image has read the counter long ago
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
        GNMS_T0();
        if (kpw == 32) sort_count_body<32, true>(scores, nullptr, N, counts, ws, L, order_out, 0, bx - b * nsort, b, 0, tag);
        else sort_count_body<64, true>(scores, nullptr, N, counts, ws, L, order_out, 0, bx - b * nsort, b, 0, tag);
        GNMS_TACC_IF(bx == 0, 17);                                               // (developer build: the first sort workgroup, start to flag)
        GNMS_TFLUSH(ws, L, bx == 0 ? 0 : 1);
        return;
    }
    bx -= B * nsort;
    if (bx < B * nbits) {
        b = bx / nbits;
        const unsigned tag = (unsigned)gnms_next_epoch(coh_load(img_ptrs(ws, L, b).misc + 8));
        GNMS_T0();
        one_launch_bits_from_matrix(src, N, ld, counts, P.nms_threshold, ws, L, b, bx - b * nbits, split, tag, nsort);
        GNMS_TACC_IF(bx == nbits - 1, 18);                                       // (developer build: image 0's last table workgroup, start to flag)
        GNMS_TFLUSH(ws, L, bx == nbits - 1 ? 0 : 1);
        return;
    }
    bx -= B * nbits;
    one_launch_chain_or_csr<SRC>(src, N, ld, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, B, bx, nsort, nbits);
}

}  // namespace
}  // namespace gnms
