// iou3d_tile.h -- the wave tile of the 3D NMS-overlap matrix 0.5 * (1 + GIoU3D), shared by iou3d_nms_fast_kernel (iou_kernels.hip)
// and the launch that carries the per-image chain beside the write (nms_layer.hip).  Reference: lib/core.py:305-421 (generalized),
// lib/loss/rpn_3d.py:781; arithmetic and guard band: iou3d_pair.h.
#pragma once
#include "iou3d_pair.h"
#include "iou_tile.h"

namespace gnms_iou3d {

// One wave: rows i0 .. i0 + tile_rows - 1 (below row_end) of image `img` against the 256 columns starting at c0.
// RA [B][M][kRec], RB [B][N][kRec] corner-AABB records, out [B][M][ld].  Entries within the guard band of `thr` are written in the
// reference's exact operation order (nms_overlap3d_guarded4).
template <bool VEC>
__device__ __forceinline__ void nms_overlap3d_tile(const float* __restrict__ RA, const float* __restrict__ RB, int M, int N,
                                                   float* __restrict__ out, long ld, int img, int i0, int c0, int lane, int tile_rows,
                                                   int row_end, float thr) {
    if (row_end > M) row_end = M;                                 // a launch may cover the rows [row0, row_end) only
    if (c0 >= N || i0 >= row_end) return;
    const float* ra = RA + (size_t)img * M * kRec;
    const float* rb = RB + (size_t)img * N * kRec;
    float* o3 = out + (size_t)img * M * ld;
    Cols2 cols[2];
    int col[4];
    unsigned colbad = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        col[j] = VEC ? (c0 + 4 * lane + j) : (c0 + lane + 64 * j);
        const int cc = col[j] < N ? col[j] : (N - 1);
        const float4* p = reinterpret_cast<const float4*>(rb + (size_t)cc * kRec);
        const float4 e = p[2];
        cols2_set(cols[j >> 1], j & 1, p[0], p[1], e);
        colbad |= (e.w != 0.0f) ? (1u << j) : 0u;
    }
    const bool cols_sane = __all(colbad == 0u);
    const int nrows = min(tile_rows, row_end - i0);
    for (int r = 0; r < nrows; ++r) {
        // the row record is wave-uniform and read-only: scalar loads (s_load_dwordx4 x 3), no VALU, no LDS.  (Broadcasting it
        // from a lane with 10 v_readlane per row measured 111 instead of 99 us at B=8, N=4096.)
        const float* rr = ra + (size_t)(i0 + r) * kRec;
        Row a;
        a.vol = rr[0]; a.y0 = rr[1]; a.y1 = rr[2]; a.x0 = rr[3]; a.x1 = rr[4]; a.z0 = rr[5]; a.z1 = rr[6]; a.lx = rr[8]; a.ly = rr[9]; a.lz = rr[10];
        a.bad = rr[11];
        float res[4];
        nms_overlap3d_guarded4(a, cols, colbad, cols_sane, thr, res);      // entries near `thr`, boxes that are not sane: the reference's exact order
        const size_t roff = (size_t)(i0 + r) * ld;
        if (VEC && col[3] < N) {
            gnms_iou::store_nt_f4(o3 + roff + col[0], res[0], res[1], res[2], res[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (col[j] < N) o3[roff + col[j]] = res[j];
        }
    }
}

}  // namespace gnms_iou3d
