// iou3d_sym.h -- the 3D NMS-overlap matrix 0.5 * (1 + GIoU3D) of ONE box set with itself, every unordered pair evaluated ONCE.
//
// The overlap of a pair does not depend on which box is the row: min / max / add / mul are commutative and the expression
// (iou3d_pair.h) treats its two boxes alike, so out[i][j] == out[j][i] bit for bit (lib/core.py:365-419 is symmetric in b1, b2 too).
// The matrix kernel that evaluates all N^2 entries is VALU-bound (23 slots per pair: 0.51-0.58 of the HBM peak where a plain store
// stream reaches 0.70-0.73); this one evaluates the upper triangle of 128 x 128 macro tiles and stores every tile twice:
//   * directly from registers, row by row  (rows of tile I, 512-byte runs of the columns of tile J);
//   * MIRRORED through an LDS copy of the tile (rows of tile J, 512-byte runs of the columns of tile I).
// Half the arithmetic per byte; the store pattern alone (no arithmetic: gnms_profile_fill_sym, tools/store_geometry.py) reaches
// 5.6 TB/s at N = 4096 and N = 16384 (0.70 of 8 TB/s), 4-7 % below the band geometry of the 2D writers.
// Values: nms_overlap3d_guarded2 applies iou3d_pair.h's per-pair definition (re-associated expression outside the guard band around
// `thr`, the reference's exact operation order inside it and for boxes that are not sane), so the matrix equals the one
// iou3d_nms_fast_kernel writes, bit for bit (tests/test_gpu_parity.py::test_iou3d_symmetric_writer).
#pragma once
#include "iou3d_pair.h"

namespace gnms_iou3d {

constexpr int kSymT = 128;                          // macro tile: 128 x 128 entries
constexpr int kSymPitch = kSymT + 1;                // LDS row pitch in floats: the mirrored read walks a column (stride 129 = 1 mod 32 banks)
constexpr size_t kSymTileBytes = (size_t)kSymT * kSymPitch * sizeof(float);    // 66 048 B: two workgroups per CU

// two columns of one row: the per-pair definition of iou3d_pair.h (nms_overlap3d_guarded4, two columns wide)
__device__ __forceinline__ void nms_overlap3d_guarded2(const Row& a, const Cols2& b, unsigned colbad, bool cols_sane, float thr, float (&q)[2]) {
    const f2 q0 = nms_overlap3d(a, b);
    const f2 d0 = q0 - splat(thr);
    const float m = fminf(fabsf(d0.x), fabsf(d0.y));
    q[0] = q0.x; q[1] = q0.y;
    if (__any(!(m > kGuard3D)) || !cols_sane || a.bad != 0.0f) {              // rare
        const f2 e0 = nms_overlap3d_exact(a, b);
        const bool rb = a.bad != 0.0f;
        if (rb || (colbad & 1u) || !(fabsf(d0.x) > kGuard3D)) q[0] = e0.x;
        if (rb || (colbad & 2u) || !(fabsf(d0.y) > kGuard3D)) q[1] = e0.y;
    }
}

__host__ __device__ inline int sym_tiles_per_image(int N) {
    const int nt = (N + kSymT - 1) / kSymT;
    return nt * (nt + 1) / 2;
}

// tile id (row-major over the upper triangle: row I holds the tiles (I, I) .. (I, nt - 1)) -> (I, J)
__device__ __forceinline__ void sym_tile_of(int t, int nt, int* I, int* J) {
    // tiles before row i: i * nt - i (i - 1) / 2  =>  i = floor(((2 nt + 1) - sqrt((2 nt + 1)^2 - 8 t)) / 2), corrected for rounding
    const float a = (float)(2 * nt + 1);
    int i = (int)((a - sqrtf(a * a - 8.0f * (float)t)) * 0.5f);
    i = i < 0 ? 0 : (i >= nt ? nt - 1 : i);
    while (i > 0 && i * nt - i * (i - 1) / 2 > t) --i;
    while ((i + 1) * nt - (i + 1) * i / 2 <= t) ++i;
    *I = i;
    *J = i + (t - (i * nt - i * (i - 1) / 2));
}

template <bool NT>
__device__ __forceinline__ void sym_store2(float* p, float a, float b) {
    if (NT) { __builtin_nontemporal_store(a, p); __builtin_nontemporal_store(b, p + 1); }
    else *reinterpret_cast<float2*>(p) = make_float2(a, b);
}

// One workgroup of NW waves: macro tile (I, J), I <= J, of one image, in two passes with a workgroup barrier between them.
// rec [N][kRec] records, out [N][ld] (ld even, 8-byte aligned), tile: kSymTileBytes of LDS.
//   sym_tile_compute  wave w owns the rows 128 I + (128 / NW) w ... of the tile; a lane owns the columns 128 J + 2 lane, + 1 (one
//                     packed column pair: every v_pk_* works on exactly the lane's two pairs).  The row record is wave-uniform:
//                     scalar loads, the next row's requested before this row's arithmetic.  Stores the rows directly (512-byte
//                     runs) and, when I != J, parks the tile in LDS.
//   sym_tile_mirror   (I != J, after a barrier) output row = a column of the tile; its 128 entries = the tile's rows (all of them
//                     exist: I < J <= last tile).  Lane l stores the entries of the local rows 2 l, 2 l + 1: one 512-byte run per
//                     instruction (ds_read2_b32 at offsets k, k + 129: the odd pitch spreads a column over the banks).
template <int NW, bool NT>
__device__ __forceinline__ void sym_tile_compute(const float* __restrict__ rec, int N, float* __restrict__ out, long ld, int I, int J,
                                                 float thr, float* __restrict__ tile) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: the row records then come by SCALAR loads
    constexpr int RW = kSymT / NW;
    const int rl0 = wave * RW;
    const int r0 = I * kSymT + rl0;
    const int c = J * kSymT + 2 * lane;
    Cols2 cols;
    unsigned colbad = 0u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cc = (c + j) < N ? (c + j) : (N - 1);
        const float4* p = reinterpret_cast<const float4*>(rec + (size_t)cc * kRec);
        const float4 e = p[2];
        cols2_set(cols, j, p[0], p[1], e);
        colbad |= (e.w != 0.0f) ? (1u << j) : 0u;
    }
    const bool cols_sane = __all(colbad == 0u);
    const bool mirror = I != J;
    const int nrows = min(RW, N - r0);
    float* orow = out + (size_t)r0 * ld + c;
    float* trow = tile + rl0 * kSymPitch + 2 * lane;
    auto row_record = [&](int r) {                                    // wave-uniform: s_load_dwordx4 x 3
        const float* rr = rec + (size_t)(r0 + r) * kRec;
        Row a;
        a.vol = rr[0]; a.y0 = rr[1]; a.y1 = rr[2]; a.x0 = rr[3]; a.x1 = rr[4]; a.z0 = rr[5]; a.z1 = rr[6]; a.lx = rr[8]; a.ly = rr[9]; a.lz = rr[10];
        a.bad = rr[11];
        return a;
    };
    Row nxt = row_record(nrows > 0 ? 0 : -rl0 - I * kSymT);           // (no rows: any valid record)
    for (int r = 0; r < nrows; ++r) {
        const Row a = nxt;
        nxt = row_record(r + 1 < nrows ? r + 1 : r);                  // the next row's record is requested before this row's arithmetic
        float q[2];
        nms_overlap3d_guarded2(a, cols, colbad, cols_sane, thr, q);
        if (c + 1 < N) sym_store2<NT>(orow, q[0], q[1]);
        else if (c < N) orow[0] = q[0];
        if (mirror) { trow[0] = q[0]; trow[1] = q[1]; }               // ds_write2_b32 (4-byte aligned: the pitch is odd)
        orow += ld;
        trow += kSymPitch;
    }
}

template <int NW, bool NT>
__device__ __forceinline__ void sym_tile_mirror(int N, float* __restrict__ out, long ld, int I, int J, const float* __restrict__ tile) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int RW = kSymT / NW;
    const int rl0 = wave * RW;
    const float* tcol = tile + (2 * lane) * kSymPitch + rl0;
    float* mrow = out + (size_t)(J * kSymT + rl0) * ld + I * kSymT + 2 * lane;
    const int ncols = min(RW, N - (J * kSymT + rl0));
    for (int k = 0; k < ncols; ++k) {
        sym_store2<NT>(mrow, tcol[k], tcol[k + kSymPitch]);
        mrow += ld;
    }
}

// one macro tile per workgroup (iou3d_sym_kernel)
template <int NW, bool NT>
__device__ __forceinline__ void nms_overlap3d_sym_tile(const float* __restrict__ rec, int N, float* __restrict__ out, long ld, int I, int J,
                                                       float thr, float* __restrict__ tile) {
    sym_tile_compute<NW, NT>(rec, N, out, ld, I, J, thr, tile);
    if (I == J) return;                                               // (workgroup-uniform)
    __syncthreads();
    sym_tile_mirror<NW, NT>(N, out, ld, I, J, tile);
}

}  // namespace gnms_iou3d
