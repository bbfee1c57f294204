// iou_tile.h -- the 2D IoU tile body shared by iou2d_kernel (iou_kernels.hip) and the fused IoU + score-sort kernel
// (nms_layer.hip).  Reference semantics: lib/core.py:178-218 intersect, :480-508 iou (mode='combinations').
#pragma once
#include "gnms_common.h"

namespace gnms_iou {

constexpr int kTileRows = 64;
constexpr int kWaveCols = 256;   // 64 lanes x 4 columns
constexpr int kWavesPerWG = 8;   // tools/bw_variants.hip: 8 waves + non-temporal stores = 5.6 TB/s (4 waves, plain stores: 5.3)
constexpr int kWGCols = kWaveCols * kWavesPerWG;

// Rows per workgroup.  A wave walks its rows one after the other (one 1-KiB store each), so a launch with few workgroups is
// bound by that chain, not by HBM: B=8, N=512 is 64 workgroups of 64 rows = 22 us for 8 MB.  Small problems get shorter tiles
// until the grid has a few workgroups per CU.
__host__ inline int tile_rows_for(int B, int M, int N) {
    int rows = kTileRows;
    while (rows > 8 && (long long)B * ((M + rows - 1) / rows) * ((N + kWGCols - 1) / kWGCols) < 1024) rows >>= 1;
    return rows;
}

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float relu0(float v) { return fmaxf(v, 0.0f); }
// v_min_f32 / v_max_f32 issued directly, first operand wave-uniform (an SGPR straight from a scalar load).  fminf/fmaxf on a
// value the compiler cannot prove canonical (anything loaded from memory) cost an extra v_max_f32 x, x, x each.  Hardware
// semantics in IEEE mode = fminf/fmaxf: a NaN operand yields the other operand.
__device__ __forceinline__ float hw_min_s(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max_s(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }
// streaming 16-byte store: the matrix is written once and read once by another kernel much later
__device__ __forceinline__ void store_nt_f4(float* p, float a, float b, float c, float d) {
    __builtin_nontemporal_store(a, p);
    __builtin_nontemporal_store(b, p + 1);
    __builtin_nontemporal_store(c, p + 2);
    __builtin_nontemporal_store(d, p + 3);
}

// ---- the division of the tile kernels --------------------------------------------------------------------------------------------
// `inter / uni` must be the correctly rounded quotient (the reference divides in IEEE fp32).  The compiler's expansion of a / b is
//     v_div_scale x2, v_rcp, six fma-type steps, v_div_fmas, v_div_fixup               (10 VALU slots per quotient)
// and of the ~92 VALU instructions the tile spends per row of four entries 40 are these -- on a kernel that is as much VALU- as
// store-bound (134M pairs x 23 lane-slots = 79 us of VALU at B = 8, N = 4096, beside a ~96 us store stream).  The scale / fixup
// steps only act on zero, infinite, NaN or denormal operands and on quotients near the ends of the exponent range (ISA: V_DIV_SCALE
// scales when an operand or 1/b or a/b is denormal, when exp(a) - exp(b) >= 96, or when exp(a) <= 23); everywhere else the quotient
// is exactly the bare sequence rcp + 6 fma, which this header issues PACKED (v_pk_fma_f32, two quotients per instruction):
// 4 rcp + 14 packed = 18 slots per row instead of 40, bit for bit the same result.
// "Everywhere else" is decided per tile from the boxes (box_divides_plainly): every coordinate finite and 0 or 2^-13 <= |c| < 2^20,
// x2 >= x1, y2 >= y1.  Then widths, heights and areas are 0 or in [2^-72, 2^42] (differences of multiples of 2^-36), inter <= both
// areas (fp subtraction and multiplication are monotone), uni = 0 only for 0 / 0 (NaN either way) and otherwise >= the larger area,
// so 0 <= inter / uni <~ 1 with inter = 0 or >= 2^-72 and uni <= 2^43: no operand, reciprocal or quotient anywhere near a case
// V_DIV_SCALE acts on.  A tile with any other box (pixel boxes never are) takes the compiler's full division.
typedef float gnms_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool box_divides_plainly(const float4 v) {
    auto coord_ok = [](float c) {
        const unsigned u = __float_as_uint(c) & 0x7fffffffu;
        // +0, or 2^-13 <= |c| < 2^20 (NaN / Inf fail; so does -0: the plain row body below must never see a negative zero)
        return __float_as_uint(c) == 0u || (u - 0x39000000u) < (0x49800000u - 0x39000000u);
    };
    return coord_ok(v.x) && coord_ok(v.y) && coord_ok(v.z) && coord_ok(v.w) && v.z >= v.x && v.w >= v.y;
}

// q[j] = n[j] / d[j], correctly rounded, for operands box_divides_plainly vouches for: the steps of the compiler's fp32 division
// (LLVM AMDGPU LowerFDIV32) without scale and fixup, two quotients per packed instruction
__device__ __forceinline__ void div4_plain(const float (&n)[4], const float (&d)[4], float (&q)[4]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const gnms_f2 a = {n[2 * p], n[2 * p + 1]}, b = {d[2 * p], d[2 * p + 1]}, one = {1.0f, 1.0f};
        gnms_f2 r = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
        const gnms_f2 e = __builtin_elementwise_fma(-b, r, one);
        r = __builtin_elementwise_fma(e, r, r);
        gnms_f2 qq = a * r;
        gnms_f2 t = __builtin_elementwise_fma(-b, qq, a);
        qq = __builtin_elementwise_fma(t, r, qq);
        t = __builtin_elementwise_fma(-b, qq, a);
        qq = __builtin_elementwise_fma(t, r, qq);
        q[2 * p] = qq.x; q[2 * p + 1] = qq.y;
    }
}

__device__ __forceinline__ gnms_f2 div2_plain(const gnms_f2 a, const gnms_f2 b) {       // the same steps for one packed pair
    const gnms_f2 one = {1.0f, 1.0f};
    gnms_f2 r = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
    const gnms_f2 e = __builtin_elementwise_fma(-b, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    gnms_f2 qq = a * r;
    gnms_f2 t = __builtin_elementwise_fma(-b, qq, a);
    qq = __builtin_elementwise_fma(t, r, qq);
    t = __builtin_elementwise_fma(-b, qq, a);
    return __builtin_elementwise_fma(t, r, qq);
}

// ---- the intersection of the plain row body --------------------------------------------------------------------------------------
// lib/core.py:210-212 computes  w = relu(min(ax2, bx2) - max(ax1, bx1)).  min(ax2, bx2) - max(ax1, bx1) is, in exact arithmetic, the
// smallest of the four differences  ax2 - ax1, bx2 - bx1, ax2 - bx1, bx2 - ax1,  and the subtraction the reference performs is the
// one of the four that attains it; rounding is monotone, so its fp32 result equals the smallest of the four ROUNDED differences:
//     w = relu(min(fl(ax2 - bx1), fl(bx2 - ax1), cw, rw))          cw = fl(bx2 - bx1) per column, rw = fl(ax2 - ax1) per row
//       = med3(min3(fl(ax2 - bx1), fl(bx2 - ax1), cw), 0, rw)      (rw >= 0: box_divides_plainly)
// bit for bit (finite coordinates, no negative zero: a difference is then never -0, and equal candidates are the same bits).  The two
// differences are packed subtractions over a column pair (v_pk_add_f32, the row coordinate from an SGPR), cw / rw are the widths the
// areas need anyway: 3 VALU slots per entry and dimension where min, max, subtract, relu take 4 -- and the products and the union
// pack as well: 7.5 slots per entry in front of the division instead of 11.  The matrix writers are VALU-bound (3.2e): this is
// where their time goes.
struct ColPairs {                       // a lane's four columns as two packed pairs, field by field
    gnms_f2 x1[2], y1[2], x2[2], y2[2], w[2], h[2], area[2];
};
__device__ __forceinline__ void colpairs_set(ColPairs& c, int j, const float4 v) {
    c.x1[j >> 1][j & 1] = v.x; c.y1[j >> 1][j & 1] = v.y; c.x2[j >> 1][j & 1] = v.z; c.y2[j >> 1][j & 1] = v.w;
    const float w = v.z - v.x, h = v.w - v.y;
    c.w[j >> 1][j & 1] = w; c.h[j >> 1][j & 1] = h;
    c.area[j >> 1][j & 1] = w * h;                                    // lib/core.py:502-503
}
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_clamp0_s(float x, float hi) { float r; asm("v_med3_f32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi)); return r; }

// Rows of a full tile (every lane owns four existing columns) whose boxes all divide plainly: `ra` holds row i0 + r in lane r;
// orow = &out[i0][first column of the lane].  ROWS_CT > 0: exactly that many rows, unrolled into one basic block; 0: `rows` at run time.
template <int ROWS_CT>
__device__ __forceinline__ void iou2d_rows_plain(const ColPairs& c, const float4 ra, float* __restrict__ orow, long ld, int rows = ROWS_CT) {
    const float rw = ra.z - ra.x, rh = ra.w - ra.y;
    const float rarea = rw * rh;                                      // lib/core.py:500-501
    auto one_row = [&](int r) {
        const float ax1 = bcast(ra.x, r), ay1 = bcast(ra.y, r), ax2 = bcast(ra.z, r), ay2 = bcast(ra.w, r);
        const float aw = bcast(rw, r), ah = bcast(rh, r), aarea = bcast(rarea, r);
        const gnms_f2 sx1 = {ax1, ax1}, sy1 = {ay1, ay1}, sx2 = {ax2, ax2}, sy2 = {ay2, ay2}, sa = {aarea, aarea};
        gnms_f2 q[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const gnms_f2 dx1 = sx2 - c.x1[p], dx2 = c.x2[p] - sx1;
            const gnms_f2 dy1 = sy2 - c.y1[p], dy2 = c.y2[p] - sy1;
            const gnms_f2 w = {hw_clamp0_s(hw_min3(dx1.x, dx2.x, c.w[p].x), aw), hw_clamp0_s(hw_min3(dx1.y, dx2.y, c.w[p].y), aw)};
            const gnms_f2 h = {hw_clamp0_s(hw_min3(dy1.x, dy2.x, c.h[p].x), ah), hw_clamp0_s(hw_min3(dy1.y, dy2.y, c.h[p].y), ah)};
            const gnms_f2 inter = w * h;                              // :218
            const gnms_f2 uni = (sa + c.area[p]) - inter;             // :507
            q[p] = div2_plain(inter, uni);                            // :508
        }
        store_nt_f4(orow, q[0].x, q[0].y, q[1].x, q[1].y);           // (ordinary stores, or a vmcnt throttle per row: both measured slower)
        orow += ld;
    };
    if constexpr (ROWS_CT > 0) {
#pragma unroll
        for (int r = 0; r < ROWS_CT; ++r) one_row(r);
    } else {
        for (int r = 0; r < rows; ++r) one_row(r);
    }
}

// The row loop of a wave tile: the column boxes (bx1 .. barea, 4 per lane) against the rows held one per lane in `ra` (lane r = row
// i0 + r, broadcast with v_readlane), one 16-byte (VEC) or four 4-byte stores per row.  `o` = the image's matrix.  ROWS_CT > 0: exactly
// that many rows, unrolled; 0: `rows` at run time.  ALLCOLS (with VEC): every column of the tile exists -- with ROWS_CT > 0 the body is
// then ONE basic block of ROWS_CT stores, so the compiler can count the stores in flight behind an earlier memory operation and wait
// for that one alone (writers_staged_2d).
template <bool VEC, int ROWS_CT, bool ALLCOLS = false, bool PLAINDIV = false>
__device__ __forceinline__ void iou2d_rows(const float (&bx1)[4], const float (&by1)[4], const float (&bx2)[4], const float (&by2)[4],
                                           const float (&barea)[4], const int (&col)[4], float4 ra, int rows, float* __restrict__ o, int i0,
                                           long ld, int N) {
    const float rarea = (ra.z - ra.x) * (ra.w - ra.y);               // lib/core.py:500-501
    float* orow = o + (size_t)i0 * ld;                               // advanced by one row per trip (not (i0 + r) * ld: two quarter-rate multiplies)
    auto one_row = [&](int r) {
        const float ax1 = bcast(ra.x, r), ay1 = bcast(ra.y, r), ax2 = bcast(ra.z, r), ay2 = bcast(ra.w, r);
        const float aarea = bcast(rarea, r);
        float res[4], inter[4], uni[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float w = relu0(hw_min_s(ax2, bx2[j]) - hw_max_s(ax1, bx1[j]));   // lib/core.py:210-212
            float h = relu0(hw_min_s(ay2, by2[j]) - hw_max_s(ay1, by1[j]));
            inter[j] = w * h;                                           // :218
            uni[j] = (aarea + barea[j]) - inter[j];                     // :507
        }
        if (PLAINDIV) {
            div4_plain(inter, uni, res);                                // :508, see above
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) res[j] = inter[j] / uni[j];     // :508
        }
        if (VEC) {
            if (ALLCOLS || col[3] < N) {
                store_nt_f4(orow + col[0], res[0], res[1], res[2], res[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (col[j] < N) orow[col[j]] = res[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (col[j] < N) orow[col[j]] = res[j];
        }
        orow += ld;
    };
    if (ROWS_CT > 0) {
#pragma unroll
        for (int r = 0; r < ROWS_CT; ++r) one_row(r);
    } else {
        for (int r = 0; r < rows; ++r) one_row(r);
    }
}

// One wave: rows i0..i0+63 of image `img` against the 256 columns starting at c0.  a [B][M][4], b [B][N][4], out [B][M][ld].
// VEC: ld % 4 == 0 and out 16-byte aligned -> lane owns columns c0+4*lane+{0..3}, one 16-B store per row; otherwise lane
// owns columns c0+lane+64*{0..3} and stores dwords (still coalesced).
template <bool VEC>
__device__ __forceinline__ void iou2d_tile(const float* __restrict__ A, const float* __restrict__ Bx, int M, int N,
                                           float* __restrict__ out, long ld, int img, int i0, int c0, int lane,
                                           int tile_rows = kTileRows, int row_end = 0x7fffffff) {
    if (row_end > M) row_end = M;                                 // a launch may cover the rows [row0, row_end) only
    if (c0 >= N || i0 >= row_end) return;
    const float* a = A + (size_t)img * M * 4;
    const float* b = Bx + (size_t)img * N * 4;
    float* o = out + (size_t)img * M * ld;

    // column boxes -> registers
    float bx1[4], by1[4], bx2[4], by2[4], barea[4];
    int col[4];
    bool plain = true;                                            // every box of the tile divides without scale / fixup (div4_plain)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        col[j] = VEC ? (c0 + 4 * lane + j) : (c0 + lane + 64 * j);
        int cc = col[j] < N ? col[j] : (N - 1);
        float4 v = *reinterpret_cast<const float4*>(b + (size_t)cc * 4);
        bx1[j] = v.x; by1[j] = v.y; bx2[j] = v.z; by2[j] = v.w;
        barea[j] = (v.z - v.x) * (v.w - v.y);                        // lib/core.py:502-503
        plain = plain && box_divides_plainly(v);
    }
    const int rows = min(tile_rows, row_end - i0);                // tile_rows <= 64: lane r holds row i0 + r

    // row boxes: lane r holds row i0+r; the row loop broadcasts it with v_readlane.  (Scalar loads of the row box --
    // s_load_dwordx4, also issued a row ahead -- measured 6 % slower: 100.7 vs 94.8 us at B=8, N=4096.)
    const int myrow = i0 + lane;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f);
    if (myrow < M) ra = *reinterpret_cast<const float4*>(a + (size_t)myrow * 4);
    plain = plain && box_divides_plainly(ra);                     // (the zero box of a lane past the last row passes)
    if (VEC && c0 + kWaveCols <= N && __all(plain)) {             // full tile of plain boxes: the packed row body
        ColPairs cp;
#pragma unroll
        for (int j = 0; j < 4; ++j) colpairs_set(cp, j, make_float4(bx1[j], by1[j], bx2[j], by2[j]));
        iou2d_rows_plain<0>(cp, ra, o + (size_t)i0 * ld + col[0], ld, rows);
    } else if (__all(plain)) iou2d_rows<VEC, 0, false, true>(bx1, by1, bx2, by2, barea, col, ra, rows, o, i0, ld, N);
    else iou2d_rows<VEC, 0>(bx1, by1, bx2, by2, barea, col, ra, rows, o, i0, ld, N);
}

// The same wave tile with its boxes already in LDS: column box of column c at scol[c - colbase], the ROWS_CT row boxes at srow[0 ..]
// -- no vector-memory load anywhere, so nothing in the wave waits for its own earlier stores and consecutive tiles stream back
// to back.  issue() runs after the LDS reads and before the first store, consume() after the last: a full tile (ROWS_CT rows, VEC,
// N % 4 == 0) is one basic block in between.
template <bool VEC, int ROWS_CT, typename Issue, typename Consume>
__device__ __forceinline__ void iou2d_tile_staged(const float4* scol, int colbase, const float4* srow, int M, int N, float* __restrict__ o,
                                                  long ld, int i0, int c0, int lane, Issue issue, Consume consume) {
    float bx1[4], by1[4], bx2[4], by2[4], barea[4];
    int col[4];
    bool plain = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        col[j] = VEC ? (c0 + 4 * lane + j) : (c0 + lane + 64 * j);
        const int cc = col[j] < N ? col[j] : (N - 1);
        const float4 v = scol[cc - colbase];
        bx1[j] = v.x; by1[j] = v.y; bx2[j] = v.z; by2[j] = v.w;
        barea[j] = (v.z - v.x) * (v.w - v.y);
        plain = plain && box_divides_plainly(v);
    }
    const int rows = min(ROWS_CT, M - i0);
    const float4 ra = srow[lane < rows ? lane : rows - 1];
    plain = plain && box_divides_plainly(ra);
    if (VEC && rows == ROWS_CT && (N & 3) == 0 && c0 + 4 * ROWS_CT <= N && __all(plain)) {
        // (N % 4 == 0: a lane's four columns exist together; the lanes past the last column of a ragged tile just sit the block out.
        // The first ROWS_CT lanes must NOT: the rows are broadcast out of their registers with v_readlane, and what a lane that sits a
        // branch out holds there is undefined -- the compiler sinks the LDS read of the row box into the branch.  A tile with fewer
        // than 4 * ROWS_CT columns takes the general path below.)
        if (col[0] < N) {
            ColPairs cp;
#pragma unroll
            for (int j = 0; j < 4; ++j) colpairs_set(cp, j, make_float4(bx1[j], by1[j], bx2[j], by2[j]));
            issue();
            iou2d_rows_plain<ROWS_CT>(cp, ra, o + (size_t)i0 * ld + col[0], ld);
            consume();
        }
    } else {
        issue();
        iou2d_rows<VEC, 0>(bx1, by1, bx2, by2, barea, col, ra, rows, o, i0, ld, N);
        consume();
    }
}

}  // namespace gnms_iou
