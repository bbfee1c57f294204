// iou_kernels.hip -- pairwise overlap matrices for gfx950 (MI355X).
//
// Reference semantics: lib/core.py:178-218 intersect, :480-508 iou (mode='combinations'),
// :305-421 iou3d_approximate (+ get_hull :423, get_volume :434, remove_rotation_in_boxes :463),
// lib/math_3d.py:364-435 get_corners_of_cuboid.
//
// Roofline class: HBM write stream (4*M*N bytes out, 16..32 bytes per box in).  One wave owns a
// 64-row x 256-column tile: the 4 column boxes of a lane live in registers for the whole tile,
// the row box is wave-uniform (v_readlane from the lane that loaded it), and every row of the tile
// leaves the wave as one 1-KiB coalesced global_store_dwordx4 (the 4 waves of a workgroup cover
// 4 KiB of one row).  Arithmetic follows the oracle/reference operation order exactly and the file
// is compiled with -ffp-contract=off, so the 2D matrix is bit-identical to torch's CPU result.
#include "gnms_prof.h"
#include "iou_tile.h"
#include "iou3d_pair.h"
#include "iou3d_tile.h"
#include "iou3d_sym.h"

namespace {

using namespace gnms_iou;

// ------------------------------------------------------------------------------------------------
// 2D IoU.  a [B][M][4], b [B][N][4], out [B][M][ld].
// VEC: ld % 4 == 0 and out 16-byte aligned -> lane owns columns c0+4*lane+{0..3}, one 16-B store per row;
// otherwise lane owns columns c0+lane+64*{0..3} and stores dwords (still coalesced).
// ------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kWavesPerWG * 64) void iou2d_kernel(const float* __restrict__ A, const float* __restrict__ Bx,
                                                                 int M, int N, float* __restrict__ out, long ld, int tile_rows, int row0,
                                                                 int row_end) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    iou2d_tile<VEC>(A, Bx, M, N, out, ld, blockIdx.z, row0 + blockIdx.y * tile_rows, blockIdx.x * kWGCols + wave * kWaveCols, lane, tile_rows,
                    row_end);
}

// ------------------------------------------------------------------------------------------------
// 3D: per-box axis-aligned record.  rec[12] = {vol, y0, y1, x0, x1, z0, z1, area_bev, len x, len y, len z, 0}
//   vol: product of the per-axis extents over all 8 corners (get_volume, lib/core.py:434-451)
//   y0,y1: min/max corner y (:365-368); x/z extents from corners {2,3,6,7} (:383-388, :463-476)
// ------------------------------------------------------------------------------------------------
using gnms_iou3d::kRec;

__device__ __forceinline__ void aabb_record(const float (&cx)[8], const float (&cy)[8], const float (&cz)[8], float* rec) {
    float mnx = cx[0], mxx = cx[0], mny = cy[0], mxy = cy[0], mnz = cz[0], mxz = cz[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        mnx = fminf(mnx, cx[k]); mxx = fmaxf(mxx, cx[k]);
        mny = fminf(mny, cy[k]); mxy = fmaxf(mxy, cy[k]);
        mnz = fminf(mnz, cz[k]); mxz = fmaxf(mxz, cz[k]);
    }
    float vol = ((mxx - mnx) * (mxy - mny)) * (mxz - mnz);
    float x0 = fminf(fminf(cx[2], cx[3]), fminf(cx[6], cx[7])), x1 = fmaxf(fmaxf(cx[2], cx[3]), fmaxf(cx[6], cx[7]));
    float z0 = fminf(fminf(cz[2], cz[3]), fminf(cz[6], cz[7])), z1 = fmaxf(fmaxf(cz[2], cz[3]), fmaxf(cz[6], cz[7]));
    rec[0] = vol; rec[1] = mny; rec[2] = mxy; rec[3] = x0; rec[4] = x1; rec[5] = z0; rec[6] = z1;
    rec[7] = (x1 - x0) * (z1 - z0);
    rec[8] = x1 - x0; rec[9] = mxy - mny; rec[10] = z1 - z0;                      // extents, used by the fast NMS-overlap kernel
    rec[11] = gnms_iou3d::record_bad_flag(x0, x1, mny, mxy, z0, z1, rec[8], rec[9], rec[10]);   // 0 = sane (iou3d_pair.h)
}


__global__ void aabb_from_corners_kernel(const float* __restrict__ corners, long count, float* __restrict__ rec) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float* c = corners + i * 24;
    float cx[8], cy[8], cz[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { cx[k] = c[k]; cy[k] = c[8 + k]; cz[k] = c[16 + k]; }
    float r[kRec];
    aabb_record(cx, cy, cz, r);
    float4* o = reinterpret_cast<float4*>(rec + i * kRec);
    o[0] = make_float4(r[0], r[1], r[2], r[3]);
    o[1] = make_float4(r[4], r[5], r[6], r[7]);
    o[2] = make_float4(r[8], r[9], r[10], r[11]);
}

// get_corners_of_cuboid, lib/math_3d.py:364-435 (same operation order as oracle/gnms_oracle.c)
__device__ __forceinline__ void corners_of(const float* p, float (&cx)[8], float (&cy)[8], float (&cz)[8]) {
    const float x = p[0], y = p[1], z = p[2], w = p[3], h = p[4], l = p[5], ry = p[6];
    const float c = cosf(ry), s = sinf(ry);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool xh = (k == 1) | (k == 3) | (k == 5) | (k == 6);   // :401
        const bool yh = (k == 2) | (k == 3) | (k == 6) | (k == 7);   // :402
        const bool zh = k >= 4;                                      // :403
        float bx = (xh ? l : 0.0f) - l / 2;
        float by = (yh ? h : 0.0f) - h / 2;
        float bz = (zh ? w : 0.0f) - w / 2;
        float rx = c * bx + 0.0f * by + s * bz;                      // bmm(R, corners) :430
        float ryy = 0.0f * bx + 1.0f * by + 0.0f * bz;
        float rz = (-s) * bx + 0.0f * by + c * bz;
        cx[k] = rx + x; cy[k] = ryy + y; cz[k] = rz + z;             // :433-435
    }
}

__global__ void corners_kernel(const float* __restrict__ params, long count, float* __restrict__ corners) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float cx[8], cy[8], cz[8];
    corners_of(params + i * 7, cx, cy, cz);
    float* c = corners + i * 24;
#pragma unroll
    for (int k = 0; k < 8; ++k) { c[k] = cx[k]; c[8 + k] = cy[k]; c[16 + k] = cz[k]; }
}

__global__ void aabb_from_params_kernel(const float* __restrict__ params, long count, float* __restrict__ rec) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float cx[8], cy[8], cz[8];
    corners_of(params + i * 7, cx, cy, cz);
    float r[kRec];
    aabb_record(cx, cy, cz, r);
    float4* o = reinterpret_cast<float4*>(rec + i * kRec);
    o[0] = make_float4(r[0], r[1], r[2], r[3]);
    o[1] = make_float4(r[4], r[5], r[6], r[7]);
    o[2] = make_float4(r[8], r[9], r[10], r[11]);
}

// gnms_forward_with_iou3d: the records once contiguous (the matrix kernel's input), once into the per-image workspace regions (the
// layer's copy), + the pseudo boxes (x0, lx, x1, z0 + z1) its column sort orders the columns by (z band, then x centre) -- one pass over
// the cuboids
__global__ __launch_bounds__(256) void aabb_for_layer_kernel(const float* __restrict__ params, int N, float* __restrict__ rec, char* ws,
                                                             gnms_ws_layout L, float4* __restrict__ xkeys) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const size_t g = (size_t)b * N + i;
    float cx[8], cy[8], cz[8];
    corners_of(params + g * 7, cx, cy, cz);
    float r[kRec];
    aabb_record(cx, cy, cz, r);
    const float4 u = make_float4(r[0], r[1], r[2], r[3]), v = make_float4(r[4], r[5], r[6], r[7]), e = make_float4(r[8], r[9], r[10], r[11]);
    float4* o = reinterpret_cast<float4*>(rec + g * kRec);
    o[0] = u; o[1] = v; o[2] = e;
    float4* w = reinterpret_cast<float4*>(ws + (size_t)b * L.per_image + L.off_rec) + (size_t)i * 3;
    w[0] = u; w[1] = v; w[2] = e;
    xkeys[g] = make_float4(r[3], r[8], r[4], r[5] + r[6]);
}

// pairwise 3D overlap from the records, the reference's operation order (lib/core.py:305-421), bit for bit the CPU oracle's result.
// METHOD 0 normal, 1 generalized, 2 0.5*(1+generalized); BEV: iou_bev as a second output of the same pass.
//
// Round 4: the PLAIN body.  The reference's expression costs ~64 VALU slots per pair as written (three IEEE divisions at 10 slots,
// three axes of min / max / subtract / relu twice over) and ran at 0.30-0.35 of the HBM peak; what the 2D tile learned (iou_tile.h)
// carries over for tiles whose boxes are all SANE (record[11] == 0: extents in (1e-4, 1e5), |coordinates| < 1e5) and free of negative
// zeros:
//   * per axis the overlap  relu(min(a1, b1) - max(a0, b0))  and the hull extent  relu(max(a1, b1) - min(a0, b0))  are the smallest /
//     largest of the SAME four differences a1 - a0, b1 - b0, a1 - b0, b1 - a0; rounding is monotone, so the rounded result of the
//     subtraction the reference performs is the min / max of the four rounded differences: two packed subtractions per column pair
//     (the other two are the extents the records carry), v_min3 + v_med3 for the overlap (clamped to [0, row extent]), v_max3 + v_max
//     for the hull (>= the row extent > 0: the relu is the identity) -- 5 slots per entry and axis instead of 8;
//   * the divisions are the compiler's fp32 division without V_DIV_SCALE / V_DIV_FIXUP, packed (div2_plain): exact wherever no operand,
//     reciprocal or quotient comes near the ends of the exponent range.  Sane boxes bound every denominator (u3 >= the larger volume
//     > 1e-12, hull volume in (1e-12, 1e17), bird's-eye union in (1e-8, 1e11)) and the numerator vh - u3 (0 or >= an ulp of vh); only
//     the intersections i3 and inter can be arbitrarily small, and a row with one in (0, 2^-60) is evaluated again with IEEE divisions
//     (an integer min over the row's bit patterns and one compare decide that).
// ~31 slots per pair (generalized), ~38 with iou_bev; any other tile (a box that is not sane, a negative zero) takes the scalar body.
// ------------------------------------------------------------------------------------------------
using gnms_iou3d::f2;
using gnms_iou3d::splat;
struct ExCols {                                                     // a lane's four columns as two packed pairs
    f2 x0[2], x1[2], y0[2], y1[2], z0[2], z1[2], vol[2], area[2], lx[2], ly[2], lz[2];
};
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max_vs(float x, float s) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(x)); return r; }
__device__ __forceinline__ bool is_neg_zero(float c) { return __float_as_uint(c) == 0x80000000u; }

template <bool VEC, int METHOD, bool BEV>
__global__ __launch_bounds__(kWavesPerWG * 64) void iou3d_kernel(const float* __restrict__ RA, const float* __restrict__ RB, int M, int N,
                                                    float* __restrict__ out_bev, float* __restrict__ out3d, long ld, int tile_rows) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int img = blockIdx.z;
    const int i0 = blockIdx.y * tile_rows;
    const int c0 = blockIdx.x * kWGCols + wave * kWaveCols;
    if (c0 >= N) return;
    const float* ra = RA + (size_t)img * M * kRec;
    const float* rb = RB + (size_t)img * N * kRec;
    float* o3 = out3d + (size_t)img * M * ld;
    float* ob = BEV ? out_bev + (size_t)img * M * ld : nullptr;

    int col[4];
    ExCols c;                                                       // (the scalar body reads the same registers, element j at [j >> 1][j & 1])
    bool cplain = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        col[j] = VEC ? (c0 + 4 * lane + j) : (c0 + lane + 64 * j);
        int cc = col[j] < N ? col[j] : (N - 1);
        const float4* p = reinterpret_cast<const float4*>(rb + (size_t)cc * kRec);
        float4 u = p[0], v = p[1], e = p[2];
        c.vol[j >> 1][j & 1] = u.x; c.y0[j >> 1][j & 1] = u.y; c.y1[j >> 1][j & 1] = u.z; c.x0[j >> 1][j & 1] = u.w; c.x1[j >> 1][j & 1] = v.x;
        c.z0[j >> 1][j & 1] = v.y; c.z1[j >> 1][j & 1] = v.z; c.area[j >> 1][j & 1] = v.w; c.lx[j >> 1][j & 1] = e.x; c.ly[j >> 1][j & 1] = e.y;
        c.lz[j >> 1][j & 1] = e.z;
        cplain = cplain && e.w == 0.0f && !is_neg_zero(u.y) && !is_neg_zero(u.z) && !is_neg_zero(u.w) && !is_neg_zero(v.x) && !is_neg_zero(v.y) && !is_neg_zero(v.z);
    }
    const int myrow = i0 + lane;
    float4 ru = make_float4(0.f, 0.f, 0.f, 0.f), rv = ru, re = ru;
    if (myrow < M) {
        const float4* p = reinterpret_cast<const float4*>(ra + (size_t)myrow * kRec);
        ru = p[0]; rv = p[1]; re = p[2];
    }
    const int rows = min(tile_rows, M - i0);
    const bool rplain = (lane >= rows) || (re.w == 0.0f && !is_neg_zero(ru.y) && !is_neg_zero(ru.z) && !is_neg_zero(ru.w) && !is_neg_zero(rv.x) &&
                                            !is_neg_zero(rv.y) && !is_neg_zero(rv.z));
    const bool plain = VEC && __all(cplain && rplain) && (c0 + 4 * 64 <= N);        // (a full tile: every lane owns four existing columns)

    // one row in the reference's scalar order (every tile that is not plain; the rare rows of a plain tile with a tiny intersection)
    auto scalar_row = [&](int r) {
        const float avol = bcast(ru.x, r), ay0 = bcast(ru.y, r), ay1 = bcast(ru.z, r), ax0 = bcast(ru.w, r);
        const float ax1 = bcast(rv.x, r), az0 = bcast(rv.y, r), az1 = bcast(rv.z, r), aar = bcast(rv.w, r);
        float res3[4], resb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bvol = c.vol[j >> 1][j & 1], by0 = c.y0[j >> 1][j & 1], by1 = c.y1[j >> 1][j & 1], bx0 = c.x0[j >> 1][j & 1];
            const float bx1 = c.x1[j >> 1][j & 1], bz0 = c.z0[j >> 1][j & 1], bz1 = c.z1[j >> 1][j & 1], bar = c.area[j >> 1][j & 1];
            float vol = avol + bvol;                                          // lib/core.py:357
            float yi = relu0(fminf(ay1, by1) - fmaxf(ay0, by0));              // :371-376
            float w = relu0(fminf(ax1, bx1) - fmaxf(ax0, bx0));               // intersect(bev) :410
            float h = relu0(fminf(az1, bz1) - fmaxf(az0, bz0));
            float inter = w * h;
            if (BEV) resb[j] = inter / ((aar + bar) - inter);                 // iou(bev) :408
            float i3 = inter * yi;                                            // :415
            float u3 = vol - i3;                                              // :416
            float q = i3 / u3;                                                // :417
            if (METHOD >= 1) {                                                // :390-406, :418-419
                float xh = relu0(fmaxf(ax1, bx1) - fminf(ax0, bx0));
                float yh = relu0(fmaxf(ay1, by1) - fminf(ay0, by0));
                float zh = relu0(fmaxf(az1, bz1) - fminf(az0, bz0));
                float vh = (xh * yh) * zh;
                q = q - ((vh - u3) / vh);
            }
            if (METHOD == 2) q = 0.5f * (1.0f + q);                           // lib/loss/rpn_3d.py:781
            res3[j] = q;
        }
        const size_t roff = (size_t)(i0 + r) * ld;
        if (VEC && col[3] < N) {
            store_nt_f4(o3 + roff + col[0], res3[0], res3[1], res3[2], res3[3]);
            if (BEV) store_nt_f4(ob + roff + col[0], resb[0], resb[1], resb[2], resb[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (col[j] < N) { o3[roff + col[j]] = res3[j]; if (BEV) ob[roff + col[j]] = resb[j]; }
        }
    };
    if (!plain) {
        for (int r = 0; r < rows; ++r) scalar_row(r);
        return;
    }
    unsigned long long redo = 0ull;                                                        // rows of this tile to evaluate again with IEEE divisions
    for (int r = 0; r < rows; ++r) {
        const float* rr = ra + (size_t)(i0 + r) * kRec;                       // wave-uniform: scalar loads
        const float avol = rr[0], ay0 = rr[1], ay1 = rr[2], ax0 = rr[3], ax1 = rr[4], az0 = rr[5], az1 = rr[6], aar = rr[7];
        const float alx = rr[8], aly = rr[9], alz = rr[10];
        f2 q3[2], qb[2];
        unsigned tiny = 0xffffffffu;                                          // min over the row of (bits of the intersections) - 1
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f2 dx1 = splat(ax1) - c.x0[p], dx2 = c.x1[p] - splat(ax0);
            const f2 dy1 = splat(ay1) - c.y0[p], dy2 = c.y1[p] - splat(ay0);
            const f2 dz1 = splat(az1) - c.z0[p], dz2 = c.z1[p] - splat(az0);
            const f2 w = {gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dx1.x, dx2.x, c.lx[p].x), alx), gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dx1.y, dx2.y, c.lx[p].y), alx)};
            const f2 yi = {gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dy1.x, dy2.x, c.ly[p].x), aly), gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dy1.y, dy2.y, c.ly[p].y), aly)};
            const f2 h = {gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dz1.x, dz2.x, c.lz[p].x), alz), gnms_iou::hw_clamp0_s(gnms_iou::hw_min3(dz1.y, dz2.y, c.lz[p].y), alz)};
            const f2 inter = w * h;                                           // :410-412
            if (BEV) {
                qb[p] = gnms_iou::div2_plain(inter, (splat(aar) + c.area[p]) - inter);   // :408
                tiny = min(tiny, min(__float_as_uint(inter.x) - 1u, __float_as_uint(inter.y) - 1u));
            }
            const f2 i3 = inter * yi;                                         // :415
            const f2 u3 = (splat(avol) + c.vol[p]) - i3;                      // :357, :416
            tiny = min(tiny, min(__float_as_uint(i3.x) - 1u, __float_as_uint(i3.y) - 1u));
            f2 q = gnms_iou::div2_plain(i3, u3);                              // :417
            if (METHOD >= 1) {
                const f2 xh = {hw_max_vs(hw_max3(dx1.x, dx2.x, c.lx[p].x), alx), hw_max_vs(hw_max3(dx1.y, dx2.y, c.lx[p].y), alx)};
                const f2 yh = {hw_max_vs(hw_max3(dy1.x, dy2.x, c.ly[p].x), aly), hw_max_vs(hw_max3(dy1.y, dy2.y, c.ly[p].y), aly)};
                const f2 zh = {hw_max_vs(hw_max3(dz1.x, dz2.x, c.lz[p].x), alz), hw_max_vs(hw_max3(dz1.y, dz2.y, c.lz[p].y), alz)};
                const f2 vh = (xh * yh) * zh;                                 // :390-406
                q = q - gnms_iou::div2_plain(vh - u3, vh);                    // :418-419
            }
            if (METHOD == 2) q = (f2){0.5f, 0.5f} * ((f2){1.0f, 1.0f} + q);   // lib/loss/rpn_3d.py:781
            q3[p] = q;
        }
        if (__any(tiny < 0x21800000u - 1u)) redo |= 1ull << r;                // an intersection in (0, 2^-60): the row again, below
        float* o = o3 + (size_t)(i0 + r) * ld + col[0];
        store_nt_f4(o, q3[0].x, q3[0].y, q3[1].x, q3[1].y);
        if (BEV) store_nt_f4(ob + (size_t)(i0 + r) * ld + col[0], qb[0].x, qb[0].y, qb[1].x, qb[1].y);
    }
    while (redo) {                                                            // (wave-uniform; rare)
        const int r = __builtin_ctzll(redo);
        redo &= redo - 1ull;
        scalar_row(r);
    }
}

// ------------------------------------------------------------------------------------------------
// The NMS overlap 0.5 * (1 + GIoU3D) of lib/loss/rpn_3d.py:781 / lib/rpn_util.py:1312 for a whole matrix, arithmetic
// re-associated so that the kernel is bound by the HBM write stream instead of by two IEEE divisions per pair:
//     0.5 * (1 + i3/u3 - (vh - u3)/vh)  =  0.5 * (i3*vh + u3*u3) / (u3*vh)        one reciprocal (v_rcp_f32, 1 ulp)
//     hull extent per axis  max(a1,b1) - min(a0,b0)  =  (lenA + lenB) - (min(a1,b1) - max(a0,b0))   reuses the overlap's d
// and every add/mul runs two columns at a time (v_pk_*_f32).  The row records come from LDS (broadcast reads, no VALU).
// About 24 VALU slots per pair instead of 43.  Result within 1e-6 of the exact expression order (tested at 1e-5 against
// the reference vectors; north_star tolerance 1e-4); gnms_iou3d_approximate / methods 0 and 1 keep the exact kernel above.
// ------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kWavesPerWG * 64) __attribute__((amdgpu_waves_per_eu(7))) void iou3d_nms_fast_kernel(const float* __restrict__ RA, const float* __restrict__ RB, int M,
                                                                          int N, float* __restrict__ out, long ld, int tile_rows, int row0,
                                                                          int row_end, float thr) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    gnms_iou3d::nms_overlap3d_tile<VEC>(RA, RB, M, N, out, ld, blockIdx.z, row0 + blockIdx.y * tile_rows, blockIdx.x * kWGCols + wave * kWaveCols,
                                        lane, tile_rows, row_end, thr);
}

// The same matrix for ONE box set with itself, every unordered pair evaluated once (iou3d_sym.h): one 8-wave workgroup per
// 128 x 128 macro tile of the upper triangle, tile ids [tile0, tile0 + gridDim.x) of every image.
template <int NW, bool NT>
__global__ __launch_bounds__(NW * 64) void iou3d_sym_kernel(const float* __restrict__ rec, int N, float* __restrict__ out, long ld, float thr, int tile0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int I, J;
    gnms_iou3d::sym_tile_of(tile0 + (int)blockIdx.x, (N + gnms_iou3d::kSymT - 1) / gnms_iou3d::kSymT, &I, &J);
    const int b = blockIdx.z;
    gnms_iou3d::nms_overlap3d_sym_tile<NW, NT>(rec + (size_t)b * N * kRec, N, out + (size_t)b * N * ld, ld, I, J, thr, reinterpret_cast<float*>(smem));
}

// The same writer as PERSISTENT workgroups (one 16-wave workgroup per CU): workgroup w takes the macro tiles w, w + G, w + 2 G, ... of
// the batch (image-major, row-major over each triangle) -- together the G workgroups advance one frontier through the matrices -- and
// alternates between two LDS tiles, so that one barrier per tile suffices and the mirrored stores of a tile overlap the arithmetic of
// the next (the layer's writers_sym_persistent, nms_layer.hip, with a static round robin instead of claims: no counter to reset).
template <bool NT>
__global__ __launch_bounds__(1024) void iou3d_sym_persistent_kernel(const float* __restrict__ rec, int N, int nimg, float* __restrict__ out, long ld,
                                                                    float thr) {
    using namespace gnms_iou3d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tile0 = reinterpret_cast<float*>(smem);
    const int nt = (N + kSymT - 1) / kSymT;
    const int tpi = sym_tiles_per_image(N);
    const long total = (long)tpi * nimg;
    int ph = 0;
    for (long cur = blockIdx.x; cur < total; cur += gridDim.x) {
        const int img = (int)(cur / tpi);
        int I, J;
        sym_tile_of((int)(cur - (long)img * tpi), nt, &I, &J);
        const float* r = rec + (size_t)img * N * kRec;
        float* o = out + (size_t)img * N * ld;
        float* const tile = tile0 + (size_t)ph * (kSymTileBytes / sizeof(float));
        sym_tile_compute<16, NT>(r, N, o, ld, I, J, thr, tile);
        __syncthreads();                                            // the tile is in LDS; the other buffer is free again
        if (I != J) sym_tile_mirror<16, NT>(N, o, ld, I, J, tile);
        ph ^= 1;
    }
}

template <bool VEC, int METHOD>
void launch_iou3d(const float* ra, const float* rb, int B, int M, int N, float* bev, float* o3, long ld, hipStream_t st) {
    const int tr = tile_rows_for(B, M, N);
    dim3 grid(gnms_div_up(N, kWGCols), gnms_div_up(M, tr), B);
    if (bev) gnms_launch_prof(kProfMatrixWrite, iou3d_kernel<VEC, METHOD, true>, grid, dim3(kWavesPerWG * 64), 0, st, ra, rb, M, N, bev, o3, ld, tr);
    else gnms_launch_prof(kProfMatrixWrite, iou3d_kernel<VEC, METHOD, false>, grid, dim3(kWavesPerWG * 64), 0, st, ra, rb, M, N, (float*)nullptr, o3, ld, tr);
}

int iou3d_from_records(const float* ra, const float* rb, int B, int M, int N, int method, float* bev, float* o3, int64_t ld,
                       hipStream_t st, bool fast_nms_overlap, int row0 = 0, int row_end = 0x7fffffff, float guard_thr = 0.0f) {
    const bool vec = (ld % 4 == 0) && ((uintptr_t)o3 % 16 == 0) && (!bev || (uintptr_t)bev % 16 == 0);
    if (method == 2 && !bev && fast_nms_overlap) {
        const int tr = tile_rows_for(B, M, N);
        if (row_end > M) row_end = M;
        if (row0 >= row_end) return GNMS_OK;
        dim3 grid(gnms_div_up(N, kWGCols), gnms_div_up(row_end - row0, tr), B);
        if (vec) gnms_launch_prof(kProfMatrixWrite, iou3d_nms_fast_kernel<true>, grid, dim3(kWavesPerWG * 64), 0, st, ra, rb, M, N, o3, (long)ld, tr, row0, row_end, guard_thr);
        else gnms_launch_prof(kProfMatrixWrite, iou3d_nms_fast_kernel<false>, grid, dim3(kWavesPerWG * 64), 0, st, ra, rb, M, N, o3, (long)ld, tr, row0, row_end, guard_thr);
        GNMS_CHECK_LAUNCH();
        return GNMS_OK;
    }
    if (vec) {
        if (method == 0) launch_iou3d<true, 0>(ra, rb, B, M, N, bev, o3, ld, st);
        else if (method == 1) launch_iou3d<true, 1>(ra, rb, B, M, N, bev, o3, ld, st);
        else launch_iou3d<true, 2>(ra, rb, B, M, N, bev, o3, ld, st);
    } else {
        if (method == 0) launch_iou3d<false, 0>(ra, rb, B, M, N, bev, o3, ld, st);
        else if (method == 1) launch_iou3d<false, 1>(ra, rb, B, M, N, bev, o3, ld, st);
        else launch_iou3d<false, 2>(ra, rb, B, M, N, bev, o3, ld, st);
    }
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

}  // namespace

// used by gnms_forward_with_iou3d (nms_layer.hip): per-box records from cuboid parameters, and the NMS-overlap matrix from records
int gnms_internal_records_from_params(const float* params, long count, float* rec, hipStream_t st) {
    if (count <= 0) return GNMS_OK;
    aabb_from_params_kernel<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(params, count, rec);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
int gnms_internal_records_for_layer(const float* params, int B, int N, float* rec, char* ws, const gnms_ws_layout& L, float* xkeys,
                                    hipStream_t st) {
    if (B <= 0 || N <= 0) return GNMS_OK;
    aabb_for_layer_kernel<<<dim3(gnms_div_up(N, 256), B), 256, 0, st>>>(params, N, rec, ws, L, reinterpret_cast<float4*>(xkeys));
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
int gnms_internal_nms_overlap3d(const float* rec, int B, int N, float* out, int64_t ld, hipStream_t st, float thr, int row0, int row_end) {
    return iou3d_from_records(rec, rec, B, N, N, 2, nullptr, out, ld, st, true, row0, row_end, thr);
}
// The symmetric writer (iou3d_sym.h) for the square matrix of one box set: the macro tiles [pct0, pct1) percent of every image's
// upper triangle (a call may split the write over two launches / streams).  Needs ld even and `out` 8-byte aligned.
bool gnms_internal_overlap3d_sym_ok(int N, int64_t ld, const float* out) {
    return N >= 256 && (ld % 2 == 0) && ((uintptr_t)out % 8 == 0);
}
int gnms_internal_nms_overlap3d_sym(const float* rec, int B, int N, float* out, int64_t ld, hipStream_t st, float thr, int pct0, int pct1,
                                    int leave_cus, int force_persist) {
    const int tiles = gnms_iou3d::sym_tiles_per_image(N);
    const int t0 = (int)((long long)tiles * pct0 / 100), t1 = (int)((long long)tiles * pct1 / 100);
    if (t1 <= t0 || B <= 0) return GNMS_OK;
    // (non-temporal stores throughout: measured faster than ordinary ones)
    // waves per workgroup: 16 up to N = 4096 (0.104 against 0.113 ms at B = 8: the guard band's exact-order branch costs 8 % with 8 waves and
    // nothing with 16), 8 above (N = 16384: 1.645 against 1.666 ms)
    const int nw = N <= 4096 ? 16 : 8;
    int rc;
    // persistent workgroups (iou3d_sym_persistent_kernel) for the largest images only: B = 8, N = 16384 1.63 -> 1.565 ms (0.69 of the HBM
    // peak); at N = 4096 (0.104 -> 0.119) and 8192 (0.38 -> 0.42, the one-tile-per-workgroup kernel reaches 0.71 there) the static round
    // robin ends unevenly.
    const bool persist = N > 8192 || force_persist;
    if (persist && pct0 == 0 && pct1 == 100) {
        const size_t lds2 = 2 * gnms_iou3d::kSymTileBytes;
        const int cus = gnms_device_cu_count();                       // (cached per device: nms_layer.hip)
        if ((rc = gnms_allow_lds_raw(reinterpret_cast<const void*>(iou3d_sym_persistent_kernel<true>), lds2))) return rc;
        gnms_launch_prof(kProfMatrixWrite, iou3d_sym_persistent_kernel<true>, dim3((unsigned)std::max(1, cus - leave_cus)), dim3(1024), lds2, st, rec, N, B, out, (long)ld, thr);
        GNMS_CHECK_LAUNCH();
        return GNMS_OK;
    }
    const size_t lds = gnms_iou3d::kSymTileBytes;
    const dim3 grid((unsigned)(t1 - t0), 1, (unsigned)B);
#define GNMS_SYM_LAUNCH(NW_, NT_)                                                                                                     \
    do {                                                                                                                              \
        if ((rc = gnms_allow_lds_raw(reinterpret_cast<const void*>(iou3d_sym_kernel<NW_, NT_>), lds))) return rc;                     \
        gnms_launch_prof(kProfMatrixWrite, iou3d_sym_kernel<NW_, NT_>, grid, dim3(NW_ * 64), lds, st, rec, N, out, (long)ld, thr, t0); \
    } while (0)
    if (nw == 16) GNMS_SYM_LAUNCH(16, true);
    else GNMS_SYM_LAUNCH(8, true);
#undef GNMS_SYM_LAUNCH
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
// rows [row0, row_end) of every image's square 2D IoU matrix (arguments already checked by the caller)
int gnms_internal_iou2d_rows(const float* boxes, int B, int N, float* out, int64_t ld, hipStream_t st, int row0, int row_end) {
    const int tr = tile_rows_for(B, N, N);
    if (row_end > N) row_end = N;
    if (row0 >= row_end) return GNMS_OK;
    dim3 grid(gnms_div_up(N, kWGCols), gnms_div_up(row_end - row0, tr), B);
    const bool vec = (ld % 4 == 0) && ((uintptr_t)out % 16 == 0);
    if (vec) gnms_launch_prof(kProfMatrixWrite, iou2d_kernel<true>, grid, dim3(kWavesPerWG * 64), 0, st, boxes, boxes, N, N, out, (long)ld, tr, row0, row_end);
    else gnms_launch_prof(kProfMatrixWrite, iou2d_kernel<false>, grid, dim3(kWavesPerWG * 64), 0, st, boxes, boxes, N, N, out, (long)ld, tr, row0, row_end);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// defined in nms_layer.hip (write_staged_kernel)
bool gnms_internal_iou2d_wants_staged(int B, int M, int N, int64_t ld, const float* out);
int gnms_internal_iou2d_staged(const float* a, const float* b, int B, int M, int N, float* out, int64_t ld, hipStream_t st);
bool gnms_internal_iou2d_wants_self(const float* a, const float* b, int B, int M, int N, int64_t ld, const float* out);
int gnms_internal_iou2d_self(const float* boxes, int B, int N, float* out, int64_t ld, hipStream_t st);

extern "C" int gnms_iou2d(const float* boxes_a, const float* boxes_b, int B, int M, int N, float* out, int64_t ld,
                          void* stream) {
    GNMS_CHECK_ARG(B >= 0 && M >= 0 && N >= 0, "gnms_iou2d: negative size (B=%d M=%d N=%d)", B, M, N);
    if (B == 0 || M == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(boxes_a && boxes_b && out, "gnms_iou2d: null pointer");
    GNMS_CHECK_ARG(ld >= N, "gnms_iou2d: ld (%lld) < N (%d)", (long long)ld, N);
    GNMS_CHECK_ARG(((uintptr_t)boxes_a % 16 == 0) && ((uintptr_t)boxes_b % 16 == 0), "gnms_iou2d: boxes must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (gnms_internal_iou2d_wants_self(boxes_a, boxes_b, B, M, N, ld, out)) {
        const int rc = gnms_internal_iou2d_self(boxes_a, B, N, out, ld, st);
        if (rc <= 0) return rc;                                     // (1: no claim slot for this stream / capture -- the tiles below)
    }
    if (gnms_internal_iou2d_wants_staged(B, M, N, ld, out)) return gnms_internal_iou2d_staged(boxes_a, boxes_b, B, M, N, out, ld, st);
    const int tr = tile_rows_for(B, M, N);
    dim3 grid(gnms_div_up(N, kWGCols), gnms_div_up(M, tr), B);
    const bool vec = (ld % 4 == 0) && ((uintptr_t)out % 16 == 0);
    if (vec) gnms_launch_prof(kProfMatrixWrite, iou2d_kernel<true>, grid, dim3(kWavesPerWG * 64), 0, st, boxes_a, boxes_b, M, N, out, (long)ld, tr, 0, M);
    else gnms_launch_prof(kProfMatrixWrite, iou2d_kernel<false>, grid, dim3(kWavesPerWG * 64), 0, st, boxes_a, boxes_b, M, N, out, (long)ld, tr, 0, M);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

extern "C" int gnms_corners_of_cuboid(const float* params, int64_t count, float* corners, void* stream) {
    GNMS_CHECK_ARG(count >= 0, "gnms_corners_of_cuboid: negative count");
    if (count == 0) return GNMS_OK;
    GNMS_CHECK_ARG(params && corners, "gnms_corners_of_cuboid: null pointer");
    corners_kernel<<<(unsigned)((count + 255) / 256), 256, 0, (hipStream_t)stream>>>(params, (long)count, corners);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// ------------------------------------------------------------------------------------------------
// The float64 NumPy branches the reference's INFERENCE call site takes (lib/rpn_util.py:1292-1320): `aboxes` is float64 after the
// hstack at :1258, so lib/core.py:205-207, 512-513 build the IoU matrix in float64 and only lib/groomed_nms.py:36 rounds it to fp32,
// once; lib/math_3d.py:438-490 builds the corners in float64 (np.einsum) before `.float()`.  An fp32 evaluation differs from that in
// the last ulp of some entries, under a strict `> nms_threshold`.  These two kernels run the same IEEE double operations in the same
// order (compiled -ffp-contract=off): the matrix is bit-identical to NumPy's, the corners up to libm's sin / cos (a sub-ulp of double,
// gone after the caller's rounding to fp32).  Throughput is irrelevant here (N = 500 at that call site).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iou2d_f64_kernel(const double* __restrict__ A, const double* __restrict__ Bx, int M, int N,
                                                        double* __restrict__ out, long ld) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= N) return;
    const double* a = A + ((size_t)b * M + i) * 4;
    const double* c = Bx + ((size_t)b * N + j) * 4;
    const double ax1 = a[0], ay1 = a[1], ax2 = a[2], ay2 = a[3];
    const double bx1 = c[0], by1 = c[1], bx2 = c[2], by2 = c[3];
    const double area_a = (ax2 - ax1) * (ay2 - ay1);                              // lib/core.py:499-500
    const double area_b = (bx2 - bx1) * (by2 - by1);                              // :502-503
    // :205-207.  (np.minimum / np.maximum / np.clip propagate a NaN coordinate where fmin / fmax drop it -- but such a box has a NaN
    // area, hence a NaN union and a NaN quotient either way, exactly as in the fp32 tile, iou_tile.h)
    const double w = fmax(fmin(ax2, bx2) - fmax(ax1, bx1), 0.0);
    const double h = fmax(fmin(ay2, by2) - fmax(ay1, by1), 0.0);
    const double inter = w * h;                                                   // :218
    out[((size_t)b * M + i) * ld + j] = inter / ((area_a + area_b) - inter);      // :512-513
}

__global__ void corners_f64_kernel(const double* __restrict__ params, long count, int trig_f32, double* __restrict__ corners) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double* p = params + i * 7;
    const double x = p[0], y = p[1], z = p[2], w = p[3], h = p[4], l = p[5], ry = p[6];
    // lib/math_3d.py:443-447: np.cos / np.sin evaluate in the dtype of `ry3d` -- float32 at the inference call site (coords_3d_raw is the
    // network's fp32 output, lib/rpn_util.py:1186,1211) -- and the result is widened into the float64 matrix R
    const double c = trig_f32 ? (double)cosf((float)ry) : cos(ry), s = trig_f32 ? (double)sinf((float)ry) : sin(ry);
    double* o = corners + i * 24;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool xh = (k == 1) | (k == 3) | (k == 5) | (k == 6);               // :470
        const bool yh = (k == 2) | (k == 3) | (k == 6) | (k == 7);               // :471
        const bool zh = k >= 4;                                                  // :472
        const double bx = (xh ? l : 0.0) - l / 2, by = (yh ? h : 0.0) - h / 2, bz = (zh ? w : 0.0) - w / 2;   // :474-476
        // np.einsum('ijk,ikl->ijl', R, corners) :480: sum over k in ascending order
        const double rx = (c * bx + 0.0 * by) + s * bz;
        const double ryy = (0.0 * bx + 1.0 * by) + 0.0 * bz;
        const double rz = ((-s) * bx + 0.0 * by) + c * bz;
        o[k] = rx + x; o[8 + k] = ryy + y; o[16 + k] = rz + z;                   // :483-485
    }
}

extern "C" int gnms_iou2d_f64(const double* boxes_a, const double* boxes_b, int B, int M, int N, double* out, int64_t ld, void* stream) {
    GNMS_CHECK_ARG(B >= 0 && M >= 0 && N >= 0, "gnms_iou2d_f64: negative size (B=%d M=%d N=%d)", B, M, N);
    if (B == 0 || M == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(boxes_a && boxes_b && out, "gnms_iou2d_f64: null pointer");
    GNMS_CHECK_ARG(ld >= N, "gnms_iou2d_f64: ld (%lld) < N (%d)", (long long)ld, N);
    GNMS_CHECK_ARG(M <= 65535 && B <= 65535, "gnms_iou2d_f64: M and B must be <= 65535");
    iou2d_f64_kernel<<<dim3(gnms_div_up(N, 256), M, B), 256, 0, (hipStream_t)stream>>>(boxes_a, boxes_b, M, N, out, (long)ld);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

extern "C" int gnms_corners_of_cuboid_f64(const double* params, int64_t count, int trig_f32, double* corners, void* stream) {
    GNMS_CHECK_ARG(count >= 0, "gnms_corners_of_cuboid_f64: negative count");
    if (count == 0) return GNMS_OK;
    GNMS_CHECK_ARG(params && corners, "gnms_corners_of_cuboid_f64: null pointer");
    corners_f64_kernel<<<(unsigned)((count + 255) / 256), 256, 0, (hipStream_t)stream>>>(params, (long)count, trig_f32, corners);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// ------------------------------------------------------------------------------------------------
// "projected" 2D boxes of the NMS (lib/loss/rpn_3d.py:746-768, diff_nms_boxes_2d == "projected"): cuboid corners
// (get_corners_of_cuboid) -> image plane with the 4x4 projection p2 (lib/math_3d.py:47-72: rows 0,1 divided by row 2 where
// |row 2| > 1e-2) -> min/max over the 8 corners -> times the image's scale factor.  One thread per box.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void project_boxes3d_kernel(const float* __restrict__ params, const float* __restrict__ p2,
                                                              const float* __restrict__ scale, int N, long total, float4* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / N);
    const float* P = p2 + (size_t)b * 16;
    float cx[8], cy[8], cz[8];
    corners_of(params + i * 7, cx, cy, cz);
    float x1 = INFINITY, y1 = INFINITY, x2 = -INFINITY, y2 = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float u = ((P[0] * cx[k] + P[1] * cy[k]) + P[2] * cz[k]) + P[3];          // torch.matmul(p2, [x y z 1]^T), math_3d.py:66
        float v = ((P[4] * cx[k] + P[5] * cy[k]) + P[6] * cz[k]) + P[7];
        const float w = ((P[8] * cx[k] + P[9] * cy[k]) + P[10] * cz[k]) + P[11];
        if (fabsf(w) > 1e-2f) { u = u / w; v = v / w; }                           // :67, :69
        x1 = fminf(x1, u); x2 = fmaxf(x2, u); y1 = fminf(y1, v); y2 = fmaxf(y2, v);   // rpn_3d.py:762-765
    }
    const float sf = scale ? scale[b] : 1.0f;
    out[i] = make_float4(x1 * sf, y1 * sf, x2 * sf, y2 * sf);                     // :767
}

// ------------------------------------------------------------------------------------------------
// Best box per ground truth after the NMS (lib/loss/rpn_3d.py:801-825):
//     score[i][j] = 0.5 * (1 + GIoU3D(pred_i, gt_j)) * IoU2D(pred_i, gt_j)           (:819)
//     best[j] = argmax_i score[i][j]  (:820), kept only if score > beta (:822)  ->  targets[best[j]] = 1  (:826)
// One workgroup per (ground truth, image): the N x M score matrices of the reference are never materialised -- every thread
// evaluates its predictions against the one ground truth in registers (exact operation order of iou3d_kernel / iou2d_tile) and
// the workgroup reduces (score, index) to the FIRST maximum; a NaN score is a maximum, as in torch.max, and then fails `> beta`.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better_target(float s, int i, float bs, int bi) {   // (s,i) beats (bs,bi): larger, NaN largest, first index
    const bool sn = s != s, bn = bs != bs;
    if (bi < 0) return true;
    if (sn != bn) return sn;
    if (!sn && s != bs) return s > bs;
    return i < bi;
}

__global__ __launch_bounds__(256) void best_targets_kernel(const float* __restrict__ pred_params, const float* __restrict__ pred_boxes,
                                                           const float* __restrict__ gt_params, const float* __restrict__ gt_boxes, int N, int M,
                                                           const int* __restrict__ pred_counts, const int* __restrict__ gt_counts, float beta,
                                                           long long* __restrict__ best_index, float* __restrict__ best_score,
                                                           float* __restrict__ targets) {
    __shared__ float ws_s[4];
    __shared__ int ws_i[4];
    const int j = blockIdx.x, b = blockIdx.y;
    const int n = gnms_count(pred_counts, b, N), m = gnms_count(gt_counts, b, M);
    if (j >= m) {
        if (threadIdx.x == 0) { if (best_index) best_index[(size_t)b * M + j] = -1; if (best_score) best_score[(size_t)b * M + j] = 0.0f; }
        return;
    }
    float cx[8], cy[8], cz[8], g[kRec];
    corners_of(gt_params + ((size_t)b * M + j) * 7, cx, cy, cz);
    aabb_record(cx, cy, cz, g);
    const float4 gb = reinterpret_cast<const float4*>(gt_boxes)[(size_t)b * M + j];
    const float garea = (gb.z - gb.x) * (gb.w - gb.y);
    float bs = 0.0f;
    int bi = -1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float r[kRec];
        corners_of(pred_params + ((size_t)b * N + i) * 7, cx, cy, cz);
        aabb_record(cx, cy, cz, r);
        // prediction = box "a" (rows), ground truth = box "b" (columns): iou3d_kernel<METHOD 1> for one pair
        const float vol = r[0] + g[0];
        const float yi = relu0(fminf(r[2], g[2]) - fmaxf(r[1], g[1]));
        const float w = relu0(fminf(r[4], g[4]) - fmaxf(r[3], g[3]));
        const float h = relu0(fminf(r[6], g[6]) - fmaxf(r[5], g[5]));
        const float i3 = (w * h) * yi;
        const float u3 = vol - i3;
        float q = i3 / u3;
        const float xh = relu0(fmaxf(r[4], g[4]) - fminf(r[3], g[3]));
        const float yh = relu0(fmaxf(r[2], g[2]) - fminf(r[1], g[1]));
        const float zh = relu0(fmaxf(r[6], g[6]) - fminf(r[5], g[5]));
        const float vh = (xh * yh) * zh;
        q = q - ((vh - u3) / vh);
        // iou2d_tile for one pair
        const float4 pb = reinterpret_cast<const float4*>(pred_boxes)[(size_t)b * N + i];
        const float w2 = relu0(fminf(pb.z, gb.z) - fmaxf(pb.x, gb.x));
        const float h2 = relu0(fminf(pb.w, gb.w) - fmaxf(pb.y, gb.y));
        const float in2 = w2 * h2;
        const float iou2 = in2 / (((pb.z - pb.x) * (pb.w - pb.y) + garea) - in2);
        const float sc = (0.5f * (1.0f + q)) * iou2;                                   // rpn_3d.py:819
        if (better_target(sc, i, bs, bi)) { bs = sc; bi = i; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float os = __shfl_xor(bs, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (oi >= 0 && better_target(os, oi, bs, bi)) { bs = os; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { ws_s[wave] = bs; ws_i[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < 4; ++wv)
            if (ws_i[wv] >= 0 && better_target(ws_s[wv], ws_i[wv], bs, bi)) { bs = ws_s[wv]; bi = ws_i[wv]; }
        const bool keep = bi >= 0 && bs > beta;                                        // :822 (false for NaN)
        if (best_index) best_index[(size_t)b * M + j] = keep ? bi : -1;
        if (best_score) best_score[(size_t)b * M + j] = bi >= 0 ? bs : 0.0f;
        if (keep && targets) targets[(size_t)b * N + bi] = 1.0f;                       // :826 (several ground truths may pick the same box)
    }
}

// The per-box records are tiny (32 B/box); they live in a stream-ordered allocation so the public
// entry points stay allocation-free for the caller.  hipMallocAsync/hipFreeAsync are stream ordered.
static int iou3d_common(const float* in_a, const float* in_b, bool from_params, int B, int M, int N, int method, float* iou_bev,
                        float* iou_3d, int64_t ld, void* stream) {
    GNMS_CHECK_ARG(B >= 0 && M >= 0 && N >= 0, "gnms_iou3d: negative size");
    GNMS_CHECK_ARG(method >= 0 && method <= 2, "gnms_iou3d: method %d not in {0,1,2}", method);
    if (B == 0 || M == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(in_a && in_b && iou_3d, "gnms_iou3d: null pointer");
    GNMS_CHECK_ARG(ld >= N, "gnms_iou3d: ld < N");
    hipStream_t st = (hipStream_t)stream;
    float* rec = nullptr;
    const size_t na = (size_t)B * M, nb = (size_t)B * N;
    GNMS_CHECK_HIP(hipMallocAsync((void**)&rec, (na + nb) * kRec * sizeof(float), st));
    float* ra = rec;
    float* rb = rec + na * kRec;
    if (from_params) {
        aabb_from_params_kernel<<<(unsigned)((na + 255) / 256), 256, 0, st>>>(in_a, (long)na, ra);
        aabb_from_params_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, st>>>(in_b, (long)nb, rb);
    } else {
        aabb_from_corners_kernel<<<(unsigned)((na + 255) / 256), 256, 0, st>>>(in_a, (long)na, ra);
        aabb_from_corners_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, st>>>(in_b, (long)nb, rb);
    }
    // always the reference's operation order: the caller's later threshold is unknown here (gnms_nms_overlap3d_from_params takes it)
    int rc = iou3d_from_records(ra, rb, B, M, N, method, iou_bev, iou_3d, ld, st, false);
    hipError_t fe = hipFreeAsync(rec, st);
    if (rc != GNMS_OK) return rc;
    if (fe != hipSuccess) { gnms_set_error("hipFreeAsync failed: %s", hipGetErrorString(fe)); return GNMS_ERR_HIP; }
    return GNMS_OK;
}

extern "C" int gnms_iou3d_approximate(const float* corners_a, const float* corners_b, int B, int M, int N, int method,
                                      float* iou_bev, float* iou_3d, int64_t ld, void* stream) {
    return iou3d_common(corners_a, corners_b, false, B, M, N, method, iou_bev, iou_3d, ld, stream);
}

extern "C" int gnms_iou3d_from_params(const float* params_a, const float* params_b, int B, int M, int N, int method,
                                      float* iou_bev, float* iou_3d, int64_t ld, void* stream) {
    return iou3d_common(params_a, params_b, true, B, M, N, method, iou_bev, iou_3d, ld, stream);
}

bool gnms_internal_overlap3d_sym_ok(int N, int64_t ld, const float* out);
int gnms_internal_nms_overlap3d_sym(const float* rec, int B, int N, float* out, int64_t ld, hipStream_t st, float thr, int pct0, int pct1,
                                    int leave_cus = 0, int force_persist = 0);

extern "C" int gnms_nms_overlap3d_from_params(const float* params3d, int B, int N, float nms_threshold, float* out, int64_t ld, void* stream) {
    GNMS_CHECK_ARG(B >= 0 && N >= 0, "gnms_nms_overlap3d_from_params: negative size");
    if (B == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(params3d && out, "gnms_nms_overlap3d_from_params: null pointer");
    GNMS_CHECK_ARG(ld >= N, "gnms_nms_overlap3d_from_params: ld < N");
    hipStream_t st = (hipStream_t)stream;
    float* rec = nullptr;
    GNMS_CHECK_HIP(hipMallocAsync((void**)&rec, (size_t)B * N * kRec * sizeof(float), st));
    int rc = gnms_internal_records_from_params(params3d, (long)B * N, rec, st);
    if (!rc) {
        // one box set against itself: every unordered pair once (iou3d_sym.h), else the all-pairs kernel
        if (gnms_internal_overlap3d_sym_ok(N, ld, out)) rc = gnms_internal_nms_overlap3d_sym(rec, B, N, out, ld, st, nms_threshold, 0, 100);
        else rc = iou3d_from_records(rec, rec, B, N, N, 2, nullptr, out, ld, st, true, 0, 0x7fffffff, nms_threshold);
    }
    const hipError_t fe = hipFreeAsync(rec, st);
    if (rc != GNMS_OK) return rc;
    if (fe != hipSuccess) { gnms_set_error("hipFreeAsync failed: %s", hipGetErrorString(fe)); return GNMS_ERR_HIP; }
    return GNMS_OK;
}

extern "C" int gnms_project_boxes3d(const float* params, const float* p2, const float* scale, int B, int N, float* boxes2d, void* stream) {
    GNMS_CHECK_ARG(B >= 0 && N >= 0, "gnms_project_boxes3d: negative size");
    if (B == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(params && p2 && boxes2d, "gnms_project_boxes3d: null pointer");
    GNMS_CHECK_ARG((uintptr_t)boxes2d % 16 == 0, "gnms_project_boxes3d: boxes2d must be 16-byte aligned");
    const long total = (long)B * N;
    project_boxes3d_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(params, p2, scale, N, total,
                                                                                             reinterpret_cast<float4*>(boxes2d));
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

extern "C" int gnms_best_targets(const float* pred_params, const float* pred_boxes2d, const float* gt_params, const float* gt_boxes2d, int B,
                                 int N, int M, const int32_t* pred_counts, const int32_t* gt_counts, float beta, int64_t* best_index,
                                 float* best_score, float* targets, void* stream) {
    GNMS_CHECK_ARG(B >= 0 && N >= 0 && M >= 0, "gnms_best_targets: negative size");
    hipStream_t st = (hipStream_t)stream;
    if (targets && B > 0 && N > 0) GNMS_CHECK_HIP(hipMemsetAsync(targets, 0, sizeof(float) * (size_t)B * N, st));   // :800 "all other boxes 0"
    if (B == 0 || M == 0) return GNMS_OK;
    if (N == 0) {
        if (best_index) GNMS_CHECK_HIP(hipMemsetAsync(best_index, 0xff, sizeof(int64_t) * (size_t)B * M, st));
        if (best_score) GNMS_CHECK_HIP(hipMemsetAsync(best_score, 0, sizeof(float) * (size_t)B * M, st));
        return GNMS_OK;
    }
    GNMS_CHECK_ARG(pred_params && pred_boxes2d && gt_params && gt_boxes2d, "gnms_best_targets: null pointer");
    GNMS_CHECK_ARG(((uintptr_t)pred_boxes2d % 16 == 0) && ((uintptr_t)gt_boxes2d % 16 == 0), "gnms_best_targets: 2D boxes must be 16-byte aligned");
    best_targets_kernel<<<dim3(M, B), 256, 0, st>>>(pred_params, pred_boxes2d, gt_params, gt_boxes2d, N, M, pred_counts, gt_counts, beta,
                                                    (long long*)best_index, best_score, targets);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
