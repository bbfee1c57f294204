// iou3d_pair.h -- the NMS overlap 0.5 * (1 + GIoU3D) of two corner-AABB records, for TWO columns at a time (packed fp32).
// ONE definition shared by iou3d_nms_fast_kernel (which writes the matrix) and bitmask_rec3d_kernel (which thresholds the same
// pairs without reading the matrix back): the two must produce the same bits, so they run the same instruction sequence.
// Reference: lib/core.py:305-421 (iou3d_approximate, method "generalized"), lib/loss/rpn_3d.py:781 (0.5 * (1 + giou)).
#pragma once
#include "gnms_common.h"

namespace gnms_iou3d {

constexpr int kRec = 12;   // floats per record: vol, y0, y1, x0, x1, z0, z1, area_bev, len x, len y, len z, 0

typedef float f2 __attribute__((ext_vector_type(2)));

// v_min_f32 / v_max_f32 issued directly, first operand wave-uniform (an SGPR): fminf/fmaxf on values the compiler cannot prove
// canonical (anything loaded from memory) cost an extra v_max_f32 x, x, x each.  IEEE mode: a NaN operand yields the other operand.
__device__ __forceinline__ float vmin_s(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax_s(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 min2(float a, f2 b) { return (f2){vmin_s(a, b.x), vmin_s(a, b.y)}; }
__device__ __forceinline__ f2 max2(float a, f2 b) { return (f2){vmax_s(a, b.x), vmax_s(a, b.y)}; }
__device__ __forceinline__ f2 splat(float v) { return (f2){v, v}; }
__device__ __forceinline__ f2 relu2(f2 a) { return __builtin_elementwise_max(a, (f2){0.0f, 0.0f}); }   // arithmetic results are canonical

struct Cols2 {             // two column boxes, field by field
    f2 x0, x1, y0, y1, z0, z1, vol, lx, ly, lz;
};
__device__ __forceinline__ void cols2_set(Cols2& c, int k, const float4 u, const float4 v, const float4 e) {   // record = u | v | e
    c.vol[k] = u.x; c.y0[k] = u.y; c.y1[k] = u.z; c.x0[k] = u.w; c.x1[k] = v.x; c.z0[k] = v.y; c.z1[k] = v.z;
    c.lx[k] = e.x; c.ly[k] = e.y; c.lz[k] = e.z;
}
struct Row {               // one row box, every field wave-uniform
    float x0, x1, y0, y1, z0, z1, vol, lx, ly, lz, bad;
};

// record[11]: 0 for a SANE box -- extents in (1e-4, 1e5), coordinates below 1e5 in magnitude, hence a positive finite volume -- and 1
// otherwise (NaN anywhere included).  For two sane boxes every intermediate of the overlap expressions is a normal finite number
// (u3 >= max volume > 1e-12, hull volume in (1e-12, 1e17), their product far inside the fp32 range), so the re-associated expression
// yields a finite value in [0, 1] within 1e-6 of the reference order.  Pairs with a box that is not sane always take the reference order.
__device__ __forceinline__ float record_bad_flag(float x0, float x1, float y0, float y1, float z0, float z1, float lx, float ly, float lz) {
    const bool ok = (lx > 1e-4f) && (lx < 1e5f) && (ly > 1e-4f) && (ly < 1e5f) && (lz > 1e-4f) && (lz < 1e5f) &&
                    (fabsf(x0) < 1e5f) && (fabsf(x1) < 1e5f) && (fabsf(y0) < 1e5f) && (fabsf(y1) < 1e5f) && (fabsf(z0) < 1e5f) && (fabsf(z1) < 1e5f);
    return ok ? 0.0f : 1.0f;
}

// 0.5 * (1 + i3/u3 - (vh - u3)/vh) re-associated to 0.5 * (i3*vh + u3*u3) / (u3*vh): one v_rcp_f32, hull extents from the
// overlap's own d, everything two columns wide.  Symmetric in (row, column) bit for bit.
__device__ __forceinline__ f2 nms_overlap3d(const Row& a, const Cols2& b) {
    const f2 dx = min2(a.x1, b.x1) - max2(a.x0, b.x0);
    const f2 dy = min2(a.y1, b.y1) - max2(a.y0, b.y0);
    const f2 dz = min2(a.z1, b.z1) - max2(a.z0, b.z0);
    const f2 i3 = (relu2(dx) * relu2(dz)) * relu2(dy);                       // lib/core.py:410-415
    const f2 u3 = (splat(a.vol) + b.vol) - i3;                               // :357, :416
    const f2 hx = (splat(a.lx) + b.lx) - dx;                                 // :390-406 hull extents
    const f2 hy = (splat(a.ly) + b.ly) - dy;
    const f2 hz = (splat(a.lz) + b.lz) - dz;
    const f2 vh = (hx * hy) * hz;
    const f2 num = __builtin_elementwise_fma(u3, u3, i3 * vh);
    const f2 den = u3 * vh;
    const f2 rc = (f2){__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    return (num * rc) * (f2){0.5f, 0.5f};
}

// The reference's own operation order for the same two columns (lib/core.py:357-419 + lib/loss/rpn_3d.py:781: i3/u3, then
// - (vh - u3)/vh, then 0.5 * (1 + .)), two IEEE divisions per pair: what iou3d_kernel<METHOD 2> and the CPU oracle compute, bit
// for bit.  Only the guard band below pays for it.
__device__ __forceinline__ f2 nms_overlap3d_exact(const Row& a, const Cols2& b) {
    const f2 vol = splat(a.vol) + b.vol;                                     // :357
    const f2 yi = relu2(min2(a.y1, b.y1) - max2(a.y0, b.y0));                // :371-376
    const f2 w = relu2(min2(a.x1, b.x1) - max2(a.x0, b.x0));                 // intersect(bev) :410
    const f2 h = relu2(min2(a.z1, b.z1) - max2(a.z0, b.z0));
    const f2 i3 = (w * h) * yi;                                              // :415
    const f2 u3 = vol - i3;                                                  // :416
    const f2 xh = relu2(max2(a.x1, b.x1) - min2(a.x0, b.x0));                // :390-406
    const f2 yh = relu2(max2(a.y1, b.y1) - min2(a.y0, b.y0));
    const f2 zh = relu2(max2(a.z1, b.z1) - min2(a.z0, b.z0));
    const f2 vh = (xh * yh) * zh;
    const f2 e = vh - u3;
    f2 q;
    q.x = i3.x / u3.x - e.x / vh.x;                                          // :417-419
    q.y = i3.y / u3.y - e.y / vh.y;
    return (f2){0.5f, 0.5f} * ((f2){1.0f, 1.0f} + q);                        // rpn_3d.py:781
}

// GUARD BAND.  The layer decides `overlap > nms_threshold` (lib/groomed_nms.py:249-250) on the matrix entries, and the re-associated
// expression above differs from the reference's order by a few fp32 roundings of a value in [0, 1] (measured <= 3e-7, bound 2e-6).
// The value of a pair is therefore DEFINED as
//     both boxes sane and |fast - thr| > kGuard3D = 8e-6 :  the re-associated expression
//     otherwise                                          :  the reference's exact operation order
// -- outside the band both expressions lie on the same side of the threshold, inside it the entry IS the reference's fp32 value, so
// thresholding takes exactly the reference's decision for every pair.  The definition is per pair (it does not depend on which other
// boxes share a wave tile), and the matrix kernel, the bit-matrix kernels and the single-pair lookups all evaluate it through the
// functions below, hence agree bit for bit.  A few dozen pairs per image fall into the band (N = 4096 .. 16384): the exact order runs
// behind ONE wave-uniform branch per row; the test itself is a packed subtract and a min3 per four pairs.
constexpr float kGuard3D = 8e-6f;

// four columns of one row.  colbad: bit j set iff the lane's column j is not sane (record[11] != 0); cols_sane: no lane of the wave has
// such a column (wave-uniform, computed once per tile).
__device__ __forceinline__ void nms_overlap3d_guarded4(const Row& a, const Cols2 (&b)[2], unsigned colbad, bool cols_sane, float thr, float (&q)[4]) {
    const f2 q0 = nms_overlap3d(a, b[0]), q1 = nms_overlap3d(a, b[1]);
    const f2 d0 = q0 - splat(thr), d1 = q1 - splat(thr);
    const float m = fminf(fminf(fminf(fabsf(d0.x), fabsf(d0.y)), fabsf(d1.x)), fabsf(d1.y));
    q[0] = q0.x; q[1] = q0.y; q[2] = q1.x; q[3] = q1.y;
    if (__any(!(m > kGuard3D)) || !cols_sane || a.bad != 0.0f) {              // rare
        const f2 e0 = nms_overlap3d_exact(a, b[0]), e1 = nms_overlap3d_exact(a, b[1]);
        const bool rb = a.bad != 0.0f;
        if (rb || (colbad & 1u) || !(fabsf(d0.x) > kGuard3D)) q[0] = e0.x;
        if (rb || (colbad & 2u) || !(fabsf(d0.y) > kGuard3D)) q[1] = e0.y;
        if (rb || (colbad & 4u) || !(fabsf(d1.x) > kGuard3D)) q[2] = e1.x;
        if (rb || (colbad & 8u) || !(fabsf(d1.y) > kGuard3D)) q[3] = e1.y;
    }
}

// ONE column of one row (the slot-culled bit-matrix kernel evaluates a row against the 64-column slots that survive its cull, one slot
// at a time).  Same operations in the same order as nms_overlap3d, one column wide: the packed instructions round each half exactly like
// their scalar counterparts (-ffp-contract=off), so the value equals the packed evaluation bit for bit.
struct Col1 {
    float x0, x1, y0, y1, z0, z1, vol, lx, ly, lz;
};
__device__ __forceinline__ void col1_set(Col1& c, const float4 u, const float4 v, const float4 e) {                  // record = u | v | e
    c.vol = u.x; c.y0 = u.y; c.y1 = u.z; c.x0 = u.w; c.x1 = v.x; c.z0 = v.y; c.z1 = v.z; c.lx = e.x; c.ly = e.y; c.lz = e.z;
}
__device__ __forceinline__ float relu1(float a) { return __builtin_fmaxf(a, 0.0f); }                                 // arithmetic results are canonical
__device__ __forceinline__ float nms_overlap3d_1(const Row& a, const Col1& b) {
    const float dx = vmin_s(a.x1, b.x1) - vmax_s(a.x0, b.x0);
    const float dy = vmin_s(a.y1, b.y1) - vmax_s(a.y0, b.y0);
    const float dz = vmin_s(a.z1, b.z1) - vmax_s(a.z0, b.z0);
    const float i3 = (relu1(dx) * relu1(dz)) * relu1(dy);
    const float u3 = (a.vol + b.vol) - i3;
    const float hx = (a.lx + b.lx) - dx;
    const float hy = (a.ly + b.ly) - dy;
    const float hz = (a.lz + b.lz) - dz;
    const float vh = (hx * hy) * hz;
    const float num = __builtin_fmaf(u3, u3, i3 * vh);
    const float den = u3 * vh;
    return (num * __builtin_amdgcn_rcpf(den)) * 0.5f;
}
// guarded value of one column (the definition of nms_overlap3d_guarded4, one column wide).  colbad: this lane's column is not sane;
// cols_sane: no lane of the wave has such a column in this slot.
__device__ __forceinline__ float nms_overlap3d_guarded1(const Row& a, const Col1& b, bool colbad, bool cols_sane, float thr) {
    float q = nms_overlap3d_1(a, b);
    const float d = fabsf(q - thr);
    if (__any(!(d > kGuard3D)) || !cols_sane || a.bad != 0.0f) {              // rare
        Cols2 c;
        c.x0 = splat(b.x0); c.x1 = splat(b.x1); c.y0 = splat(b.y0); c.y1 = splat(b.y1); c.z0 = splat(b.z0); c.z1 = splat(b.z1);
        c.vol = splat(b.vol); c.lx = splat(b.lx); c.ly = splat(b.ly); c.lz = splat(b.lz);
        const f2 e = nms_overlap3d_exact(a, c);
        if (a.bad != 0.0f || colbad || !(d > kGuard3D)) q = e.x;
    }
    return q;
}

// The same value for ONE pair of records with per-lane operands (the layer's O(N) single-entry lookups when it runs beside the
// matrix write instead of after it).  Same operations in the same order as nms_overlap3d / nms_overlap3d_exact, one column wide:
// with -ffp-contract=off every product, sum, division and the v_rcp_f32 round exactly as there, and v_min/v_max do not depend on
// which operand sits in an SGPR, so the result equals the matrix entry bit for bit (tests/test_gpu_parity.py compares the paths).
__device__ __forceinline__ float nms_overlap3d_pair(const float* __restrict__ ra, const float* __restrict__ rb, float thr) {
    const float4 au = reinterpret_cast<const float4*>(ra)[0], av = reinterpret_cast<const float4*>(ra)[1], ae = reinterpret_cast<const float4*>(ra)[2];
    const float4 bu = reinterpret_cast<const float4*>(rb)[0], bv = reinterpret_cast<const float4*>(rb)[1], be = reinterpret_cast<const float4*>(rb)[2];
    const float dx = fminf(av.x, bv.x) - fmaxf(au.w, bu.w);
    const float dy = fminf(au.z, bu.z) - fmaxf(au.y, bu.y);
    const float dz = fminf(av.z, bv.z) - fmaxf(av.y, bv.y);
    const float i3 = (fmaxf(dx, 0.0f) * fmaxf(dz, 0.0f)) * fmaxf(dy, 0.0f);
    const float u3 = (au.x + bu.x) - i3;
    const float hx = (ae.x + be.x) - dx;
    const float hy = (ae.y + be.y) - dy;
    const float hz = (ae.z + be.z) - dz;
    const float vh = (hx * hy) * hz;
    const float num = __builtin_fmaf(u3, u3, i3 * vh);
    const float den = u3 * vh;
    float q = (num * __builtin_amdgcn_rcpf(den)) * 0.5f;
    if (ae.w != 0.0f || be.w != 0.0f || !(fabsf(q - thr) > kGuard3D)) {      // exact order (nms_overlap3d_exact, one column wide)
        const float xh = fmaxf(fmaxf(av.x, bv.x) - fminf(au.w, bu.w), 0.0f);
        const float yh = fmaxf(fmaxf(au.z, bu.z) - fminf(au.y, bu.y), 0.0f);
        const float zh = fmaxf(fmaxf(av.z, bv.z) - fminf(av.y, bv.y), 0.0f);
        const float vhe = (xh * yh) * zh;
        const float g = i3 / u3 - (vhe - u3) / vhe;
        q = 0.5f * (1.0f + g);
    }
    return q;
}

}  // namespace gnms_iou3d
