// gnms_common.h -- shared device/host helpers of libgroomed_nms_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/groomed_nms_hip.h"

#define GNMS_WAVE 64

// thread-local last-error message (gnms_last_error)
void gnms_set_error(const char* fmt, ...);

#define GNMS_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            gnms_set_error(__VA_ARGS__);          \
            return GNMS_ERR_INVALID_ARGUMENT;     \
        }                                         \
    } while (0)

#define GNMS_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            gnms_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return GNMS_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

// (GNMS_TRACE_LAUNCH=1, developer: every launch site prints its file:line and waits for the device -- a memory fault then aborts
// right behind the line of the launch that caused it)
static inline bool gnms_trace_launch() { static const bool on = [] { const char* e = getenv("GNMS_TRACE_LAUNCH"); return e && e[0] == '1'; }(); return on; }
#define GNMS_CHECK_LAUNCH()                                                                   \
    do {                                                                                      \
        if (gnms_trace_launch()) { fprintf(stderr, "[gnms launch] %s:%d\n", __FILE__, __LINE__); fflush(stderr); (void)hipDeviceSynchronize(); } \
        hipError_t e__ = hipGetLastError();                                                   \
        if (e__ != hipSuccess) {                                                              \
            gnms_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return GNMS_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

// allows `bytes` (> 64 KiB) of dynamic LDS for `kernel` on the current device, remembered per (device, kernel); nms_layer.hip
int gnms_allow_lds_raw(const void* kernel, size_t bytes);
// compute units of the current device, cached per device (nms_layer.hip)
int gnms_device_cu_count();

// A stream-ordered temporary (hipMallocAsync) that is returned to the pool on EVERY exit from the scope, error paths included:
// the GNMS_CHECK_* macros return early.  `release()` frees explicitly and reports the result.
struct gnms_async_buffer {
    void* p = nullptr;
    hipStream_t st = nullptr;
    gnms_async_buffer() = default;
    gnms_async_buffer(const gnms_async_buffer&) = delete;
    gnms_async_buffer& operator=(const gnms_async_buffer&) = delete;
    ~gnms_async_buffer() { if (p) (void)hipFreeAsync(p, st); }
    hipError_t alloc(size_t bytes, hipStream_t stream) { st = stream; return hipMallocAsync(&p, bytes, stream); }
    hipError_t release() { void* q = p; p = nullptr; return q ? hipFreeAsync(q, st) : hipSuccess; }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};


// one step of a host-side poll loop: a spin hint on x86 / ARM, nothing elsewhere
static inline void gnms_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    __asm__ __volatile__("yield" ::: "memory");
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}
// Bounded spin, then yield, then sleep (host_mailbox.hip, classic_nms.hip: the host polls a word of pinned memory the kernels store to).
// ~4 k pauses cover a call at the reference's size without a system call; the next ~4 k polls give the core away between looks; from there
// the poll sleeps 2, 4, ... 128 us.  wait() returns true every 1024 steps of the spin / yield phases and after every sleep: "look at the
// stream now" (a failed or empty stream never delivers the word).
struct gnms_poll_backoff {
    unsigned n = 0, sleep_us = 2;
    bool wait();
};

static inline int gnms_div_up(int a, int b) { return (a + b - 1) / b; }
static inline size_t gnms_align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------------
// workspace layout of the NMS layer (per call; all offsets 256-byte aligned)
//
//   per image b (N = padded box count of the batch, NB = ceil(N/64) rank blocks, NC = round_up(N,4)):
//     order      int32 [N]      rank -> input index (stable descending argsort of the scores)
//     sscore     float [N]      scores in rank order
//     rankof     int32 [N]      input index -> rank (inverse of order)
//     rem        int32 [N]      rank of the leader that removed rank k from `remaining` (k itself for leaders)
//     head       int32 [N]      rank of the first member of k's group (lib/groomed_nms.py:99 groups[j][0]),
//                               -1 when k is in no group
//     gpos       int32 [N]      position of k inside its group
//     gsorted    int32 [N]      ranks sorted by (leader, rank), members only: the groups as contiguous runs
//     gstart     int32 [N]      for rank k: index into gsorted where k's group starts (valid when head>=0)
//     glen       int32 [N]      for rank k: number of members of k's group kept under the cap
//     hlist      int32 [N]      ranks of the heads of groups with more than one member (misc[1] of them, any order); from the END, the heads of
//                               groups of more than 16 members once more (misc[4] of them)
//     plead      float [N]      masked mode: prune(iou[k][head]) after tril (0 for heads / non-members)
//     pre        float [N]      (M s)_k before the clamp (lib/groomed_nms.py:111); NMS order
//     r2         float [N]      clamp(pre, 0, 1)
//     sidx       int32 [N]      argsort (descending, stable) of the thresholded probabilities (:116-121)
//     xsol       float [N]      unmasked/ungrouped: workspace for the triangular solves
//     gx         float [N]      backward: dL/d(M s) after the clamp / threshold masks, by NMS position
//     leadc      int32 [N]      input index of the i-th leader (in rank order)
//     leadr      int32 [N]      rank of the i-th leader
//     leadw      u64   [NB]     bit k%64 of word k/64 set iff rank k is a leader
//     leadpfx    int32 [NB+1]   number of leaders in rank blocks < kb
//     misc       int32 [16]     [0]=number of leaders, [1]=length of hlist, ... [8]=the workspace's call counter (`epoch`: the sorts of every call
//                               add 1; it tags the hand-offs between the workgroups of one image's leader scan, leaders_sb_body)
//     gran       u64   [17][32] leader scan across workgroups: [s][2 bb + h] = (epoch << 32 | half h of the leader mask of block bb of super-block
//                               s), published by the workgroup that resolved s; [16][s] = (epoch << 32 | 1): its rem[] entries are stored
//     xidx       int32 [N]      from-boxes path: input index of the p-th box by ascending x centre
//     xbox       float4 [N]     from-boxes path: the boxes in that order
//     rbox       float4 [N]     from-boxes path: the boxes in RANK order (written by the score sort: the bit-matrix kernel's row boxes
//                               without the order -> box gather)
//     rec        float [N][12]  gnms_forward_with_iou3d: corner-AABB records of the cuboids (iou3d_pair.h)
//     xrec       float [N][12]  the same records in the COLUMN order of the 3D bit-matrix kernel (z band, then x centre; written by the sort)
//     W          u64   [NB][NC] W[kb][k'] bit r set iff !(iou[order[64*kb+r]][order[k']] <= thr): the ranks of block kb that
//                               rank k', were it a leader, takes out of `remaining` (:249-262).  Rank x rank space: the
//                               bit-matrix kernel reads input columns and scatters each word to its rank position.
//                               (one_launch_kernel, N <= 1024: the region holds the leader scan's triangular table instead, nms_one_launch.h)
// ------------------------------------------------------------------------------------------------
struct gnms_ws_layout {
    int N, NB, NC;
    size_t off_order, off_sscore, off_rankof, off_rem, off_head, off_gpos, off_gsorted, off_gstart, off_glen, off_hlist, off_plead, off_pre,
        off_r2, off_sidx, off_xsol, off_gx, off_leadc, off_leadr, off_leadw, off_leadpfx, off_misc, off_gran, off_xidx, off_xbox, off_rbox, off_rec, off_xrec, off_W;
    size_t per_image;  // bytes
};

static inline gnms_ws_layout gnms_make_layout(int N) {
    gnms_ws_layout L;
    L.N = N;
    L.NB = (N + 63) / 64;
    L.NC = (N + 3) / 4 * 4;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = gnms_align_up(o + bytes, 256); return r; };
    size_t n4 = (size_t)(N > 0 ? N : 1) * 4;
    L.off_order = take(n4); L.off_sscore = take(n4); L.off_rankof = take(n4); L.off_rem = take(n4); L.off_head = take(n4);
    L.off_gpos = take(n4); L.off_gsorted = take(n4); L.off_gstart = take(n4); L.off_glen = take(n4); L.off_hlist = take(n4);
    L.off_plead = take(n4); L.off_pre = take(n4); L.off_r2 = take(n4); L.off_sidx = take(n4);
    L.off_xsol = take(n4); L.off_gx = take(n4); L.off_leadc = take(n4); L.off_leadr = take(n4);
    L.off_leadw = take((size_t)(L.NB > 0 ? L.NB : 1) * 8);
    L.off_leadpfx = take((size_t)(L.NB + 1) * 4);
    L.off_misc = take(64);
    L.off_gran = take(17 * 32 * 8);
    L.off_xidx = take(n4);
    L.off_xbox = take(n4 * 4);
    L.off_rbox = take(n4 * 4);
    L.off_rec = take(n4 * 12);
    L.off_xrec = take(n4 * 12);
    {   // (up to one super-block the region also holds the image of the scan's triangular table, nms_one_launch.h: NB (NB + 1) / 2 pairs of 64 words)
        size_t wbytes = (size_t)(L.NB > 0 ? L.NB : 1) * (size_t)(L.NC > 0 ? L.NC : 4) * 8;
        const size_t tbytes = (size_t)L.NB * (L.NB + 1) / 2 * 64 * 8;
        if (L.NB <= 16 && wbytes < tbytes) wbytes = tbytes;
        L.off_W = take(wbytes);
    }
    L.per_image = o;
    return L;
}

#ifdef __HIPCC__
// the workspace's call counter (misc[8]) tags the hand-off granules of the leader scan; a granule the sorts have just zeroed carries tag 0, so
// the counter skips 0 -- a recycled workspace may hold ANY bytes there (0xffffffff: the -1 padding of a freed index list, found by the fuzz
// test in round 4b), and the sum must not wrap onto the tag of a cleared granule
__device__ __forceinline__ int gnms_next_epoch(int e) { const int n = e + 1; return n == 0 ? 1 : n; }

// number of boxes of image b: counts[b] clamped to [0, N] (a bad count must not turn into an out-of-bounds access)
__device__ __forceinline__ int gnms_count(const int* __restrict__ counts, int b, int N) {
    if (!counts) return N;
    const int c = counts[b];
    return c < 0 ? 0 : (c > N ? N : c);
}

// pruning_function (lib/groomed_nms.py:167-189) and its derivative, fp32, same expression order as the oracle
__device__ __forceinline__ float gnms_prune(float x, float thr, float temp, int method) {
    if (method == GNMS_PRUNE_LINEAR) return x;
    if (method == GNMS_PRUNE_SIGMOIDAL) {
        float z = (x - thr) / temp;
        return 1.0f / (1.0f + expf(-z));
    }
    return 1.0f - expf(-(x * x) / temp);
}
__device__ __forceinline__ float gnms_prune_grad(float x, float thr, float temp, int method) {
    if (method == GNMS_PRUNE_LINEAR) return 1.0f;
    if (method == GNMS_PRUNE_SIGMOIDAL) {
        float z = (x - thr) / temp;
        float sg = 1.0f / (1.0f + expf(-z));
        return sg * (1.0f - sg) / temp;
    }
    return expf(-(x * x) / temp) * (2.0f * x / temp);
}

// order-preserving map float -> uint32 for a DESCENDING sort done as an ascending sort of the key:
// larger floats get smaller keys; NaN is the greatest value (torch.sort semantics) -> key 0; -0.0 == +0.0.
__device__ __forceinline__ uint32_t gnms_desc_key(float v) {
    if (v != v) return 0u;                                     // every non-NaN key below is >= 0x007fffff
    uint32_t u = __float_as_uint(v + 0.0f);                    // -0.0 + 0.0 = +0.0
    uint32_t asc = (u >> 31) ? ~u : (u | 0x80000000u);         // ascending-sortable image of the float
    return ~asc;
}

// inverse of gnms_desc_key for non-NaN scores (-0.0 comes back as +0.0); key 0 decodes to NaN
__device__ __forceinline__ float gnms_desc_key_decode(uint32_t key) {
    if (key == 0u) return __uint_as_float(0x7fc00000u);
    const uint32_t asc = ~key;
    const uint32_t u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
    return __uint_as_float(u);
}

// wave-wide OR on the VALU: DPP row_shr 1,2,4,8 + row_bcast:15 + row_bcast:31 leave the total in lane 63 (an inclusive
// OR-scan on the way); 6 dependent DPP ops instead of 6 ds_bpermute round trips (checked on gfx950 with a stand-alone kernel in round 1)
__device__ __forceinline__ unsigned gnms_or_scan32(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);   // row_bcast:15 -> rows 1,3
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);   // row_bcast:31 -> rows 2,3
    return v;                                                                       // inclusive OR-scan over the 64 lanes
}
// the same network with additions: inclusive prefix sum over the 64 lanes (Hillis-Steele inside each row of 16, then the row carries)
__device__ __forceinline__ unsigned gnms_add_scan32(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);   // row_bcast:15 -> rows 1,3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);   // row_bcast:31 -> rows 2,3
    return v;
}
// inclusive running maximum of signed ints (identity INT_MIN comes in through bound_ctrl=0 -> use old = INT_MIN explicitly)
__device__ __forceinline__ int gnms_max_scan32(int v) {
    const int lo = (int)0x80000000;
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x111, 0xF, 0xF, false));   // row_shr:1 (lanes without a source keep `lo`)
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x118, 0xF, 0xF, false));   // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1,3
    v = max(v, __builtin_amdgcn_update_dpp(lo, v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2,3
    return v;
}
// wave-wide min / max of a float on the VALU (same DPP network; the result is wave-uniform, read from lane 63).  fminf/fmaxf
// semantics: a NaN operand yields the other operand.
__device__ __forceinline__ float gnms_wave_min_f(float v) {
    const int id = 0x7f800000;   // +inf for lanes without a source
#define GNMS_DPP_MIN(ctrl, rmask) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(id, __float_as_int(v), ctrl, rmask, 0xF, false)))
    GNMS_DPP_MIN(0x111, 0xF); GNMS_DPP_MIN(0x112, 0xF); GNMS_DPP_MIN(0x114, 0xF); GNMS_DPP_MIN(0x118, 0xF);
    GNMS_DPP_MIN(0x142, 0xA); GNMS_DPP_MIN(0x143, 0xC);
#undef GNMS_DPP_MIN
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float gnms_wave_max_f(float v) {
    const int id = (int)0xff800000;   // -inf
#define GNMS_DPP_MAX(ctrl, rmask) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(id, __float_as_int(v), ctrl, rmask, 0xF, false)))
    GNMS_DPP_MAX(0x111, 0xF); GNMS_DPP_MAX(0x112, 0xF); GNMS_DPP_MAX(0x114, 0xF); GNMS_DPP_MAX(0x118, 0xF);
    GNMS_DPP_MAX(0x142, 0xA); GNMS_DPP_MAX(0x143, 0xC);
#undef GNMS_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int gnms_wave_min_i(int v) {
    const int id = 0x7fffffff;
#define GNMS_DPP_MINI(ctrl, rmask) v = min(v, __builtin_amdgcn_update_dpp(id, v, ctrl, rmask, 0xF, false))
    GNMS_DPP_MINI(0x111, 0xF); GNMS_DPP_MINI(0x112, 0xF); GNMS_DPP_MINI(0x114, 0xF); GNMS_DPP_MINI(0x118, 0xF);
    GNMS_DPP_MINI(0x142, 0xA); GNMS_DPP_MINI(0x143, 0xC);
#undef GNMS_DPP_MINI
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned long long gnms_or_scan64(unsigned long long v) {
    const unsigned lo = gnms_or_scan32((unsigned)(v & 0xffffffffu));
    const unsigned hi = gnms_or_scan32((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// wave-wide OR of a 64-bit value; the result is wave-uniform (read from lane 63)
__device__ __forceinline__ unsigned long long gnms_wave_or(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)gnms_or_scan32((unsigned)(v & 0xffffffffu)), 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)gnms_or_scan32((unsigned)(v >> 32)), 63);
    return ((unsigned long long)hi << 32) | lo;
}
#endif
