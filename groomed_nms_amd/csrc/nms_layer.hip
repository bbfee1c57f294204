// nms_layer.hip -- host side of the GrooMeD-NMS C ABI: argument checks, workspace carving, launches.
// Kernels: nms_kernels.h (forward), nms_backward_kernels.h, nms_solve_kernels.h.
#include <stdarg.h>
#include <string.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <atomic>
#include <vector>
#include "gnms_prof.h"
#include "iou_tile.h"
#include "iou3d_tile.h"
#include "iou3d_sym.h"
#include "nms_solve_kernels.h"
#include "nms_one_launch.h"

// defined in iou_kernels.hip
bool gnms_internal_overlap3d_sym_ok(int N, int64_t ld, const float* out);
int gnms_internal_nms_overlap3d_sym(const float* rec, int B, int N, float* out, int64_t ld, hipStream_t st, float thr, int pct0, int pct1,
                                    int leave_cus = 0, int force_persist = 0);
int gnms_internal_iou2d_rows(const float* boxes, int B, int N, float* out, int64_t ld, hipStream_t st, int row0, int row_end);

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void gnms_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gnms_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// profiling: event pairs around the HBM-bound launches (bench.py's roofline is computed from these)
// ------------------------------------------------------------------------------------------------
namespace {
struct ProfPair { int dev; hipEvent_t start, stop; };
struct ProfState {
    std::atomic<bool> armed{false};
    std::mutex mu;
    std::map<int, std::vector<hipEvent_t>> pool;                   // recycled events, per device (an event belongs to the device it was created on)
    std::vector<ProfPair> pairs[kProfSlots];
    hipEvent_t take(int dev) {
        std::vector<hipEvent_t>& p = pool[dev];
        if (!p.empty()) { hipEvent_t e = p.back(); p.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};
ProfState& prof() { static ProfState P; return P; }
}  // namespace

bool gnms_prof_armed() { return prof().armed.load(std::memory_order_relaxed); }
bool gnms_prof_pair(int slot, hipEvent_t* start, hipEvent_t* stop) {
    ProfState& P = prof();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(P.mu);
    hipEvent_t a = P.take(dev), b = P.take(dev);
    if (!a || !b) { if (a) P.pool[dev].push_back(a); if (b) P.pool[dev].push_back(b); return false; }
    P.pairs[slot].push_back(ProfPair{dev, a, b});
    *start = a;
    *stop = b;
    return true;
}

extern "C" int gnms_profile_events(int enable) {
    prof().armed.store(enable != 0, std::memory_order_relaxed);
    return GNMS_OK;
}

extern "C" int gnms_profile_collect(int slot, double* ms_sum, int* launches) {
    GNMS_CHECK_ARG(slot >= 0 && slot < kProfSlots && ms_sum && launches, "gnms_profile_collect: bad argument");
    ProfState& P = prof();
    std::lock_guard<std::mutex> lock(P.mu);
    double sum = 0.0;
    int n = 0;
    hipError_t first_err = hipSuccess;
    // every pair leaves the list and goes back to its device's pool whatever happens: an error on one pair is reported after the
    // walk, it never strands the others (or leaves recycled events behind in the list)
    for (const ProfPair& pr : P.pairs[slot]) {
        float ms = 0.0f;
        hipError_t e = hipEventSynchronize(pr.stop);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, pr.start, pr.stop);
        if (e == hipSuccess) { sum += ms; ++n; }
        else if (first_err == hipSuccess) first_err = e;
        P.pool[pr.dev].push_back(pr.start);
        P.pool[pr.dev].push_back(pr.stop);
    }
    P.pairs[slot].clear();
    *ms_sum = sum;
    *launches = n;
    if (first_err != hipSuccess) {
        gnms_set_error("gnms_profile_collect: %s", hipGetErrorString(first_err));
        return GNMS_ERR_HIP;
    }
    return GNMS_OK;
}

namespace {
// plain streams: what the HBM delivers to a kernel that does nothing else (the achievable ceiling the roofline is quoted beside)
__global__ __launch_bounds__(512) void prof_fill_kernel(float4* __restrict__ dst, size_t n4, float v) {
    const float4 x = make_float4(v, v, v, v);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float* p = reinterpret_cast<float*>(dst + i);
        __builtin_nontemporal_store(x.x, p); __builtin_nontemporal_store(x.y, p + 1);
        __builtin_nontemporal_store(x.z, p + 2); __builtin_nontemporal_store(x.w, p + 3);
    }
}
__global__ __launch_bounds__(512) void prof_read_kernel(const float4* __restrict__ src, size_t n4, float* __restrict__ sink) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float* p = reinterpret_cast<const float*>(src + i);
        acc += __builtin_nontemporal_load(p) + __builtin_nontemporal_load(p + 1) + __builtin_nontemporal_load(p + 2) + __builtin_nontemporal_load(p + 3);
    }
    if (acc == 123456.789f) *sink = acc;                          // never true for the data it is run on; keeps the loads alive
}
int device_cu_count() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    int& c = cus[dev];
    if (c == 0) {
        hipDeviceProp_t prop;
        c = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return c;
}


// the same bytes in the geometry of the matrix writers: a wave stores ROWS rows x 1 KiB (rows `ld` floats apart), 16 waves side by side,
// one workgroup per CU walking the row bands -- no loads, no arithmetic
template <int ROWS, bool NT>
__global__ __launch_bounds__(1024) void prof_fill_tiles_kernel(float* __restrict__ dst, int N, long ld, long bands, float v, int order) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncc = (N + 255) >> 8;
    const int ncg = (ncc + 15) >> 4;                              // column groups of 16 wave tiles (4096 columns)
    const long units = bands * ncg;
    const long bands_img = N / ROWS;
    const long per = (units + gridDim.x - 1) / gridDim.x;
    for (long u = order == 2 ? blockIdx.x * per : blockIdx.x; u < (order == 2 ? min(units, (blockIdx.x + 1) * per) : units); u += order == 2 ? 1 : gridDim.x) {
        long band;
        int g;
        if (order == 0) { band = u / ncg; g = (int)(u - band * ncg); }           // row band major: a band's column groups side by side
        else {                                                                    // (image, column group) major: bands of one group in a row
            const long pair = u / bands_img, bi = u - pair * bands_img;
            const long img = pair / ncg;
            g = (int)(pair - img * ncg);
            band = img * bands_img + bi;
        }
        const int col = (g * 16 + wave) * 256 + 4 * lane;
        if (col + 3 >= N) continue;
        float* p = dst + band * ROWS * ld + col;
#pragma unroll 16
        for (int r = 0; r < ROWS; ++r) {
            if (NT) {
                __builtin_nontemporal_store(v, p); __builtin_nontemporal_store(v, p + 1);
                __builtin_nontemporal_store(v, p + 2); __builtin_nontemporal_store(v, p + 3);
            } else {
                *reinterpret_cast<float4*>(p) = make_float4(v, v, v, v);
            }
            p += ld;
        }
    }
}
}  // namespace
int gnms_device_cu_count() { return device_cu_count(); }
extern "C" int gnms_profile_fill_tiles(float* dst, int B, int N, int64_t ld, int rows, int nontemporal, void* stream) {
    GNMS_CHECK_ARG(dst && B > 0 && N > 0 && ld >= N && ld % 4 == 0 && (uintptr_t)dst % 16 == 0,
                   "gnms_profile_fill_tiles: dst 16-byte aligned, ld >= N a multiple of 4");
    GNMS_CHECK_ARG((rows == 4 || rows == 8 || rows == 16 || rows == 32 || rows == 64) && N % rows == 0,
                   "gnms_profile_fill_tiles: rows per wave tile must be 4, 8, 16, 32 or 64 and divide N (rows=%d N=%d)", rows, N);
    const int order = 0;                                           // row band major (the other orders of round 2 measured slower: LABNOTES.md)
    const dim3 grid((unsigned)device_cu_count());
    hipStream_t st = (hipStream_t)stream;
    const long bands = (long)B * N / rows;
#define GNMS_FILL_TILES(R)                                                                                                                  \
    do {                                                                                                                                    \
        if (nontemporal) gnms_launch_prof(kProfPlainStream, prof_fill_tiles_kernel<R, true>, grid, dim3(1024), 0, st, dst, N, (long)ld, bands, 0.5f, order);   \
        else gnms_launch_prof(kProfPlainStream, prof_fill_tiles_kernel<R, false>, grid, dim3(1024), 0, st, dst, N, (long)ld, bands, 0.5f, order);            \
    } while (0)
    switch (rows) {
        case 4: GNMS_FILL_TILES(4); break;
        case 8: GNMS_FILL_TILES(8); break;
        case 16: GNMS_FILL_TILES(16); break;
        case 32: GNMS_FILL_TILES(32); break;
        default: GNMS_FILL_TILES(64); break;
    }
#undef GNMS_FILL_TILES
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
namespace {
// The store pattern of a SYMMETRIC matrix writer, with no arithmetic: upper-triangular T x T macro tiles, each written once where it
// is and once mirrored (rows <-> columns), row runs of T floats.  CPL floats per lane and store instruction: T / CPL lanes cover a row
// run, 64 / (T / CPL) row runs per instruction.  persist: one workgroup per CU walks a contiguous range of tiles (row-major over the
// upper triangle: a strip I, J = I .. end), else one workgroup per tile.
template <int T, int CPL, bool NT>
__global__ __launch_bounds__(1024) void prof_fill_sym_kernel(float* __restrict__ dst, int N, long ld, int nimg, int persist, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    constexpr int LPR = T / CPL, RPI = 64 / LPR;
    const int RW = T / waves;
    const int nt = N / T;
    const long per_img = (long)nt * (nt + 1) / 2, total = per_img * nimg;
    long t0 = blockIdx.x, t1 = blockIdx.x + 1;
    if (persist) { t0 = total * blockIdx.x / gridDim.x; t1 = total * (blockIdx.x + 1) / gridDim.x; }
    const long step = persist == 2 ? gridDim.x : 1;                  // persist 2: round robin (one compact write frontier), else contiguous strips
    if (persist == 2) { t0 = blockIdx.x; t1 = total; }
    for (long t = t0; t < t1 && t < total; t += step) {
        const int img = (int)(t / per_img);
        long r = t - (long)img * per_img;
        int I = 0;
        while (r >= nt - I) { r -= nt - I; ++I; }                      // (a benchmark: the linear walk is a few dozen trips)
        const int J = I + (int)r;
        float* base = dst + (size_t)img * N * ld;
        auto put = [&](int R0, int C0) {
            for (int rr = 0; rr < RW; rr += RPI) {
                float* p = base + (size_t)(R0 + wave * RW + rr + lane / LPR) * ld + C0 + (lane % LPR) * CPL;
                if (NT) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) __builtin_nontemporal_store(v, p + j);
                } else if (CPL == 4) {
                    *reinterpret_cast<float4*>(p) = make_float4(v, v, v, v);
                } else {
                    *reinterpret_cast<float2*>(p) = make_float2(v, v);
                }
            }
        };
        put(I * T, J * T);
        if (I != J) put(J * T, I * T);
    }
}
}  // namespace
extern "C" int gnms_profile_fill_sym(float* dst, int B, int N, int64_t ld, int tile, int cols_per_lane, int nontemporal, int persist, void* stream) {
    GNMS_CHECK_ARG(dst && B > 0 && N > 0 && ld >= N && ld % 4 == 0 && (uintptr_t)dst % 16 == 0, "gnms_profile_fill_sym: dst 16-byte aligned, ld >= N a multiple of 4");
    GNMS_CHECK_ARG((tile == 128 || tile == 256) && N % tile == 0 && (cols_per_lane == 2 || cols_per_lane == 4) && tile / cols_per_lane <= 64,
                   "gnms_profile_fill_sym: tile 128 or 256 dividing N, 2 or 4 columns per lane");
    hipStream_t st = (hipStream_t)stream;
    const int nt = N / tile;
    const long total = (long)nt * (nt + 1) / 2 * B;
    const dim3 grid((unsigned)(persist ? device_cu_count() * (tile == 128 ? 2 : 1) : total));
    const dim3 block(tile == 128 ? 512 : 1024);
#define GNMS_FILL_SYM(TT, CC)                                                                                                              \
    do {                                                                                                                                   \
        if (nontemporal) gnms_launch_prof(kProfPlainStream, prof_fill_sym_kernel<TT, CC, true>, grid, block, 0, st, dst, N, (long)ld, B, persist, 0.5f);   \
        else gnms_launch_prof(kProfPlainStream, prof_fill_sym_kernel<TT, CC, false>, grid, block, 0, st, dst, N, (long)ld, B, persist, 0.5f);            \
    } while (0)
    if (tile == 128 && cols_per_lane == 2) GNMS_FILL_SYM(128, 2);
    else if (tile == 128) GNMS_FILL_SYM(128, 4);
    else GNMS_FILL_SYM(256, 4);
#undef GNMS_FILL_SYM
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
extern "C" int gnms_profile_fill(float* dst, size_t count, void* stream) {
    GNMS_CHECK_ARG(dst && count % 4 == 0 && (uintptr_t)dst % 16 == 0, "gnms_profile_fill: dst must be 16-byte aligned, count a multiple of 4");
    if (count == 0) return GNMS_OK;
    gnms_launch_prof(kProfPlainStream, prof_fill_kernel, dim3(256 * 16), dim3(512), 0, (hipStream_t)stream, reinterpret_cast<float4*>(dst), count / 4, 0.5f);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
extern "C" int gnms_profile_read(const float* src, size_t count, float* sink, void* stream) {
    GNMS_CHECK_ARG(src && sink && count % 4 == 0 && (uintptr_t)src % 16 == 0, "gnms_profile_read: src must be 16-byte aligned, count a multiple of 4");
    if (count == 0) return GNMS_OK;
    gnms_launch_prof(kProfPlainStream, prof_read_kernel, dim3(256 * 16), dim3(512), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(src), count / 4, sink);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
extern "C" int gnms_abi_version(void) { return GNMS_ABI_VERSION; }

extern "C" void gnms_default_params(gnms_params* p) {
    if (!p) return;
    p->nms_threshold = 0.4f;             // lib/groomed_nms.py:10 defaults
    p->temperature = 0.01f;
    p->valid_box_prob_threshold = 0.3f;
    p->pruning_method = GNMS_PRUNE_LINEAR;
    p->return_sorted_prob = 0;
    p->group_boxes = 1;
    p->mask_group_boxes = 1;
    p->group_size = 100;
    p->presorted = 0;
}

// Kernels that need more than 64 KiB of dynamic LDS must be told so once per (device, kernel); the attribute call is not free
// (a driver round trip per launch adds up on the small-N path), so what has been granted is remembered.  (Declared in gnms_common.h.)
int gnms_allow_lds_raw(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return GNMS_OK;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    GNMS_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& g = granted[std::make_pair(dev, kernel)];
    if (g >= bytes) return GNMS_OK;
    GNMS_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    g = bytes;
    return GNMS_OK;
}

namespace {

using namespace gnms;

int next_pow2(int n) {
    int p = 64;               // block_sort works on >= 64 keys (one wave)
    while (p < n) p <<= 1;
    return p;
}

// block_sort<E>: P = threads * E.  P <= 1024 -> E = 1; above, 1024 threads and E = P / 1024 (N <= 16384 -> E <= 16).
#define GNMS_DISPATCH_SORT(P2, ...)                               \
    do {                                                          \
        const int e__ = (P2) <= 1024 ? 1 : (P2) / 1024;           \
        if (e__ == 1) { constexpr int E = 1; __VA_ARGS__; }              \
        else if (e__ == 2) { constexpr int E = 2; __VA_ARGS__; }         \
        else if (e__ == 4) { constexpr int E = 4; __VA_ARGS__; }         \
        else if (e__ == 8) { constexpr int E = 8; __VA_ARGS__; }         \
        else { constexpr int E = 16; __VA_ARGS__; }                      \
    } while (0)

size_t leaders_lds_bytes(int N) { return leaders_lds_size((N + 63) / 64); }

int allow_lds_raw(const void* kernel, size_t bytes) { return gnms_allow_lds_raw(kernel, bytes); }
template <typename K>
int allow_lds(K kernel, size_t bytes) { return allow_lds_raw(reinterpret_cast<const void*>(kernel), bytes); }

int check_common(const char* fn, int B, int N, int64_t ld, const gnms_params* P, const void* ws, size_t ws_bytes) {
    GNMS_CHECK_ARG(P != nullptr, "%s: params is NULL", fn);
    GNMS_CHECK_ARG(B >= 0 && N >= 0, "%s: negative size (B=%d N=%d)", fn, B, N);
    GNMS_CHECK_ARG(ld >= N, "%s: ld (%lld) < N (%d)", fn, (long long)ld, N);
    if (N > GNMS_MAX_BOXES) {
        gnms_set_error("%s: N=%d exceeds GNMS_MAX_BOXES=%d", fn, N, GNMS_MAX_BOXES);
        return GNMS_ERR_UNSUPPORTED;
    }
    if (P->pruning_method < 0 || P->pruning_method > 2) {
        gnms_set_error("%s: Pruning method not implemented! (pruning_method=%d)", fn, P->pruning_method);   // lib/groomed_nms.py:178
        return GNMS_ERR_UNSUPPORTED;
    }
    GNMS_CHECK_ARG(P->group_size >= 0, "%s: group_size < 0", fn);
    if (P->group_boxes && !P->mask_group_boxes && (long long)P->group_size + 1 > kGroupMaxMembers && N > kGroupMaxMembers) {
        gnms_set_error("%s: unmasked groups support group_size+1 <= %d", fn, kGroupMaxMembers);
        return GNMS_ERR_UNSUPPORTED;
    }
    if (B > 0 && N > 0) {
        GNMS_CHECK_ARG(ws != nullptr, "%s: workspace is NULL", fn);
        if (ws_bytes < gnms_workspace_bytes(B, N, P)) {
            gnms_set_error("%s: workspace too small (%zu < %zu)", fn, ws_bytes, gnms_workspace_bytes(B, N, P));
            return GNMS_ERR_WORKSPACE;
        }
        GNMS_CHECK_ARG((uintptr_t)ws % 256 == 0, "%s: workspace must be 256-byte aligned", fn);
    }
    return GNMS_OK;
}

// K2, the one full read of the matrix: workgroups of 8 waves (2048 columns side by side), 16 from N = 4096 (whole 16-KiB rows), eight
// one-KiB loads in flight per wave.
// The matrix-in layer decides on the device whether the thresholded matrix is symmetric (every caller in the reference passes
// iou(boxes, boxes)): bitmask_kernel then stores the rows of W in full and wsym_check_kernel compares the 64 x 64 blocks with
// their transposes; K3 / K4 (sym = 2) take the pulling, attributing scan for the images that pass.  GNMS_MATRIX_SYM=0: never.
bool matrix_sym_detection(int N) {
    static const int forced = [] { const char* e = getenv("GNMS_MATRIX_SYM"); return e ? atoi(e) : -1; }();
    return forced >= 0 ? forced != 0 : N >= 256;
}
// full: 0 = only the words a leader scan reads; 1 = whole rows + wsym_check_kernel behind; 2 = whole rows, the check rides in the tail launch
int launch_bitmask(const float* iou, int B, int N, int64_t ld, const int32_t* counts, float thr, char* ws, const gnms_ws_layout& L, hipStream_t st, int full = 0) {
    // (measured at B = 8, N = 4096, three interleaved repetitions: 16 waves 91.6-91.8 us = 0.733 of the HBM peak, 8 waves 95.2-95.7 us;
    // 16 loads in flight per wave change nothing either way)
    const bool vec = (ld % 4 == 0) && ((uintptr_t)iou % 16 == 0);
#define GNMS_BITMASK(V, W, R)                                                                                                         \
    gnms_launch_prof(kProfMatrixRead, bitmask_kernel<V, W, R>, dim3(gnms_div_up(N, W * 256), L.NB, B), dim3(W * 64), 0, st, iou, N, (long)ld, counts, thr, ws, L, full)
    // few, small images: one batch of loads per wave instead of eight in a row (bitmask_small_kernel)
    constexpr int small_wgs = 1024;
    if (!vec) GNMS_BITMASK(false, kMaskWaves, kMaskRB);
    else if (N <= 2048 && (long)B * L.NB * gnms_div_up(N, 256) <= (long)small_wgs)        // (N = 4096, B = 1 keeps the row-buffered 16-wave kernel)
        gnms_launch_prof(kProfMatrixRead, bitmask_small_kernel, dim3(gnms_div_up(N, 256), L.NB, B), dim3(512), 0, st, iou, N, (long)ld, counts, thr, ws, L, full);
    else if (N >= 4096) GNMS_BITMASK(true, 16, 8);
    else GNMS_BITMASK(true, 8, 8);
#undef GNMS_BITMASK
    GNMS_CHECK_LAUNCH();
    if (full == 1) {
        const int nb = (N + 63) / 64;
        wsym_check_kernel<<<dim3(gnms_div_up(nb * (nb + 1) / 2, 4), B), 256, 0, st>>>(N, counts, ws, L);
        GNMS_CHECK_LAUNCH();
    }
    return GNMS_OK;
}

// grouping pipeline K2..K4 (shared by gnms_forward and gnms_get_groups)
int run_grouping(const float* iou, int B, int N, int64_t ld, const int32_t* counts, float thr, char* ws, const gnms_ws_layout& L,
                 hipStream_t st) {
    const int sym = matrix_sym_detection(N) ? 2 : 0;
    int rc0 = launch_bitmask(iou, B, N, ld, counts, thr, ws, L, st, sym ? 1 : 0);
    if (rc0) return rc0;
    const size_t lds = leaders_lds_bytes(N);
    int rc = allow_lds(leaders_kernel, lds);
    if (rc) return rc;
    { const int spw = leaders_chain_wgs(N, sym); leaders_kernel<<<B * spw, 1024, lds, st>>>(N, counts, ws, L, sym, B, spw); }
    GNMS_CHECK_LAUNCH();
    attribute_kernel<false><<<dim3(L.NB, B), 64, 0, st>>>(iou, (long)ld, N, counts, thr, ws, L, sym);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

}  // namespace

extern "C" size_t gnms_workspace_bytes(int B, int N, const gnms_params* params) {
    if (B <= 0 || N <= 0) return 0;
    size_t bytes = gnms_make_layout(N).per_image * (size_t)B;
    if (params && !params->group_boxes) bytes += gnms::ungrouped_scratch_bytes(B, N);   // the sorted strictly-lower-triangular matrix
    return bytes;
}

namespace {

// ------------------------------------------------------------------------------------------------
// The matrix write as a ROLE inside the launch of the per-image chain (masked from-boxes layer, gnms_forward_with_iou2d).
// Nothing in that layer reads the matrix, so the write is independent of the chain sorts -> threshold bits -> K3..K6.  As launches of
// their own the chain K3..K6 (one workgroup per image: 8 of 256 CUs at B = 8) ran strictly behind the write; as two streams the fork
// and join cost more than the overlap bought (LABNOTES.md 3.2d).  tail_write_kernel is both in ONE launch: the first B x nsb workgroups ARE
// the chain (the device functions of tail_kernel), the others write the matrix.  The launch asks for the chain's LDS (> 80 KiB), so a
// CU holds ONE workgroup: the chain workgroups are dispatched first and have a CU each to themselves (beside the write they run
// within 10 % of their stand-alone time; sharing a CU with streaming waves they ran 1.5-3x slower), the other CUs stream the matrix;
// when a chain workgroup retires, a writer still waiting in the grid takes its CU.
// A chunk = 16 wave tiles of tile_rows x 256 entries (tile_rows full rows of one image at N = 4096), written by the 16 waves of a
// workgroup with iou2d_tile -- the tile code of gnms_iou2d, so the matrix is the same bit for bit.
// Measured and dropped: slices of the write ALSO inside the sort launches and the bit-matrix launch (7 % / 5 % / 10-20 % of the rows,
// behind those kernels' own workgroups in the grid).  The sort and bit-matrix workgroups then share their CUs with streaming waves
// and slow down by more than the slices save: sort runs 7.8 -> 15.6 us, merges 5.9 -> 15.0, bit matrix 24 -> 46, chain + rest of the
// write 114 -> 98 (B = 8, N = 4096: 0.172 -> 0.173-0.184 ms per step for every split tried).
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline int write_chunk_count(int N, int nimg, int tile_rows, int row0, int row_end) {
    if (row_end <= row0) return 0;
    const long tiles = (long)nimg * ((N + gnms_iou::kWaveCols - 1) / gnms_iou::kWaveCols) * ((row_end - row0 + tile_rows - 1) / tile_rows);
    return (int)((tiles + 15) >> 4);
}

// what the writers compute: SRC = kFromBoxes: `in` = boxes [B][N][4], lib/core.py iou (iou2d_tile);
// SRC = kFromRecords: `in` = corner-AABB records [B][N][12], 0.5 * (1 + GIoU3D) with the guard band around `thr` (nms_overlap3d_tile)
template <bool VEC, int SRC>
__device__ __forceinline__ void write_chunk(const float* __restrict__ in, int N, float* __restrict__ out, long ld, int nimg, int tile_rows,
                                            int row0, int row_end, float thr, int chunk) {
    using namespace gnms_iou;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncc = (N + kWaveCols - 1) / kWaveCols;
    const int nrt = (row_end - row0 + tile_rows - 1) / tile_rows;
    const int per_img = ncc * nrt;
    const int t = chunk * 16 + wave;                               // tiles numbered image-major, column chunk fastest
    if (t >= per_img * nimg) return;
    const int img = t / per_img;
    const int r = t - img * per_img;
    const int rt = r / ncc, cc = r - rt * ncc;
    if (SRC == kFromRecords)
        gnms_iou3d::nms_overlap3d_tile<VEC>(in, in, N, N, out, ld, img, row0 + rt * tile_rows, cc * kWaveCols, lane, tile_rows, row_end, thr);
    else
        iou2d_tile<VEC>(in, in, N, N, out, ld, img, row0 + rt * tile_rows, cc * kWaveCols, lane, tile_rows, row_end);
}

// PERSISTENT writers: one workgroup stays on its CU and claims chunk after chunk from `counter` (zeroed by the sort kernels of this
// call), one claim ahead so that the atomic's round trip hides behind the chunk in progress.  Claims are the ONLY cross-workgroup
// traffic of the launch: anything that needs a release/acquire fence between workgroups (the whole chain sorts -> bits -> K3..K6 as
// tasks of one persistent launch was the plan) is out of the question beside the write stream -- one __threadfence() per chunk
// (buffer_wbl2 + buffer_inv at agent scope) took this launch from 0.115 to 1.13 ms.  (Ordinary workgroups of one chunk each
// leave the CU empty between a workgroup's last wave and the next workgroup's start, and with one workgroup per CU nothing covers
// that gap: 116 against 107 us.  One claim per WAVE tile serialises on the counter: 16384 device-scope atomics on one address took
// 430 us.)
template <bool VEC, int SRC>
__device__ __forceinline__ void writers_persistent(const float* __restrict__ in, int N, float* __restrict__ out, long ld, int nimg,
                                                   int tile_rows, int row0, int row_end, float thr, int* counter) {
    __shared__ int s_next[2];
    const int nchunks = write_chunk_count(N, nimg, tile_rows, row0, row_end);
    if (threadIdx.x == 0) s_next[0] = atomicAdd(counter, 1);
    __syncthreads();
    int cur = s_next[0], ph = 0;
    while (cur < nchunks) {
        int nx = 0;
        if (threadIdx.x == 0) nx = atomicAdd(counter, 1);          // the claim after this one, in flight during the chunk
        write_chunk<VEC, SRC>(in, N, out, ld, nimg, tile_rows, row0, row_end, thr, cur);
        if (threadIdx.x == 0) s_next[ph ^ 1] = nx;
        __syncthreads();
        ph ^= 1;
        cur = s_next[ph];
    }
}

// STAGED writers (2D, N <= 4096): what keeps the persistent writers above from the rate of a plain store stream is vmcnt.  On gfx9
// loads, stores and returning atomics retire through ONE in-order counter, so the wave that waits for its next tile's boxes (5 loads)
// first waits for the 16 stores of the tile before, and thread 0, waiting for its claim, drains wave 0's stores while 15 waves sit at
// the barrier -- at 16 waves per CU nothing covers those gaps (tail_write_kernel 0.115 ms where gnms_iou2d's many small workgroups
// take 0.100).  Here the steady state has no vector load at all:
//   * the image's boxes are staged in LDS once per (workgroup, image) -- the chain's dynamic LDS, unused by writer workgroups -- and a
//     tile reads its column and row boxes from there (lgkmcnt);
//   * every image has its own claim counter (misc[5] of its workspace, zeroed by its sort) and a workgroup starts on image
//     (index mod B), moving on when that image is exhausted: 1-2 stagings per workgroup instead of B;
//   * the claim for the unit after next is issued BEFORE the tile's stores and consumed after them, the rows unrolled so that the
//     compiler counts the stores behind it and waits with vmcnt(8), not vmcnt(0).  (The atomic's address is made lane-dependent on
//     purpose: with a wave-uniform address the compiler's atomic optimizer wraps it in a readfirstlane that needs the result at once.)
// A unit = 16 wave tiles of 8 rows x 256 columns, numbered row band major inside an image.  (Rows per tile, B = 8: 16 -> 8 took the
// launch at N = 4096 from 102.5 to 93.0 us and the large-image launch at N = 16384 from 1.57 to 1.45 ms -- twice as many, shorter units
// even out what 2048 units on 248 workgroups leave uneven; 4 rows lose again, 111 us / 1.77 ms.)
constexpr int kStagedRows = 8;

// claim0 / claim_stride: image i's claim counter is claim0[i * claim_stride] (the layer: misc[5] of the image's workspace; gnms_iou2d:
// a zeroed array of its own); first_wg: blockIdx.x of the first writer workgroup of the launch
template <bool VEC>
__device__ __forceinline__ void writers_staged_2d(const float* __restrict__ boxes, int N, float* __restrict__ out, long ld, int nimg, int* claim0,
                                                  size_t claim_stride, const int first_wg) {
    using namespace gnms_iou;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sbox = reinterpret_cast<float4*>(smem);                  // [N] boxes of the staged image
    __shared__ int s_claim[2];                                       // (image << 16 | unit) or -1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncc = (N + kWaveCols - 1) / kWaveCols;
    const int nrt = (N + kStagedRows - 1) / kStagedRows;
    const int units = (ncc * nrt + 15) >> 4;
    // thread 0 only: the image it claims from and how many images it has not yet seen exhausted
    int cur_img = ((int)blockIdx.x - first_wg) % nimg, left = nimg;
    // never 1 at run time, but not provably 0 either: keeps the claim's address lane-dependent in the compiler's eyes
    const int lane_dep = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 6);
    auto counter = [&](int img) { return claim0 + (size_t)img * claim_stride + lane_dep; };
    auto claim_now = [&]() -> int {                                  // the next unit of cur_img, or of the next image that has one
        while (left > 0) {
            const int u = atomicAdd(counter(cur_img), 1);
            if (u < units) return (cur_img << 16) | u;
            cur_img = cur_img + 1 == nimg ? 0 : cur_img + 1;
            --left;
        }
        return -1;
    };
    if (tid == 0) s_claim[0] = claim_now();
    __syncthreads();
    int cur = s_claim[0], ph = 0, staged = -1;
    // PLAIN images (every box divides plainly, iou_tile.h -- pixel boxes always do; decided once per staging): the tile runs
    // iou2d_rows_plain, and the lane's column boxes stay in registers from unit to unit while the wave keeps its column chunk (at
    // N = 4096 always: a unit is one band of 16 chunks) -- gathering them, their areas and the plain test were 17 % of a tile's VALU work.
    bool img_plain = false;
    int cols_of = -1;                                                // the column chunk `cp` holds (of the staged image)
    ColPairs cp;
    const bool fast_ok = VEC && (N & 3) == 0 && (N % kStagedRows) == 0;
    while (cur >= 0) {
        const int img = cur >> 16, u = cur & 0xffff;
        if (img != staged) {                                         // (every wave finished reading the old image before the last barrier)
            const float4* b4 = reinterpret_cast<const float4*>(boxes) + (size_t)img * N;
            bool ok = true;
            for (int i = tid; i < N; i += 1024) { const float4 v = b4[i]; sbox[i] = v; ok = ok && box_divides_plainly(v); }
            staged = img;
            cols_of = -1;
            img_plain = __syncthreads_and(ok) != 0;
        }
        int pre = 0;
        const bool claims = tid == 0 && left > 0;
        const int t = u * 16 + wave;                                 // (wave 0's tile always exists: it carries the claim)
        if (t < ncc * nrt) {
            const int rt = t / ncc, cc = t - rt * ncc;
            const int c0 = cc * kWaveCols;
            if (fast_ok && img_plain && c0 + 4 * kStagedRows <= N) { // (the first kStagedRows lanes hold the rows AND must own columns)
                if (cc != cols_of) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) colpairs_set(cp, j, sbox[min(c0 + 4 * lane + j, N - 1)]);
                    cols_of = cc;
                }
                const float4 ra = sbox[rt * kStagedRows + (lane & (kStagedRows - 1))];
                if (c0 + 4 * lane < N) {
                    if (claims) pre = atomicAdd(counter(cur_img), 1);                // in flight ahead of this tile's stores
                    float* o0 = out + (size_t)img * N * ld + (size_t)(rt * kStagedRows) * ld + c0 + 4 * lane;
                    iou2d_rows_plain<kStagedRows>(cp, ra, o0, ld);
                    asm volatile("" :: "v"(pre));                                    // every path waits for it here: vmcnt(8) on a full tile
                }
            } else {
                iou2d_tile_staged<VEC, kStagedRows>(sbox, 0, sbox + rt * kStagedRows, N, N, out + (size_t)img * N * ld, ld, rt * kStagedRows, c0, lane,
                    [&] { if (claims) pre = atomicAdd(counter(cur_img), 1); },
                    [&] { asm volatile("" :: "v"(pre)); });
            }
        }
        if (tid == 0) {
            int nx = -1;
            if (left > 0) {
                if (pre < units) nx = (cur_img << 16) | pre;
                else { cur_img = cur_img + 1 == nimg ? 0 : cur_img + 1; --left; nx = claim_now(); }
            }
            s_claim[ph ^ 1] = nx;
        }
        __syncthreads();
        ph ^= 1;
        cur = s_claim[ph];
    }
}

// SYMMETRIC writers (3D, round 3): the write role of tail_write_kernel for the cuboid overlap -- every unordered pair evaluated once
// (iou3d_sym.h), the all-pairs 3D writers being VALU-bound at the one workgroup per CU this launch runs at (3.2e).  A persistent
// 16-wave workgroup claims 128 x 128 macro tiles (upper triangle of every image, image-major) one claim ahead and alternates between
// TWO LDS tiles, so that ONE barrier per macro tile suffices: it publishes the tile for the mirrored pass AND the next claim, and the
// waves that finish their mirrored stores early already compute the next tile into the other buffer.
template <bool NT>
__device__ __forceinline__ void writers_sym_persistent(const float* __restrict__ rec, int N, float* __restrict__ out, long ld, int nimg, float thr,
                                                       int* counter) {
    using namespace gnms_iou3d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tile0 = reinterpret_cast<float*>(smem);           // two LDS macro tiles, kSymTileBytes apart
    __shared__ int s_next[2];
    const int nt = (N + kSymT - 1) / kSymT;
    const int tpi = sym_tiles_per_image(N), total = tpi * nimg;
    if (threadIdx.x == 0) s_next[0] = atomicAdd(counter, 1);
    __syncthreads();
    int cur = s_next[0], ph = 0;
    while (cur < total) {
        int nx = 0;
        if (threadIdx.x == 0) nx = atomicAdd(counter, 1);          // the claim after this one, in flight during the tile
        const int img = cur / tpi;
        int I, J;
        sym_tile_of(cur - img * tpi, nt, &I, &J);
        const float* r = rec + (size_t)img * N * kRec;
        float* o = out + (size_t)img * N * ld;
        float* const tile = tile0 + (size_t)ph * (kSymTileBytes / sizeof(float));
        sym_tile_compute<16, NT>(r, N, o, ld, I, J, thr, tile);
        if (threadIdx.x == 0) s_next[ph ^ 1] = nx;
        __syncthreads();                                            // the tile is in LDS, the next claim is known; buffer ph ^ 1 is free
        if (I != J) sym_tile_mirror<16, NT>(N, o, ld, I, J, tile);
        ph ^= 1;
        cur = s_next[ph];
    }
}

// gnms_iou2d of a box set WITH ITSELF (what both reference call sites compute: iou(boxes, boxes)), N <= 4096: the staged writers as a
// launch of their own -- the same claimed units, tile body and cached columns as in tail_write_kernel, the claim counters in a zeroed
// array that lives for the call.  (A version that dealt the units statically -- round robin per image, no counters -- reached 0.55 of
// the HBM peak where this one reaches 0.71 and the 64-row tiles of iou2d_kernel 0.68-0.69: the claims keep an image's write frontier
// compact when workgroups drift, and the workgroups that run out of units on their image finish the others'.)
// `claims`: one slot of the library's per-device claim ring (claim_ring_slot): [nimg] counters 64 ints apart, then an exit counter.  The
// slot is all zero when the launch starts, and the LAST workgroup to leave zeroes it again (device-scope exchanges, the path the claims
// themselves take) -- no memset in front of the launch and no allocation per call.
template <bool VEC>
__global__ __launch_bounds__(1024) void iou2d_self_kernel(const float* __restrict__ boxes, int N, int nimg, float* __restrict__ out, long ld,
                                                          int* __restrict__ claims) {
    writers_staged_2d<VEC>(boxes, N, out, ld, nimg, claims, 64, 0);      // (a counter per 256 bytes: each in an L2 line of its own)
    if (threadIdx.x == 0) {                                                 // (thread 0 issued every claim of this workgroup and has consumed them all)
        int* done = claims + (size_t)nimg * 64;
        if (atomicAdd(done, 1) == (int)gridDim.x - 1) {
            for (int i = 0; i <= nimg; ++i) atomicExch(claims + (size_t)i * 64, 0);
        }
    }
}

// LARGE images (N > 4096): the matrix write as a launch of its own on the side stream (3.2d), in the same geometry -- persistent
// workgroups of 16 waves, 8 rows x 256 columns per wave -- because that geometry is what the store stream likes: a plain fill written
// this way reaches 5.7-5.8 TB/s at N = 4096 ... 16384 where a linear grid-stride fill and gnms_iou2d's 64-row tiles reach 4.6-4.8
// (tools/kernel_times.py).  A unit = one row band (8 rows) of one column group (<= 16 wave tiles; the groups of an image are balanced) of one image, numbered
// (image, column group) major; every workgroup takes a contiguous range of them (no chain workgroup in this launch and nothing to
// balance, so no claims).  The column group's boxes are staged in LDS when the (image, group) changes; the row boxes of a workgroup's NEXT unit
// are loaded by wave 0 before the stores of the current one and parked in LDS after them (vmcnt(8), as the claim above).
template <bool VEC>
__global__ __launch_bounds__(1024) void write_staged_kernel(const float* __restrict__ A, const float* __restrict__ boxes, int M, int N, int nimg,
                                                            float* __restrict__ out, long ld) {
    using namespace gnms_iou;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sbox = reinterpret_cast<float4*>(smem);                  // [4096] column boxes of the staged (image, column group)
    float4* srow = sbox + 4096;                                      // [2][16] row boxes of the current / the next unit
    const float4* b4 = reinterpret_cast<const float4*>(boxes);        // columns: boxes [nimg][N][4]
    const float4* a4 = reinterpret_cast<const float4*>(A);            // rows: A [nimg][M][4] (the layer: the same boxes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncc = (N + kWaveCols - 1) / kWaveCols;                 // wave tiles per row band
    const int ncg = (ncc + 15) >> 4;                                 // column groups ...
    const int tg = (ncc + ncg - 1) / ncg;                            // ... of tg <= 16 wave tiles each, balanced (17 tiles: 9 + 8, not 16 + 1)
    const int nb = (M + kStagedRows - 1) / kStagedRows;
    const long units = (long)nimg * ncg * nb;
    auto row_box = [&](long u, int r) {
        const long pair = u / nb;
        const int band = (int)(u - pair * nb), img = (int)(pair / ncg);
        return a4[(size_t)img * M + min(band * kStagedRows + r, M - 1)];
    };
    // a contiguous range of units per workgroup: at most a couple of stagings each, and 248 sequential store streams are what the
    // memory likes best (plain fill in this order: 5.5 / 6.0 / 5.9 TB/s at N = 4096 / 8192 / 16384)
    // (ranges of equal WORK, not of equal length: a unit of a ragged last column group keeps fewer than 16 waves busy and ends sooner)
    const int tlast = ncc - tg * (ncg - 1);                          // wave tiles in the last group
    const long wimg = (long)nb * ncc, wfull = (long)nb * tg;
    auto unit_of_work = [&](long s) {                                // the unit that holds tile-work offset s (units in (image, group) major order)
        const long img = s / wimg, rem = s - img * wimg;
        const long g = min((long)(ncg - 1), rem / wfull);
        const long band = (rem - g * wfull) / (g == ncg - 1 ? tlast : tg);
        return (img * ncg + g) * nb + band;
    };
    const long wtot = wimg * nimg;
    long u = unit_of_work(wtot * blockIdx.x / gridDim.x);
    const long uend = blockIdx.x + 1 == gridDim.x ? units : unit_of_work(wtot * (blockIdx.x + 1) / gridDim.x);
    if (u < uend && tid < kStagedRows) srow[tid] = row_box(u, tid);
    int ph = 0;
    long staged = -1;
    for (; u < uend; ++u) {
        const long pair = u / nb;
        const int band = (int)(u - pair * nb), img = (int)(pair / ncg), g = (int)(pair - (long)img * ncg);
        if (pair != staged) {                                        // (the barrier that ended the last unit covers the old contents)
            const int c0g = g * tg * kWaveCols, ncol = min(tg * kWaveCols, N - c0g);
            for (int i = tid; i < ncol; i += 1024) sbox[i] = b4[(size_t)img * N + c0g + i];
            staged = pair;
            __syncthreads();
        }
        const long un = u + 1;
        const bool fetch = tid < kStagedRows && un < uend;
        float4 nrow = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c0 = (g * tg + wave) * kWaveCols;
        if (wave < tg && c0 < N)
            iou2d_tile_staged<VEC, kStagedRows>(sbox, g * tg * kWaveCols, srow + ph * kStagedRows, M, N, out + (size_t)img * M * ld, ld, band * kStagedRows, c0, lane,
                [&] { if (fetch) nrow = row_box(un, tid); },
                [&] { asm volatile("" :: "v"(nrow.x), "v"(nrow.y), "v"(nrow.z), "v"(nrow.w)); });
        if (fetch) srow[(ph ^ 1) * kStagedRows + tid] = nrow;
        __syncthreads();
        ph ^= 1;
    }
}

// the launch; reserve: CUs left without a writer workgroup for the layer's one-workgroup-per-image kernels on the caller's stream
int launch_write_staged(const float* a, const float* b, int B, int M, int N, float* out, int64_t ld, int reserve, hipStream_t st) {
    const int cus = device_cu_count();
    int grid = cus - reserve;
    if (grid < cus / 2) grid = cus / 2;
    const size_t lds = 96 * 1024;                                    // > 80 KiB: one writer workgroup per CU (it uses 4096 + 32 boxes = 64.5 KiB)
    int rc;
    if ((rc = allow_lds(write_staged_kernel<true>, lds))) return rc;
    gnms_launch_prof(kProfMatrixWrite, write_staged_kernel<true>, dim3((unsigned)grid), dim3(1024), lds, st, a, b, M, N, B, out, (long)ld);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
}  // namespace
// gnms_iou2d's large-matrix path (iou_kernels.hip): the persistent writers' geometry reaches 5.5-5.6 TB/s where its 64-row tiles reach 4.6-5.2
bool gnms_internal_iou2d_wants_staged(int B, int M, int N, int64_t ld, const float* out) {
    if ((ld % 4) != 0 || (N % 4) != 0 || ((uintptr_t)out % 16) != 0) return false;
    // (N <= 4096: the per-unit row fetch and barrier cost more than the geometry gains -- B = 8, M = N = 4096: 108-114 us against the
    // 64-row tiles' 102; N = 16384: 1.61 ms against 1.85)
    const long units = (long)B * ((M + kStagedRows - 1) / kStagedRows) * ((N + 4095) / 4096);
    return N > 4096 && units >= 4L * device_cu_count();
}
// a == b, N <= 4096: iou2d_self_kernel (B = 8, N = 4096: 0.68-0.69 -> 0.71 of the HBM peak)
bool gnms_internal_iou2d_wants_self(const float* a, const float* b, int B, int M, int N, int64_t ld, const float* out) {
    if (a != b || M != N || N > 4096 || B > 127) return false;
    const long units = (long)B * ((N + kStagedRows - 1) / kStagedRows) * ((N + 4095) / 4096);
    return units >= 8L * device_cu_count();
}
namespace {
// The claim counters of iou2d_self_kernel live in zeroed SLOTS of a per-device pool that the library allocates once (the first eager call on
// a device; 2 MiB) -- a launch leaves its slot zeroed (see the kernel), so a call neither allocates nor memsets.  Who may share a slot
// (a host-side round robin handed the same slot to launches that can run at the same time -- another stream, or a replayed graph
// that had the pointer baked in -- which then shared claim and exit counters and each wrote only part of its matrix):
//   * eager launches: ONE slot per (device, stream) -- launches of one stream run one after the other;
//   * a launch that is being CAPTURED: a slot of its own, never handed out again -- whatever the graph later runs beside, nobody else
//     counts in its slot (two replays of the same executable graph do not overlap);
//   * no slot left, or a capture before the pool exists (no allocation inside a capture): the caller falls back to iou2d_kernel.
// Images per call <= kClaimImgs (larger batches take iou2d_kernel).
constexpr int kClaimSlots = 64, kClaimImgs = 127;
constexpr size_t kClaimSlotInts = (size_t)(kClaimImgs + 1) * 64;
constexpr int kNoClaimSlot = 1;                                      // (> 0: not an error code)
int claim_slot_for(hipStream_t st, int** slot) {
    struct Pool { int* base = nullptr; int eager_next = 0, capture_next = kClaimSlots - 1; std::map<hipStream_t, int> by_stream; };
    static std::mutex mu;
    static std::map<int, Pool> pools;
    int dev = 0;
    GNMS_CHECK_HIP(hipGetDevice(&dev));
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lock(mu);
    Pool& R = pools[dev];
    if (!R.base) {
        if (capturing) return kNoClaimSlot;
        int* p = nullptr;
        GNMS_CHECK_HIP(hipMalloc((void**)&p, kClaimSlots * kClaimSlotInts * sizeof(int)));
        const hipError_t e = hipMemset(p, 0, kClaimSlots * kClaimSlotInts * sizeof(int));
        if (e != hipSuccess) { (void)hipFree(p); GNMS_CHECK_HIP(e); }
        R.base = p;
    }
    // (the pool is finite: a slot per stream that ever called, one per captured launch, none reclaimed.  When it runs out the callers fall
    // back to their slower forms -- said once on stderr, so that the cliff is visible: ADVICE r5)
    auto exhausted = [] {
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            fprintf(stderr, "[groomed_nms_hip] all %d claim slots of this device are taken (one per stream that called, one per captured launch): "
                            "gnms_iou2d and the one-call entry on small images fall back to their slower kernels from here on\n", kClaimSlots);
        return kNoClaimSlot;
    };
    int idx;
    if (capturing) {
        if (R.capture_next < R.eager_next) return exhausted();
        idx = R.capture_next--;
    } else {
        auto it = R.by_stream.find(st);
        if (it != R.by_stream.end()) idx = it->second;
        else {
            if (R.eager_next > R.capture_next) return exhausted();
            idx = R.by_stream[st] = R.eager_next++;
        }
    }
    *slot = R.base + (size_t)idx * kClaimSlotInts;
    return GNMS_OK;
}
}  // namespace
// (returns 1 -- not an error -- when no claim slot is to be had: gnms_iou2d then runs iou2d_kernel)
int gnms_internal_iou2d_self(const float* boxes, int B, int N, float* out, int64_t ld, hipStream_t st) {
    const int cus = device_cu_count();
    int grid = cus - 8;                                              // (alone on the machine the stream likes every CU: 248 -> 0.71, 200 -> 0.64)
    if (grid < 1) grid = 1;
    int* claims = nullptr;
    int rc = claim_slot_for(st, &claims);
    if (rc) return rc;
    size_t lds = (size_t)N * 16;
    if (lds < 96 * 1024) lds = 96 * 1024;                            // > 80 KiB: one workgroup per CU
    const bool vec = (ld % 4 == 0) && ((uintptr_t)out % 16 == 0);
    if (vec) {
        if ((rc = allow_lds(iou2d_self_kernel<true>, lds))) return rc;
        gnms_launch_prof(kProfMatrixWrite, iou2d_self_kernel<true>, dim3((unsigned)grid), dim3(1024), lds, st, boxes, N, B, out, (long)ld, claims);
    } else {
        if ((rc = allow_lds(iou2d_self_kernel<false>, lds))) return rc;
        gnms_launch_prof(kProfMatrixWrite, iou2d_self_kernel<false>, dim3((unsigned)grid), dim3(1024), lds, st, boxes, N, B, out, (long)ld, claims);
    }
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
int gnms_internal_iou2d_staged(const float* a, const float* b, int B, int M, int N, float* out, int64_t ld, hipStream_t st) {
    return launch_write_staged(a, b, B, M, N, out, ld, 0, st);
}
namespace {

// chain_src: what the chain's single overlaps come from (the boxes for SRC = kFromBoxes; unused for kFromRecords: the workspace copy
// of the records); write_src: the writers' input (the boxes / the batch's contiguous records)
template <bool VEC, int E, int SRC>
__global__ __launch_bounds__(1024) void tail_write_kernel(const float* __restrict__ chain_src, const float* __restrict__ write_src, int N,
                                                          const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L, int Ppow2,
                                                          float* __restrict__ prob, long long* __restrict__ valid,
                                                          long long* __restrict__ invalid, int* __restrict__ nvalid, int* __restrict__ ninvalid,
                                                          int nimg, float* __restrict__ out, long ld, int tile_rows, int row0, int row_end,
                                                          int staged, int fast) {
#ifdef GNMS_HWID    // developer experiment (tools/hwid_map.py): where the dispatcher put every workgroup of this launch
    if (threadIdx.x == 0) {
        unsigned* dbg = reinterpret_cast<unsigned*>(img_ptrs(ws, L, 0).rec);
        dbg[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(63492);       // HW_ID
        dbg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(63508);   // XCC_ID
    }
#endif
    const int spw = leaders_chain_wgs(N, 1);                          // scan workgroups per image (the from-boxes / from-records bit matrices are symmetric)
    const int cwg = spw + (fast ? 1 : 0);                             // + the fast tail's CSR workgroup (csr_build_body)
    if ((int)blockIdx.x < nimg * cwg) {
        int b;
        GNMS_TINIT();
#ifdef GNMS_TIMING
        long long tt__ = (long long)__builtin_amdgcn_s_memtime();
#define GNMS_TW_ACC(slot) do { long long n__ = (long long)__builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && b == 0) gnms::gnms_tbuf()[slot] += n__ - tt__; tt__ = n__; } while (0)
#else
#define GNMS_TW_ACC(slot) do {} while (0)
#endif
        if constexpr (E <= 4) {
            if (fast) {                                               // masked groups: leaders_sb_body<STAGE> -> fast_final_body ‖ csr_build_body
                if ((int)blockIdx.x >= nimg * spw) { csr_build_body<E>(N, counts, ws, L, (int)blockIdx.x - nimg * spw); return; }
                const int last = leaders_chain<SRC>(N, counts, ws, L, nimg, spw, (int)blockIdx.x, 1, &b, chain_src, (long)N, P.nms_threshold,
                                                    P.temperature, P.pruning_method, Ppow2);
                if (last) { GNMS_TW_ACC(5); fast_final_body<E, SRC>(chain_src, N, (long)N, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, b, last); GNMS_TW_ACC(15); }
                GNMS_TFLUSH(ws, L, b);
                return;
            }
        }
        if (!leaders_chain(N, counts, ws, L, nimg, spw, (int)blockIdx.x, 1, &b)) return;   // (the image's other scan workgroups)
        __syncthreads();
        GNMS_TW_ACC(5);
        if (E <= 4 && P.mask_group_boxes) {                           // K4's rest rides in K5 (groups_body, FUSE), K6 starts from LDS
            if constexpr (E <= 4) {
                GNMS_TW_ACC(6);
                groups_body<E, SRC, true>(chain_src, N, (long)N, counts, P, ws, L, Ppow2, b);
                gnms::lds_barrier();
                GNMS_TW_ACC(7);
                finalize_body<E, true>(N, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, b);
                GNMS_TW_ACC(15);
            }
            return;
        } else {
            attribute_image<SRC>(chain_src, (long)N, N, counts, P.nms_threshold, ws, L, b, 1);
            __syncthreads();
            GNMS_TW_ACC(6);
            groups_body<E, SRC>(chain_src, N, (long)N, counts, P, ws, L, Ppow2, b);
        }
        if (!P.mask_group_boxes) return;                              // unmasked groups: the solves (a launch of their own) come before K6
        __syncthreads();
        GNMS_TW_ACC(7);
        finalize_body<E>(N, counts, P, ws, L, Ppow2, prob, valid, invalid, nvalid, ninvalid, b);
        GNMS_TW_ACC(15);
        return;
    }
    const int first_writer = nimg * cwg;
    if (SRC == kFromBoxes && staged) { writers_staged_2d<VEC>(write_src, N, out, ld, nimg, img_ptrs(ws, L, 0).misc + 5, L.per_image / sizeof(int), first_writer); return; }
    if (SRC == kFromRecords && staged == 2) { writers_sym_persistent<true>(write_src, N, out, ld, nimg, P.nms_threshold, img_ptrs(ws, L, 0).misc + 5); return; }
    writers_persistent<VEC, SRC>(write_src, N, out, ld, nimg, tile_rows, row0, row_end, P.nms_threshold, img_ptrs(ws, L, 0).misc + 5);
}

// gnms_forward_with_iou2d on small images as ONE launch (nms_one_launch.h): [matrix writers] [sort] [table from the boxes] [chain] [CSR].
// The writers are writers_staged_2d with a slot of the claim ring (as iou2d_self_kernel: zero at the start, re-zeroed by the last writer
// to leave).
__global__ __launch_bounds__(1024) void one_launch_boxes_kernel(const float* __restrict__ scores, const float* __restrict__ boxes, int N,
                                                                const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L,
                                                                float* __restrict__ prob, long long* __restrict__ valid, long long* __restrict__ invalid,
                                                                int* __restrict__ nvalid, int* __restrict__ ninvalid, long long* __restrict__ order_out,
                                                                int B, int kpw, int tpw, float* __restrict__ out, long ld, int* __restrict__ claims,
                                                                int nwriters) {
    GNMS_TINIT();
    // tpw == 0: the table with its sources in x order (one_launch_bits_from_boxes_x) -- [sort] [x sort] [table: NB a image] [chain] [CSR] [writers]:
    // two sort roles of B * nsort workgroups are a round of the machine by themselves, so the writers come LAST and start as the sorts retire
    const bool xt = tpw == 0;
    const int NP = (N + 63) & ~63, nsort = NP / kpw, nbits = xt ? L.NB : (L.NB * (L.NB + 1) / 2 + tpw - 1) / tpw;
    const int front = xt ? B * (2 * nsort + nbits + 2) : 0;
    int bx = (int)blockIdx.x, b;
    if (xt ? bx >= front : bx < nwriters) {
        // (rank-space table) the writers FIRST in the grid: they wait for nobody and are the longest role at N = 1024 (every workgroup of the launch holds a CU
        // to itself -- the chain's LDS -- so behind the sort and table workgroups they would only start when those retire: 36 us against 20)
        writers_staged_2d<true>(boxes, N, out, ld, B, claims, 64, front);
        if (threadIdx.x == 0) {                                             // (thread 0 issued every claim of this workgroup and has consumed them all)
            int* done = claims + (size_t)B * 64;
            if (atomicAdd(done, 1) == nwriters - 1) {
                for (int i = 0; i <= B; ++i) atomicExch(claims + (size_t)i * 64, 0);
            }
        }
        return;
    }
    if (xt) {
        if (bx < 2 * B * nsort) {
            const int role = bx >= B * nsort ? 1 : 0;
            bx -= role * B * nsort;
            b = bx / nsort;
            const unsigned tag = (unsigned)gnms_next_epoch(coh_load(img_ptrs(ws, L, b).misc + 8));
            if (kpw == 32) sort_count_body<32, true>(scores, boxes, N, counts, ws, L, order_out, 0, bx - b * nsort, b, role, tag);
            else sort_count_body<64, true>(scores, boxes, N, counts, ws, L, order_out, 0, bx - b * nsort, b, role, tag);
            return;
        }
        bx -= 2 * B * nsort;
        if (bx < B * nbits) {
            b = bx / nbits;
            const unsigned tag = (unsigned)gnms_next_epoch(coh_load(img_ptrs(ws, L, b).misc + 8));
            one_launch_bits_from_boxes_x(N, counts, P.nms_threshold, ws, L, b, bx - b * nbits, tag, nsort);
            return;
        }
        bx -= B * nbits;
        one_launch_chain_or_csr<kFromBoxes>(boxes, N, (long)N, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, B, bx, nsort, nbits);
        return;
    }
    bx -= nwriters;
    if (bx < B * nsort) {
        b = bx / nsort;
        const unsigned tag = (unsigned)gnms_next_epoch(coh_load(img_ptrs(ws, L, b).misc + 8));
        if (kpw == 32) sort_count_body<32, true>(scores, boxes, N, counts, ws, L, order_out, 0, bx - b * nsort, b, 0, tag);
        else sort_count_body<64, true>(scores, boxes, N, counts, ws, L, order_out, 0, bx - b * nsort, b, 0, tag);
        return;
    }
    bx -= B * nsort;
    if (bx < B * nbits) {
        b = bx / nbits;
        const unsigned tag = (unsigned)gnms_next_epoch(coh_load(img_ptrs(ws, L, b).misc + 8));
        one_launch_bits_from_boxes(N, counts, P.nms_threshold, ws, L, b, bx - b * nbits, tpw, tag, nsort);
        return;
    }
    bx -= B * nbits;
    one_launch_chain_or_csr<kFromBoxes>(boxes, N, (long)N, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, B, bx, nsort, nbits);
}

// bitmask_boxes_kernel: workgroups of 4 wave tiles, (row blocks) x (column chunks) tiles per image; 4 columns per lane
// (64 x 256 tiles) when that already gives every SIMD a couple of waves, else 1 column per lane (64 x 64 tiles)
int launch_bitmask_boxes(const float* boxes, int B, int N, const int32_t* counts, float thr, char* ws, const gnms_ws_layout& L, hipStream_t st) {
    const int NB = (N + 63) / 64;
    const long long tiles4 = (long long)B * NB * ((N + 255) / 256);
    if (N > 4096 && N % 1024 == 0) {
        // large images (round 4b): the scatter kernel with the row groups dealt to the XCDs (bitmask_boxes_pinned_kernel) -- a row of W is
        // >= 64 KiB here and completing its lines in ONE L2 costs less than collecting it in LDS first (one 16-wave workgroup per CU):
        // B = 8, N = 16384 280 -> 189 us (step 1.74 -> 1.66 ms), 8192 82 -> 61.  Row groups of 4 blocks where there are plenty of tiles
        // (16384: 288 / 217 / 189 / 184 us with 1 / 2 / 4 / 8).
        const int kbw = tiles4 >= 32768 ? 4 : 1;
        const unsigned gx = (unsigned)(gnms_div_up(gnms_div_up(NB, kbw), 8) * (((N + 255) / 256) / 4) * 8);
        if (kbw >= 4) bitmask_boxes_pinned_kernel<4><<<dim3(gx, 1, B), 256, 0, st>>>(boxes, N, counts, thr, ws, L);
        else bitmask_boxes_pinned_kernel<1><<<dim3(gx, 1, B), 256, 0, st>>>(boxes, N, counts, thr, ws, L);
    } else if (tiles4 >= 32768) {
        // large images (round 3): the LDS row buffer with a chunk loop -- one 16-wave workgroup per rank block, its waves walking the
        // column chunks, the full row of W leaving as ONE coalesced write (N <= GNMS_MAX_BOXES: the row fits 128 KiB).  The scatter
        // kernel it replaced issued N^2 / 64 scattered 8-byte stores per image: B = 8, N = 16384 step 1.907 -> 1.798 ms.
        const size_t lds = (size_t)L.NC * 8;
        int rc = allow_lds(bitmask_boxes_kernel<4, 1, true, true>, lds);
        if (rc) return rc;
        bitmask_boxes_kernel<4, 1, true, true><<<dim3(NB, 1, B), 1024, lds, st>>>(boxes, N, counts, thr, ws, L);
    } else if (tiles4 >= 2048 && (N + 255) / 256 <= 16) {
        // one 16-wave workgroup per rank block: words collected in an LDS copy of the row, written out coalesced
        // (from two workgroups per CU on: two rank blocks per workgroup -- half the column-side traffic, one workgroup per CU; see the body)
        if ((long)B * NB > (long)device_cu_count()) {
            const size_t lds = 2 * (size_t)L.NC * 8 + 4 * 1024 * sizeof(int);
            int rc = allow_lds(bitmask_boxes_kernel<4, 2, true>, lds);
            if (rc) return rc;
            bitmask_boxes_kernel<4, 2, true><<<dim3((NB + 1) / 2, 1, B), 1024, lds, st>>>(boxes, N, counts, thr, ws, L);
        } else
        bitmask_boxes_kernel<4, 1, true><<<dim3(NB, 1, B), 1024, (size_t)L.NC * 8 + 4 * 1024 * sizeof(int), st>>>(boxes, N, counts, thr, ws, L);   // + the ranks' stash
    } else if (tiles4 >= 2048) {
        bitmask_boxes_kernel<4, 1><<<dim3(gnms_div_up(NB * ((N + 255) / 256), 4), 1, B), 256, 0, st>>>(boxes, N, counts, thr, ws, L);
    } else {
        bitmask_boxes_kernel<1, 1><<<dim3(gnms_div_up(NB * NB, 4), 1, B), 256, 0, st>>>(boxes, N, counts, thr, ws, L);
    }
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// K1: stable descending score sort (+ the x-centre sort of the boxes when `boxes` is given).  One workgroup per image up to
// 1024 keys, the cooperative two-kernel sort above that.
// mode3d: 0 = `boxes` are 2D boxes (columns by x centre); >= 1 = pseudo boxes of cuboids whose records lie in the workspace (columns by
// (z band, x centre) with that many bands; the sort also leaves the records in column order, ImgPtrs::xrec)
int launch_sorts(const float* scores, const float* boxes, int B, int N, const int32_t* counts, char* ws, const gnms_ws_layout& L, int P2,
                 int64_t* order, hipStream_t st, int mode3d = 0) {
    int rc;
    const int roles = boxes ? 2 : 1;
    // up to 2048 keys: by counting, N / 64 workgroups per image and role (sort_count_kernel)
    constexpr bool count_sort = true;
    // (... and up to 4096 keys where its N / 64 workgroups per image and role are ONE round of the machine -- B <= 2: 256 compares per thread,
    // ~8 us in one launch against 14 us of runs + merge; from two rounds on the merge sort wins, LABNOTES R5.6)
    if (count_sort && (N <= 2048 || (N <= 4096 && (long)B * ((N + 63) / 64) * roles <= (long)device_cu_count()))) {
        const int NP = (N + 63) & ~63;
        if (NP % 128 == 0 && (long)B * (NP / 32) * roles <= (long)device_cu_count())   // half the compares per thread while the grid is one round
            sort_count_kernel<32><<<dim3(NP / 32, B, roles), 1024, (size_t)NP * 8, st>>>(scores, boxes, N, counts, ws, L, (long long*)order, mode3d);
        else
            sort_count_kernel<64><<<dim3(NP / 64, B, roles), 1024, (size_t)NP * 8, st>>>(scores, boxes, N, counts, ws, L, (long long*)order, mode3d);
        GNMS_CHECK_LAUNCH();
        return GNMS_OK;
    }
    if (P2 <= 1024) {
        sort_scores_kernel<1><<<dim3(B, roles), P2, (size_t)P2 * 8, st>>>(scores, N, counts, ws, L, P2, (long long*)order, boxes, mode3d);
        GNMS_CHECK_LAUNCH();
        return GNMS_OK;
    }
    const int R = P2 / 1024;
    const size_t lds = (size_t)P2 * 8;
    // (runs and merge as ONE launch with nonce-flag hand-offs was measured in round 3: 13.1 us against 7.8 + 6.2, the step unchanged -- a launch
    // boundary between two small kernels costs ~1 us; dropped in round 4, LABNOTES.md)
    sort_runs_kernel<<<dim3(R, B, roles), 1024, 0, st>>>(scores, boxes, N, counts, ws, L, P2, mode3d);
    GNMS_CHECK_LAUNCH();
#define GNMS_MERGE(RR)                                                                                                    \
    do {                                                                                                                  \
        if ((rc = allow_lds(sort_merge_kernel<RR>, lds))) return rc;                                                      \
        sort_merge_kernel<RR><<<dim3(RR, B, roles), 1024, lds, st>>>(scores, boxes, N, counts, ws, L, (long long*)order, mode3d); \
    } while (0)
    switch (R) {
        case 2: GNMS_MERGE(2); break;
        case 4: GNMS_MERGE(4); break;
        case 8: GNMS_MERGE(8); break;
        default: GNMS_MERGE(16); break;
    }
#undef GNMS_MERGE
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// K3..K6 as one launch or four?  Measured (HIP-graph replay, B=8): one launch wins 2-2.5 us per step up to N=2048 (three
// kernel boundaries less) and, with the general scan, loses 1.5 us at N=4096 (the attribution runs on one CU instead of 64).  Where the
// scan attributes as it goes (symmetric sources, `sym`), K4 rides in K5 and K6 starts from LDS (groups_body FUSE, E <= 4): one launch
// up to N = 4096.
bool use_tail_kernel(int N, int sym = 0) {
    return N <= 2048 || (sym && N <= 4096);
}

// chain workgroups per image when the matrix write runs beside the layer as a launch of its own on the side stream (N > 4096): ONE, which
// walks the image's super-blocks in order.  The write there is a static partition over (CUs - reserve) persistent workgroups and takes
// 1.45 ms at B = 8, N = 16384; the scan of a whole image on one CU (~1 ms beside it) hides behind that, while every further chain
// workgroup either costs the write a CU (32 reserved: write 1.52 ms, step 1.86) or, unreserved, does not find a CU before the writers
// retire and stalls the chain behind the write (16 per image, 8 reserved: step 1.82 against 1.74).
constexpr int kBesideChainWGs = 1;

// K3..K6 in one launch (masked groups); SRC/src: kFromMatrix (the matrix), kFromBoxes (the boxes), kFromRecords (src unused)
// the fast tail (nms_kernels.h): on unless GNMS_FAST_TAIL=0 (developer switch: the K5-proper path stays reachable for A/B runs and tests)
bool fast_tail_enabled() {
    static const bool on = [] { const char* e = getenv("GNMS_FAST_TAIL"); return !(e && e[0] == '0'); }();
    return on;
}

template <int BOXES>
int launch_tail(const float* src, int B, int N, int64_t ld, const int32_t* counts, const gnms_params& P, char* ws, const gnms_ws_layout& L,
                float* prob, int64_t* valid, int64_t* invalid, int32_t* nvalid, int32_t* ninvalid, hipStream_t st, int sym, int chain_cap = 0) {
    int P2 = next_pow2(N);
    if (P2 < 1024) P2 = 1024;                                   // the fused kernel always runs 1024 threads
    // (the fused K5 -> K6 hand-off, E <= 4, parks order[] and a copy of r2 behind the key region: 16 bytes per key)
    const size_t llds = leaders_lds_bytes(N), glds = (size_t)P2 * (P2 <= 4096 ? 16 : 8);
    size_t lds = llds > glds ? llds : glds;
    const int fast = (fast_tail_enabled() && fast_tail_ok(N, P, sym, chain_cap)) ? 1 : 0;
    if (fast && lds < fast_tail_lds_size(N, P2)) lds = fast_tail_lds_size(N, P2);
    GNMS_CHECK_ARG(sym != 3 || fast, "launch_tail: the in-launch symmetry check needs the fast tail");
    int rc;
    GNMS_DISPATCH_SORT(P2, {
        if ((rc = allow_lds(tail_kernel<E, BOXES>, lds))) return rc;
        const int spw = leaders_chain_wgs(N, sym, chain_cap);
        // sym 3: symmetry checkers in front of the chain (one 16-wave workgroup per 128 pairs of 64 x 64 bit blocks: fewer, so that the chain workgroups behind them in the grid start sooner, at most the CUs the chain leaves)
        int nchk = 0;
        if (sym == 3) {
            const long nb = (N + 63) / 64, pairs = (long)B * nb * (nb + 1) / 2;
            const int room = device_cu_count() - B * (spw + fast);
            nchk = (int)std::min<long>(std::max(room, 8), (pairs + 127) / 128);
            if (nchk < 1) nchk = 1;
        }
        tail_kernel<E, BOXES><<<nchk + B * (spw + fast), 1024, lds, st>>>(src, N, (long)ld, counts, P, ws, L, P2, prob, (long long*)valid, (long long*)invalid, nvalid,
                                                          ninvalid, sym, B, spw, fast, nchk);
    });
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// rows per wave tile of the write role: 16 measured best at B = 8, N = 4096 (0.168 ms per step; 8: 0.174, 32: 0.175, 64: 0.193 -- the last
// chunks of a launch end together only if chunks are short)
constexpr int kFusedTileRows = 16;

// 3D: the write role of that launch as symmetric writers (writers_sym_persistent) -- above N = 1024, where an image has enough
// macro tiles.
bool sym_writers_in_tail_launch(int N, int64_t ld, const float* out) {
    if (!gnms_internal_overlap3d_sym_ok(N, ld, out)) return false;
    return N > 1024;                                              // (B = 8: N = 2048 0.110 -> 0.090 ms, 1536 0.0895 -> 0.080, 1024 even)
}

// K3..K6 of every image + the matrix in one launch (tail_write_kernel)
template <int SRC>
int launch_tail_write(const float* chain_src, const float* write_src, int B, int N, const int32_t* counts, const gnms_params& P, char* ws,
                      const gnms_ws_layout& L, float* prob, int64_t* valid, int64_t* invalid, int32_t* nvalid, int32_t* ninvalid, float* out,
                      int64_t ld, hipStream_t st) {
    int P2 = next_pow2(N);
    if (P2 < 1024) P2 = 1024;
    // (the fused K5 -> K6 hand-off, E <= 4, parks order[] and a copy of r2 behind the key region: 16 bytes per key)
    const size_t llds = leaders_lds_bytes(N), glds = (size_t)P2 * (P2 <= 4096 ? 16 : 8);
    size_t lds = llds > glds ? llds : glds;
    const int tr = kFusedTileRows;
    int staged = (SRC == kFromBoxes && N <= 4096) ? 1 : 0;           // writers_staged_2d: the image's boxes in LDS
    if (staged && lds < (size_t)N * 16) lds = (size_t)N * 16;
    long writers = write_chunk_count(N, B, tr, 0, N);                // persistent writers: at most one per CU
    if (SRC == kFromRecords && sym_writers_in_tail_launch(N, ld, out)) {   // writers_sym_persistent: two LDS macro tiles
        staged = 2;
        if (lds < 2 * gnms_iou3d::kSymTileBytes) lds = 2 * gnms_iou3d::kSymTileBytes;
        writers = (long)gnms_iou3d::sym_tiles_per_image(N) * B;
    }
    const int fast = (fast_tail_enabled() && fast_tail_ok(N, P, 1)) ? 1 : 0;
    if (fast && lds < fast_tail_lds_size(N, P2)) lds = fast_tail_lds_size(N, P2);
    const int cus = device_cu_count();
    if (writers > cus) writers = cus;
    // How many CUs write.  With the packed row body (iou_tile.h) a writer workgroup sustains ~29 GB/s and the stream saturates near
    // 5.8 TB/s from ~200 of them; more writers add nothing to the matrix and slow the chain beside them, whose loads queue behind
    // the stores.  B = 8, N = 4096, same box, ms per step clustered / uniform: 216 writers 0.140 / 0.160, 208 0.139 / 0.158,
    // 200 0.140 / 0.151, 192 0.142 / 0.142, 184 0.144 / 0.144 -- at 24 writers per XCD the chain of the uniform images (1890
    // leaders each) stops being the longer side of the launch, at the price of 1.5 % on clustered ones.  (A plain fill in this
    // geometry: 6.2-6.4 TB/s from 64-128 workgroups, 5.8 from 248: profiles/r03*_store_geometry.jsonl.)
    const int cap = staged == 1 ? (cus * 208) / 256 : 0;
    if (cap > 0 && writers > cap) writers = cap;
    const dim3 grid((unsigned)(B * (leaders_chain_wgs(N, 1) + fast) + writers));
    const bool vec = (ld % 4 == 0) && ((uintptr_t)out % 16 == 0);
    int rc;
    GNMS_DISPATCH_SORT(P2, {
        if (vec) {
            if ((rc = allow_lds(tail_write_kernel<true, E, SRC>, lds))) return rc;
            gnms_launch_prof(kProfMatrixWrite, tail_write_kernel<true, E, SRC>, grid, dim3(1024), lds, st, chain_src, write_src, N, counts, P, ws,
                             L, P2, prob, (long long*)valid, (long long*)invalid, nvalid, ninvalid, B, out, (long)ld, tr, 0, N, staged, fast);
        } else {
            if ((rc = allow_lds(tail_write_kernel<false, E, SRC>, lds))) return rc;
            gnms_launch_prof(kProfMatrixWrite, tail_write_kernel<false, E, SRC>, grid, dim3(1024), lds, st, chain_src, write_src, N, counts, P,
                             ws, L, P2, prob, (long long*)valid, (long long*)invalid, nvalid, ninvalid, B, out, (long)ld, tr, 0, N, staged, fast);
        }
    });
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// A small image's whole forward pass as ONE launch (nms_one_launch.h): masked groups, hard sort, N <= 1024 (one super-block), a 16-byte
// aligned matrix with ld % 4 == 0, and no more workgroups than two rounds of the machine (every one asks for the chain's LDS, so a CU
// holds one).  GNMS_ONE_LAUNCH=0: never (developer / test switch: the three-launch path stays reachable for A/B runs and the tests).
bool one_launch_enabled() {
    static const bool on = [] { const char* e = getenv("GNMS_ONE_LAUNCH"); return !(e && e[0] == '0'); }();
    return on;
}
struct OneLaunchPlan { int kpw, split, grid; };
bool one_launch_plan(int B, int N, const gnms_params& P, OneLaunchPlan* plan) {
    if (!one_launch_enabled() || !fast_tail_enabled() || N > kOneLaunchMaxN || !fast_tail_ok(N, P, 1)) return false;
    const int cus = device_cu_count();
    const int NP = (N + 63) & ~63, NB = NP / 64;
    const int kpw = (NP % 128 == 0 && (long)B * (NP / 32) <= (long)cus / 2) ? 32 : 64;
    int split = 4;                                                // table workgroups: 16 rows each while that leaves the machine half empty
    while (split > 1 && (long)B * NB * split > (long)cus / 2) split >>= 1;
    const long grid = (long)B * (NP / kpw + NB * split + 2);
    if (grid > 2L * cus) return false;
    plan->kpw = kpw; plan->split = split; plan->grid = (int)grid;
    return true;
}
int launch_one_matrix(const float* scores, const float* iou, int B, int N, int64_t ld, const int32_t* counts, const gnms_params& P, char* ws,
                      const gnms_ws_layout& L, float* prob, int64_t* order, int64_t* valid, int64_t* invalid, int32_t* nvalid, int32_t* ninvalid,
                      hipStream_t st, const OneLaunchPlan& plan) {
    const size_t lds = one_launch_lds_size(N, false);
    int rc;
    if ((rc = allow_lds(one_launch_kernel<kFromMatrix>, lds))) return rc;
    // (profile slot of the matrix READ: this launch holds the layer's one pass over the matrix -- bench.py's `roofline_matrix_in`)
    gnms_launch_prof(kProfMatrixRead, one_launch_kernel<kFromMatrix>, dim3((unsigned)plan.grid), dim3(1024), lds, st, scores, iou, N, (long)ld, counts, P, ws, L, prob,
                     (long long*)valid, (long long*)invalid, nvalid, ninvalid, (long long*)order, B, plan.kpw, plan.split);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// keys per sort workgroup with the x-order table: 64.  (Measured: 128 keys a workgroup -- two sort roles + the tables then fit one round of the
// machine, so every table workgroup is resident from the start -- loses to the longer count: B = 8 N = 1024 31.0 against 29.7 us, B = 8 N = 768
// 27.4 / 26.6, B = 4 N = 1024 30.3 / 29.1; 32 keys where that fits half the machine: B = 2 N = 1024 28.0 / 28.2 -- no difference.)
constexpr int kXtKpw = 64;
// which table the one-call entry's one launch builds: 0 none (three launches), 1 rank space (one 64 x 64 task per workgroup, while all tasks
// are at most one round of the machine), 2 sources in x order (beyond that, while the launch's front is at most two rounds)
int one_launch_boxes_mode(int B, int N) {
    const int cus = device_cu_count();
    const int NP = (N + 63) & ~63, NB = NP / 64, nbp = NB * (NB + 1) / 2;
    const int kpw = (NP % 128 == 0 && (long)B * (NP / 32) <= (long)cus / 2) ? 32 : 64;
    if ((long)B * nbp <= (long)cus && nbp <= 13 * 32) return (long)B * (NP / kpw + nbp + 2) <= 3L * cus ? 1 : 0;
    return (long)B * (2 * (NP / kXtKpw) + NB + 2) <= 2L * cus ? 2 : 0;
}

// the one-call entry (gnms_forward_with_iou2d, masked groups): the same with the table from the boxes and the matrix writers behind the chain
int launch_one_boxes(const float* scores, const float* boxes, int B, int N, const int32_t* counts, const gnms_params& P, char* ws,
                     const gnms_ws_layout& L, float* prob, int64_t* order, int64_t* valid, int64_t* invalid, int32_t* nvalid, int32_t* ninvalid,
                     float* out, int64_t ld, hipStream_t st, bool* launched) {
    *launched = false;
    if (!one_launch_enabled() || !fast_tail_enabled() || N > kOneLaunchMaxN || !fast_tail_ok(N, P, 1) || B > kClaimImgs) return GNMS_OK;
    if (!((ld % 4 == 0) && ((uintptr_t)out % 16 == 0))) return GNMS_OK;
    const int cus = device_cu_count();
    const int NP = (N + 63) & ~63, NB = NP / 64, nbp = NB * (NB + 1) / 2;
    int kpw = (NP % 128 == 0 && (long)B * (NP / 32) <= (long)cus / 2) ? 32 : 64;
    // One task (a 64 x 64 block of pair decisions on 16 waves) per table workgroup, and only while all of them are about one round of the
    // machine: in rank space nothing can be culled, and where the tasks queue the three launches (x-sorted, culled bit matrix: 5-7 us) win.
    // Kernel time of the launch against sort + bits + tail_write_kernel, us (profiles/r06g_*): B = 8 N = 256 18.3 / 25.0, B = 16 N = 256
    // 17.5 / 24.9, B = 1 / 2 N = 500 18.1 / 20.5 and 18.2 / 21.9, B = 8 N = 512 22.8 / 23.4, B = 1 N = 1024 23.4 / 25.1 -- and with 2 / 4 / 8
    // tasks per workgroup B = 2 / 4 / 8 at N = 1024: 28.0 / 25.9, 31.2 / 26.7, 37.9 / 33.5 (the one launch loses).
    // Beyond one round: the table with its SOURCES IN X ORDER (one_launch_bits_from_boxes_x, tpw = 0) -- NB culled workgroups a image instead of
    // NB (NB + 1) / 2 unculled ones, for a second sort role (the boxes by x centre) in the launch and the writers behind everything.  Kernel
    // time, x-order table / rank-space table or the three launches, us, same box (profiles/r06x_*): B = 8 N = 1024 31.3 / 35.2 (three launches),
    // B = 4 N = 1024 29.8 / 32.0, B = 16 N = 512 24.1 / 29.0, B = 8 N = 768 27.6 / 32.5, B = 4 N = 768 25.8 / 27.3, B = 2 N = 1024 28.9 / 31.1,
    // B = 8 N = 512 24.4 / 25.9 -- and below a round of tasks the rank-space table wins: B = 1 N = 1024 27.1 / 25.0, B = 4 N = 512 23.2 / 20.2,
    // B = 16 N = 256 21.9 / 18.8, B = 8 N = 256 22.0 / 18.3, B = 1, 2 N = 500 21.8 / 19.4, 22.5 / 19.6.
    const int mode = one_launch_boxes_mode(B, N);
    if (mode == 0) return GNMS_OK;
    const bool xt = mode == 2;
    if (xt) kpw = kXtKpw;
    const int tpw = xt ? 0 : 1;
    const int ntab = nbp;
    const long front = xt ? (long)B * (2 * (NP / kpw) + NB + 2) : (long)B * (NP / kpw + ntab + 2);
    int* claims = nullptr;
    int rc = claim_slot_for(st, &claims);
    if (rc == kNoClaimSlot) return GNMS_OK;
    if (rc) return rc;
    const int ncc = (N + gnms_iou::kWaveCols - 1) / gnms_iou::kWaveCols, nrt = (N + kStagedRows - 1) / kStagedRows;
    long writers = (long)B * ((ncc * nrt + 15) >> 4);
    // (3/8 of the CUs write: at ~29 GB/s per writer workgroup the 32 MB of B = 8, N = 1024 then take as long as sort -> table -> chain
    // beside them on the other 5/8)
    const long cap = xt ? (long)cus / 2 : (long)cus * 3 / 8;                 // (x-order table: the writers are last in the grid and start as the sorts retire)
    if (writers > cap) writers = cap;
    if (writers < 1) writers = 1;
    size_t lds = one_launch_lds_size(N, true);
    if (lds < (size_t)N * 16) lds = (size_t)N * 16;
    if ((rc = allow_lds(one_launch_boxes_kernel, lds))) return rc;
    gnms_launch_prof(kProfMatrixWrite, one_launch_boxes_kernel, dim3((unsigned)(front + writers)), dim3(1024), lds, st, scores, boxes, N, counts, P, ws, L, prob,
                     (long long*)valid, (long long*)invalid, nvalid, ninvalid, (long long*)order, B, kpw, tpw, out, (long)ld, claims, (int)writers);
    GNMS_CHECK_LAUNCH();
    *launched = true;
    return GNMS_OK;
}

int forward_impl(const char* fn, const float* scores, const float* iou, int B, int N, int64_t ld, const int32_t* counts,
                 const gnms_params* params, float* prob, int64_t* order, int64_t* valid, int64_t* invalid, int32_t* nvalid,
                 int32_t* ninvalid, void* workspace, size_t workspace_bytes, void* stream, bool scores_already_sorted,
                 const float* boxes2d = nullptr) {
    // boxes2d: the 2D boxes `iou` was computed from (gnms_forward_with_iou2d), 16-byte aligned, or null: the ungrouped mode then builds
    // its pruned lower-triangular matrix from them instead of reading `iou` back
    int rc = check_common(fn, B, N, ld, params, workspace, workspace_bytes);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return GNMS_OK;
    if (N == 0) {
        if (nvalid) GNMS_CHECK_HIP(hipMemsetAsync(nvalid, 0, sizeof(int32_t) * B, st));
        if (ninvalid) GNMS_CHECK_HIP(hipMemsetAsync(ninvalid, 0, sizeof(int32_t) * B, st));
        return GNMS_OK;
    }
    GNMS_CHECK_ARG(scores && iou && prob, "gnms_forward: null scores/iou/prob");
    const gnms_params P = *params;
    const gnms_ws_layout L = gnms_make_layout(N);
    char* ws = (char*)workspace;
    const int P2 = next_pow2(N);
    const size_t sort_lds = (size_t)P2 * 8;
    const int sort_threads = P2 <= 1024 ? P2 : 1024;

    {   // a small image: sort, threshold bits and chain as one launch
        OneLaunchPlan plan;
        if (!scores_already_sorted && (ld % 4 == 0) && ((uintptr_t)iou % 16 == 0) && one_launch_plan(B, N, P, &plan))
            return launch_one_matrix(scores, iou, B, N, ld, counts, P, ws, L, prob, order, valid, invalid, nvalid, ninvalid, st, plan);
    }
    const bool permute_from_boxes = boxes2d && !P.group_boxes && !P.presorted && !scores_already_sorted;
    // (with the boxes the score sort also leaves them in rank order, rbox; its second role, the boxes by x centre, is not used here)
    if (!scores_already_sorted && (rc = launch_sorts(scores, permute_from_boxes ? boxes2d : nullptr, B, N, counts, ws, L, P2, order, st))) return rc;

    if (P.group_boxes && P.mask_group_boxes && use_tail_kernel(N, matrix_sym_detection(N) ? 2 : 0)) {
        // (with the fast tail the symmetry check is a role of the tail launch and the scan runs on trust beside it: sym 3)
        const int sym = matrix_sym_detection(N) ? ((fast_tail_enabled() && fast_tail_ok(N, P, 2)) ? 3 : 2) : 0;
        if ((rc = launch_bitmask(iou, B, N, ld, counts, P.nms_threshold, ws, L, st, sym == 3 ? 2 : (sym ? 1 : 0)))) return rc;
        return launch_tail<false>(iou, B, N, ld, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, st, sym);
    }
    if (P.group_boxes) {
        if ((rc = run_grouping(iou, B, N, ld, counts, P.nms_threshold, ws, L, st))) return rc;
        GNMS_DISPATCH_SORT(P2, {
            if ((rc = allow_lds(groups_kernel<E, false>, sort_lds))) return rc;
            groups_kernel<E, false><<<B, sort_threads, sort_lds, st>>>(iou, N, (long)ld, counts, P, ws, L, P2);
        });
        GNMS_CHECK_LAUNCH();
        if (!P.mask_group_boxes) {
            const size_t lds = kSolveGroupsLds;
            if ((rc = allow_lds(solve_groups_kernel<false, false>, lds))) return rc;
            solve_groups_kernel<false, false><<<dim3(solve_groups_wgs(B, device_cu_count()), B), 1024, lds, st>>>(iou, N, (long)ld, counts, P, ws, L, nullptr, nullptr);
            GNMS_CHECK_LAUNCH();
        }
    } else {
        float* Ps = reinterpret_cast<float*>(ws + (size_t)B * L.per_image);      // scratch behind the per-image regions
        const size_t plds = (size_t)N * 4;
        if ((rc = allow_lds(ungrouped_permute_kernel, plds))) return rc;
        ungrouped_prepare_kernel<<<dim3(gnms_div_up(N, 1024), B), 1024, 0, st>>>(N, counts, P, ws, L);
        GNMS_CHECK_LAUNCH();
        if (permute_from_boxes) ungrouped_permute_boxes_kernel<<<dim3(gnms_div_up(N, kPermuteRows), B), 256, 0, st>>>(N, counts, P, ws, L, Ps);
        else ungrouped_permute_kernel<<<dim3(N, B), 256, plds, st>>>(iou, N, (long)ld, counts, P, ws, L, Ps);
        GNMS_CHECK_LAUNCH();
        if ((rc = allow_lds(ungrouped_solve_forward_kernel, kUngroupedFwdLds))) return rc;
        ungrouped_solve_forward_kernel<<<dim3(gnms_div_up(N, kUB), B), kUThreads, kUngroupedFwdLds, st>>>(scores, N, counts, P, ws, L, Ps);
        GNMS_CHECK_LAUNCH();
    }
    GNMS_DISPATCH_SORT(P2, {
        if ((rc = allow_lds(finalize_kernel<E>, sort_lds))) return rc;
        finalize_kernel<E><<<B, sort_threads, sort_lds, st>>>(N, counts, P, ws, L, P2, prob, (long long*)valid, (long long*)invalid,
                                                               nvalid, ninvalid);
    });
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

}  // namespace

extern "C" int gnms_forward(const float* scores, const float* iou, int B, int N, int64_t ld, const int32_t* counts,
                            const gnms_params* params, float* prob, int64_t* order, int64_t* valid, int64_t* invalid,
                            int32_t* nvalid, int32_t* ninvalid, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_impl("gnms_forward", scores, iou, B, N, ld, counts, params, prob, order, valid, invalid, nvalid, ninvalid, workspace,
                        workspace_bytes, stream, false);
}

namespace {
// Large images (N > 4096, matrix >= 384 MiB): the matrix write takes longer than the layer behind it and nothing in the masked
// from-boxes layer reads it, so the write goes to a library-owned second stream and the layer's own kernels run beside it on the
// caller's stream.  A fork and a join cost 7-25 us of idle queue each and the layer's latency-bound kernels run 1.3-3x slower
// beside the write, which is why smaller problems stay on one stream (B=8, N=4096: 0.180-0.187 against 0.193 ms, not worth a
// second code path; B=4: 0.152 against 0.138; N=8192, B=1: even; B=2: 0.246 against 0.306).
struct SideStream { hipStream_t s = nullptr; hipEvent_t fork[2] = {nullptr, nullptr}, join = nullptr; };
std::mutex g_side_mu;
bool use_side_stream(int B, int N, int64_t ld) {
    return N > 4096 && (long long)B * N * ld * 4 >= (384ll << 20);
}
// the caller holds g_side_mu
int side_stream(SideStream** out) {
    static std::map<int, SideStream> per_dev;
    int dev = 0;
    GNMS_CHECK_HIP(hipGetDevice(&dev));
    SideStream& S = per_dev[dev];
    if (!S.s) {
        hipStream_t s = nullptr;
        // lowest priority: the write's workgroups fill every CU, and the layer's kernels on the caller's stream (some of them one
        // 128-KiB-LDS workgroup per image) must get the CU slots that free up, not wait for the write to drain
        int least = 0, greatest = 0;
        GNMS_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        GNMS_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least));
        GNMS_CHECK_HIP(hipEventCreateWithFlags(&S.fork[0], hipEventDisableTiming));
        GNMS_CHECK_HIP(hipEventCreateWithFlags(&S.fork[1], hipEventDisableTiming));
        GNMS_CHECK_HIP(hipEventCreateWithFlags(&S.join, hipEventDisableTiming));
        S.s = s;
    }
    *out = &S;
    return GNMS_OK;
}
// everything enqueued on `st` so far happens before what is enqueued on the side stream from now on (`which`: a call that
// forks twice uses a different event each time)
int side_fork(hipStream_t st, hipStream_t* side, int which = 0) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    SideStream* S = nullptr;
    int rc = side_stream(&S);
    if (rc) return rc;
    GNMS_CHECK_HIP(hipEventRecord(S->fork[which], st));
    GNMS_CHECK_HIP(hipStreamWaitEvent(S->s, S->fork[which], 0));
    *side = S->s;
    return GNMS_OK;
}
// the matrix write that gnms_forward_with_iou2d hands to the from-boxes layer for the side stream
struct MatrixWrite { float* out; int64_t ld; bool one_launch; };   // one_launch: inside the chain's launch (tail_write_kernel), else on the side stream
// The write in two launches: rows [0, r) beside the bit-matrix kernel (a VALU-bound kernel of small workgroups, which interleaves
// with the write's), the rest beside the tail.  r as a percentage of N (GNMS_SPLIT_PCT overrides), rounded down to whole 64-row
// tiles.  Measured B=8: 2D N=8192 0.625 / 0.575 / 0.568 / 0.595 ms at 0 / 20 / 40 / 50 %, N=16384 2.10 / 2.03 / 2.08 / 2.08;
// 3D N=8192 0.720 / 0.672 / 0.676 / 0.703, N=16384 2.39 / 2.37 / 2.36 / 2.36 (its write kernel has less VALU to spare).
int split_rows(int N, int pct) { return (int)((long long)N * pct / 100) & ~63; }
// everything enqueued on the side stream so far happens before what is enqueued on `st` from now on
int side_join(hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    SideStream* S = nullptr;
    int rc = side_stream(&S);
    if (rc) return rc;
    GNMS_CHECK_HIP(hipEventRecord(S->join, S->s));
    GNMS_CHECK_HIP(hipStreamWaitEvent(st, S->join, 0));
    return GNMS_OK;
}
// fork(s) and the one join of a call; the destructor joins on an early (error) return, so that the caller's stream always
// covers what was put on the side stream
struct SideScope {
    hipStream_t st;
    bool forked = false;
    explicit SideScope(hipStream_t s) : st(s) {}
    ~SideScope() { if (forked) side_join(st); }
    int fork(hipStream_t* side, int which) { const int rc = side_fork(st, side, which); if (!rc) forked = true; return rc; }
    int join() { forked = false; return side_join(st); }
};
}  // namespace

namespace {
int forward_boxes_impl(const float* boxes, const float* scores, int B, int N, const int32_t* counts, const gnms_params* params,
                       float* prob, int64_t* order, int64_t* valid, int64_t* invalid, int32_t* nvalid, int32_t* ninvalid,
                       void* workspace, size_t workspace_bytes, void* stream, bool scores_already_sorted, const MatrixWrite* mw = nullptr);
}

namespace {
// masked from-boxes layer: K3..K6 of every image and the matrix write as ONE launch (tail_write_kernel).
bool chain_rides_in_write_launch(int B, int N) {
    // (up to N = 1024 the score / x sorts could ride in the IoU launch instead, rounds 1-3's iou2d_sort_kernel; replayed as a HIP graph -- the GPU's
    // own time -- this sequence measures the same or better there too: B = 8, N = 128 / 256 / 512 / 1024: 42.4 / 41.0 / 41.5 / 52.1 us
    // against 38.0 / 36.6 / 40.0 / 47.5)
    return true;
}
}  // namespace

// name, as a kernel trace lists it, of the launch that writes the matrix inside gnms_forward_with_iou2d (dim 2) / _iou3d (dim 3)
// (default parameters, aligned inputs)
extern "C" const char* gnms_profile_write_kernel_name(int dim, int B, int N) {
    if (B <= 0 || N <= 0) return "";
    if (dim == 3 && chain_rides_in_write_launch(B, N) && sym_writers_in_tail_launch(N, N, nullptr)) return "tail_write_kernel";
    if (use_side_stream(B, N, N)) return dim == 3 ? "iou3d_nms_fast_kernel" : (N % 4 == 0 ? "write_staged_kernel" : "iou2d_kernel");
    if (dim == 2 && N <= kOneLaunchMaxN && one_launch_enabled() && fast_tail_enabled() && one_launch_boxes_mode(B, N) != 0) return "one_launch_boxes_kernel";
    if (chain_rides_in_write_launch(B, N) && (dim == 2 || N <= 2048)) return "tail_write_kernel";
    if (dim == 3) return "iou3d_nms_fast_kernel";
    return "iou2d_kernel";
}

// The matrix is an OUTPUT here, so the layer does not have to read it back: with the boxes at hand the grouped modes
// take their threshold bits and the few P[i, head] entries straight from the boxes (bit-identical arithmetic, the
// from-boxes kernels), which replaces the 537 MB read of bitmask_kernel (~100 us at B=8, N=4096) by bitmask_boxes_kernel
// (~25 us, compute bound).  The ungrouped / soft-sorted modes read the matrix they just wrote.
extern "C" int gnms_forward_with_iou2d(const float* boxes, const float* scores, int B, int N, int64_t ld, const int32_t* counts,
                                       const gnms_params* params, float* iou_out, float* prob, int64_t* order, int64_t* valid,
                                       int64_t* invalid, int32_t* nvalid, int32_t* ninvalid, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    int rc = check_common("gnms_forward_with_iou2d", B, N, ld, params, workspace, workspace_bytes);
    if (rc) return rc;
    if (B > 0 && N > 0) GNMS_CHECK_ARG(boxes && scores && iou_out && prob, "gnms_forward_with_iou2d: null pointer");
    const bool from_boxes = params->group_boxes && !params->presorted && ((uintptr_t)boxes % 16 == 0);
    const bool beside = B > 0 && N > 0 && from_boxes && params->mask_group_boxes && use_side_stream(B, N, ld);
    // (unmasked groups, round 4b: K3..K5's group structure rides in the write launch too, up to N = 4096; the per-group solves and K6 follow)
    const bool chain_in_write = B > 0 && N > 0 && from_boxes && (params->mask_group_boxes || N <= 4096) && !beside && chain_rides_in_write_launch(B, N);
    if (chain_in_write) {
        const MatrixWrite mw = {iou_out, ld, true};
        return forward_boxes_impl(boxes, scores, B, N, counts, params, prob, order, valid, invalid, nvalid, ninvalid, workspace,
                                  workspace_bytes, stream, false, &mw);
    }
    if (beside) {
        const MatrixWrite mw = {iou_out, ld, false};
        return forward_boxes_impl(boxes, scores, B, N, counts, params, prob, order, valid, invalid, nvalid, ninvalid, workspace,
                                  workspace_bytes, stream, false, &mw);
    }
    // every other mode: the matrix by gnms_iou2d's own kernels (the persistent writers for a box set with itself), then the layer.
    // (Rounds 1-3 carried the score sort in the last grid slice of a 64 x 256-tile IoU launch here, iou2d_sort_kernel: 135 us at B = 8,
    // N = 4096 where the writers take 92 and the two sort launches 15; removed in round 4b.)
    if (B > 0 && N > 0 && (rc = gnms_iou2d(boxes, boxes, B, N, N, iou_out, ld, stream))) return rc;
    if (from_boxes)
        return forward_boxes_impl(boxes, scores, B, N, counts, params, prob, order, valid, invalid, nvalid, ninvalid, workspace,
                                  workspace_bytes, stream, false);
    return forward_impl("gnms_forward_with_iou2d", scores, iou_out, B, N, ld, counts, params, prob, order, valid, invalid, nvalid,
                        ninvalid, workspace, workspace_bytes, stream, false, ((uintptr_t)boxes % 16 == 0) ? boxes : nullptr);
}

// defined in iou_kernels.hip
int gnms_internal_records_from_params(const float* params, long count, float* rec, hipStream_t st);
int gnms_internal_records_for_layer(const float* params, int B, int N, float* rec, char* ws, const gnms_ws_layout& L, float* xkeys, hipStream_t st);
int gnms_internal_nms_overlap3d(const float* rec, int B, int N, float* out, int64_t ld, hipStream_t st, float thr, int row0 = 0, int row_end = 0x7fffffff);


namespace {
// everything of gnms_forward_with_iou3d that uses the temporary `rec` ([B][N] records, then [B][N] pseudo boxes for the x sort).
// Masked hard-sorted groups: the whole layer runs from the records (threshold bits AND the O(N) single overlaps, same arithmetic
// as the matrix kernel), so nothing waits for the matrix; large images write it on the side stream beside the one-launch tail.
int forward_with_iou3d_on(float* rec, const float* params3d, const float* scores, int B, int N, int64_t ld, const int32_t* counts,
                          const gnms_params& P, float* iou_out, float* prob, int64_t* order, int64_t* valid, int64_t* invalid,
                          int32_t* nvalid, int32_t* ninvalid, char* ws, const gnms_ws_layout& L, hipStream_t st) {
    const bool from_rec = P.group_boxes && P.mask_group_boxes && !P.presorted;
    int rc;
    if (!from_rec) {
        if ((rc = gnms_internal_records_from_params(params3d, (long)B * N, rec, st))) return rc;
        return gnms_internal_nms_overlap3d(rec, B, N, iou_out, ld, st, P.nms_threshold);
    }
    float* xkeys = rec + (size_t)B * N * gnms_iou3d::kRec;         // [B][N] pseudo boxes; later the rank-ordered records
    if ((rc = gnms_internal_records_for_layer(params3d, B, N, rec, ws, L, xkeys, st))) return rc;
    // 1024 < N, no side stream: K3..K6 ride in the launch of the SYMMETRIC writers (tail_write_kernel, writers_sym_persistent) behind
    // the from-records bit-matrix kernel
    const bool culled_bits = P.nms_threshold >= 0.01f && P.nms_threshold < INFINITY;
    const bool sym_tail = culled_bits && chain_rides_in_write_launch(B, N) && sym_writers_in_tail_launch(N, ld, iou_out);
    const bool beside = !sym_tail && use_side_stream(B, N, ld);
    const int sym = (P.nms_threshold >= 0.01f && P.nms_threshold < INFINITY) ? 1 : 0;   // the culled kernel writes full symmetric rows of W
    // K3..K6 inside the write launch like the 2D entry -- up to N = 2048 only: the 3D writers are VALU-bound (23 slots per pair) and
    // at the one workgroup per CU that launch runs at they lose more than the overlap buys (B = 8, N = 4096: launch 166 us against a
    // 107-us write + 55-us chain, step 0.264 against 0.248 ms; N = 2048: 0.117 against 0.163 ms)
    const bool chain_in_write = !beside && sym && (N <= 2048 || sym_tail) && chain_rides_in_write_launch(B, N);
    if (!beside && !chain_in_write && (rc = gnms_internal_nms_overlap3d(rec, B, N, iou_out, ld, st, P.nms_threshold))) return rc;
    const int P2 = next_pow2(N);
    // + the cuboids in the column order of the bit-matrix kernel: (z band, x centre).  Bands: so that a slot of 64 consecutive columns is
    // about as deep in z as it is wide in x (a handful of bands of >= 512 cuboids each; one band up to N = 1024)
    // (B = 8 uniform cuboids, bit-matrix kernel: N = 4096 52 / 47 / 46 / 46.5 us with 1 / 4 / 8 / 12 bands, N = 16384 494 / 409 / 407 with 1 / 8 / 15)
    const int bands = std::max(1, std::min(8, N / 512));
    if ((rc = launch_sorts(scores, xkeys, B, N, counts, ws, L, P2, order, st, bands))) return rc;
    SideScope scope(st);
    // the part of the write that runs beside the bit-matrix kernel: rows [0, r1) of the all-pairs kernel
    const int r1 = beside ? split_rows(N, 20) : 0;
    if (r1 > 0) {                                                 // first part of the write beside the bit-matrix kernel
        hipStream_t side = nullptr;
        if ((rc = scope.fork(&side, 0))) return rc;
        if ((rc = gnms_internal_nms_overlap3d(rec, B, N, iou_out, ld, side, P.nms_threshold, 0, r1))) return rc;
    }
    if (!sym) {
        // no culling possible below that threshold: the triangular tile set does half the pairs of the square one
        bitmask_rec3d_kernel<<<dim3(gnms_div_up(tri_tile_count(L.NB), 4), 1, B), 256, 0, st>>>(N, counts, P.nms_threshold, ws, L);
    } else {
        // row groups of 4 blocks where there are plenty of tiles; from N > 4096 the row groups are dealt to the XCDs (see the kernel)
        const long long tiles = (long long)B * L.NB * ((N + 255) / 256);
        const int kbw = tiles >= 32768 ? 4 : 1;                       // (N = 4096: 46 / 48 / 59 us with 1 / 2 / 4; N = 16384: 338 with 2, 325 with 4)
        const int pinned = N > 4096 ? 1 : 0;
        const int nkbg = gnms_div_up(L.NB, kbw), nchunk = (N + 255) / 256;
        const unsigned gx = pinned ? (unsigned)(gnms_div_up(nkbg, 8) * gnms_div_up(nchunk, 4) * 8) : (unsigned)gnms_div_up(nkbg * nchunk, 4);
        if (kbw >= 4)
            bitmask_rec3d_culled_kernel<4><<<dim3(gx, 1, B), 256, 0, st>>>(N, counts, P.nms_threshold, ws, L, pinned);
        else
            bitmask_rec3d_culled_kernel<1><<<dim3(gx, 1, B), 256, 0, st>>>(N, counts, P.nms_threshold, ws, L, pinned);
    }
    GNMS_CHECK_LAUNCH();
    if (chain_in_write)                                           // K3..K6 and the matrix in one launch, like the 2D entry
        return launch_tail_write<kFromRecords>(nullptr, rec, B, N, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, iou_out, ld, st);
    if (beside) {                                                 // see forward_boxes_impl for why the fork sits exactly here
        hipStream_t side = nullptr;
        if ((rc = scope.fork(&side, 1))) return rc;
        if ((rc = launch_tail<kFromRecords>(nullptr, B, N, ld, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, st, sym, kBesideChainWGs))) return rc;
        if ((rc = gnms_internal_nms_overlap3d(rec, B, N, iou_out, ld, side, P.nms_threshold, r1, N))) return rc;
        return scope.join();
    }
    if (use_tail_kernel(N, sym)) return launch_tail<kFromRecords>(nullptr, B, N, ld, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, st, sym);
    const size_t llds = leaders_lds_bytes(N);
    if ((rc = allow_lds(leaders_kernel, llds))) return rc;
    { const int spw = leaders_chain_wgs(N, sym); leaders_kernel<<<B * spw, 1024, llds, st>>>(N, counts, ws, L, sym, B, spw); }
    GNMS_CHECK_LAUNCH();
    attribute_kernel<kFromRecords><<<dim3(L.NB, B), 64, 0, st>>>(nullptr, (long)ld, N, counts, P.nms_threshold, ws, L, sym);
    GNMS_CHECK_LAUNCH();
    const size_t sort_lds = (size_t)P2 * 8;
    const int sort_threads = P2 <= 1024 ? P2 : 1024;
    GNMS_DISPATCH_SORT(P2, {
        if ((rc = allow_lds(groups_kernel<E, kFromRecords>, sort_lds))) return rc;
        groups_kernel<E, kFromRecords><<<B, sort_threads, sort_lds, st>>>(nullptr, N, (long)ld, counts, P, ws, L, P2);
    });
    GNMS_CHECK_LAUNCH();
    GNMS_DISPATCH_SORT(P2, {
        if ((rc = allow_lds(finalize_kernel<E>, sort_lds))) return rc;
        finalize_kernel<E><<<B, sort_threads, sort_lds, st>>>(N, counts, P, ws, L, P2, prob, (long long*)valid, (long long*)invalid, nvalid,
                                                               ninvalid);
    });
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
}  // namespace

// The 3D analogue of gnms_forward_with_iou2d: cuboid parameters -> the NMS overlap matrix 0.5*(1+GIoU3D) (an output, written by
// iou3d_nms_fast_kernel) -> the layer.  The grouped hard-sort modes take their threshold bits from the records with the same
// instruction sequence that wrote the matrix (bitmask_rec3d_kernel) instead of reading 4 N^2 bytes back; the O(N) overlaps the
// group kernels need are gathered from the matrix.
extern "C" int gnms_forward_with_iou3d(const float* params3d, const float* scores, int B, int N, int64_t ld, const int32_t* counts,
                                       const gnms_params* params, float* iou_out, float* prob, int64_t* order, int64_t* valid,
                                       int64_t* invalid, int32_t* nvalid, int32_t* ninvalid, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    int rc = check_common("gnms_forward_with_iou3d", B, N, ld, params, workspace, workspace_bytes);
    if (rc) return rc;
    if (B == 0 || N == 0)
        return forward_impl("gnms_forward_with_iou3d", scores, iou_out, B, N, ld, counts, params, prob, order, valid, invalid, nvalid, ninvalid,
                            workspace, workspace_bytes, stream, false);
    GNMS_CHECK_ARG(params3d && scores && iou_out && prob, "gnms_forward_with_iou3d: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const gnms_params P = *params;
    const gnms_ws_layout L = gnms_make_layout(N);
    char* ws = (char*)workspace;
    // records of the whole batch in one stream-ordered temporary (the overlap kernel wants them contiguous); the layer's copy goes
    // into the per-image workspace regions
    float* rec = nullptr;
    GNMS_CHECK_HIP(hipMallocAsync((void**)&rec, (size_t)B * N * (2 * gnms_iou3d::kRec) * sizeof(float), st));
    rc = forward_with_iou3d_on(rec, params3d, scores, B, N, ld, counts, P, iou_out, prob, order, valid, invalid, nvalid, ninvalid, ws, L, st);
    const hipError_t fe = hipFreeAsync(rec, st);                  // after the side stream, if any, has joined `st`
    if (rc) return rc;
    if (fe != hipSuccess) { gnms_set_error("hipFreeAsync failed: %s", hipGetErrorString(fe)); return GNMS_ERR_HIP; }
    if (!(P.group_boxes && P.mask_group_boxes && !P.presorted))
        return forward_impl("gnms_forward_with_iou3d", scores, iou_out, B, N, ld, counts, params, prob, order, valid, invalid, nvalid, ninvalid,
                            workspace, workspace_bytes, stream, false);
    return GNMS_OK;
}

extern "C" int gnms_backward(const float* grad_prob, const float* scores, const float* iou, int B, int N, int64_t ld,
                             const int32_t* counts, const gnms_params* params, float* grad_scores, float* grad_iou,
                             void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common("gnms_backward", B, N, ld, params, workspace, workspace_bytes);
    if (rc) return rc;
    if (B == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(grad_prob && scores && iou && grad_scores, "gnms_backward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const gnms_params P = *params;
    const gnms_ws_layout L = gnms_make_layout(N);
    char* ws = (char*)workspace;
    dim3 ge(gnms_div_up(N, 256), B);
    const bool fused_gx = P.group_boxes && P.mask_group_boxes && !P.return_sorted_prob && !P.presorted;   // the default path
    if (!fused_gx) {
        bwd_gx_kernel<<<ge, 256, 0, st>>>(grad_prob, N, counts, P, ws, L);
        GNMS_CHECK_LAUNCH();
    }
    if (grad_iou) GNMS_CHECK_HIP(hipMemsetAsync(grad_iou, 0, sizeof(float) * (size_t)B * N * ld, st));
    if (P.group_boxes && P.mask_group_boxes) {
        const int head_blocks = N >= 2048 ? 128 : gnms_div_up(N, 16);
        if (fused_gx) {
            bwd_masked_fused_kernel<<<dim3(ge.x + head_blocks, B), 256, 0, st>>>(grad_prob, N, counts, P, ws, L, grad_scores, (int)ge.x);
            GNMS_CHECK_LAUNCH();
        } else {
            bwd_masked_kernel<<<ge, 256, 0, st>>>(N, counts, P, ws, L, grad_scores);
            GNMS_CHECK_LAUNCH();
            bwd_masked_heads_kernel<<<dim3(head_blocks, B), 256, 0, st>>>(N, P, ws, L, grad_scores);
            GNMS_CHECK_LAUNCH();
        }
        if (grad_iou) {
            bwd_masked_iou_kernel<<<ge, 256, 0, st>>>(iou, N, (long)ld, counts, P, ws, L, grad_iou);
            GNMS_CHECK_LAUNCH();
        }
    } else if (P.group_boxes) {
        const size_t lds = kSolveGroupsLds;
        if ((rc = allow_lds(solve_groups_kernel<true, false>, lds))) return rc;
        solve_groups_kernel<true, false><<<dim3(solve_groups_wgs(B, device_cu_count()), B), 1024, lds, st>>>(iou, N, (long)ld, counts, P, ws, L, grad_scores, grad_iou);
        GNMS_CHECK_LAUNCH();
    } else {
        const float* Ps = reinterpret_cast<const float*>(ws + (size_t)B * L.per_image);   // written by the forward pass
        ungrouped_backward_prepare_kernel<<<dim3(gnms_div_up(N, 1024), B), 1024, 0, st>>>(N, ws, L);
        GNMS_CHECK_LAUNCH();
        if ((rc = allow_lds(ungrouped_solve_backward_kernel, kUngroupedBwdLds))) return rc;
        ungrouped_solve_backward_kernel<<<dim3(gnms_div_up(N, kUB), B), kUThreads, kUngroupedBwdLds, st>>>(N, counts, P, ws, L, Ps, grad_scores);
        GNMS_CHECK_LAUNCH();
        if (grad_iou) {
            ungrouped_grad_iou_kernel<<<dim3(gnms_div_up(N, 256), N, B), 256, 0, st>>>(iou, N, (long)ld, counts, P, ws, L, grad_iou);
            GNMS_CHECK_LAUNCH();
        }
    }
    return GNMS_OK;
}

// ------------------------------------------------------------------------------------------------
// from-boxes path: same layer, the N x N matrix never materialised (grouped modes)
// ------------------------------------------------------------------------------------------------
namespace {
int forward_boxes_impl(const float* boxes, const float* scores, int B, int N, const int32_t* counts, const gnms_params* params,
                       float* prob, int64_t* order, int64_t* valid, int64_t* invalid, int32_t* nvalid, int32_t* ninvalid,
                       void* workspace, size_t workspace_bytes, void* stream, bool scores_already_sorted, const MatrixWrite* mw) {
    int rc = check_common("gnms_forward_from_boxes", B, N, N, params, workspace, workspace_bytes);
    if (rc) return rc;
    if (!params->group_boxes || params->presorted) {
        gnms_set_error("gnms_forward_from_boxes: only the grouped, hard-sorted modes run without the matrix");
        return GNMS_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return GNMS_OK;
    if (N == 0) {
        if (nvalid) GNMS_CHECK_HIP(hipMemsetAsync(nvalid, 0, sizeof(int32_t) * B, st));
        if (ninvalid) GNMS_CHECK_HIP(hipMemsetAsync(ninvalid, 0, sizeof(int32_t) * B, st));
        return GNMS_OK;
    }
    GNMS_CHECK_ARG(boxes && scores && prob, "gnms_forward_from_boxes: null boxes/scores/prob");
    GNMS_CHECK_ARG((uintptr_t)boxes % 16 == 0, "gnms_forward_from_boxes: boxes must be 16-byte aligned");
    const gnms_params P = *params;
    const gnms_ws_layout L = gnms_make_layout(N);
    char* ws = (char*)workspace;
    const int P2 = next_pow2(N);
    const size_t sort_lds = (size_t)P2 * 8;
    const int sort_threads = P2 <= 1024 ? P2 : 1024;
    // Large images: the matrix write is ONE persistent launch (write_staged_kernel) on the side stream that leaves B CUs without a
    // writer workgroup; forked behind the bit-matrix kernel, so that what runs beside it on the caller's stream is the one-workgroup-
    // per-image tail, which finds those CUs free (B = 8, N = 16384: write 1.60 ms = 5.4 TB/s, the tail 1.03 ms beside it, step 2.04 ms;
    // with gnms_iou2d's kernel in two launches beside bit matrix and tail: 2.13).  Forked in front of the bit-matrix kernel the two
    // VALU-heavy kernels share the SIMDs and the sum stays the same (write 1.97 ms, step 2.01); forked in front of the sorts as well,
    // those crawl (step 2.35).
    const bool persistent_write = mw && !mw->one_launch && (mw->ld % 4 == 0) && ((uintptr_t)mw->out % 16 == 0) && (N % 4 == 0);
    if (mw && mw->one_launch && !scores_already_sorted) {            // a small image: sort, table, chain and the matrix write as ONE launch
        bool launched = false;
        if ((rc = launch_one_boxes(scores, boxes, B, N, counts, P, ws, L, prob, order, valid, invalid, nvalid, ninvalid, mw->out, mw->ld, st, &launched))) return rc;
        if (launched) return GNMS_OK;
    }
    if (!scores_already_sorted && (rc = launch_sorts(scores, boxes, B, N, counts, ws, L, P2, order, st))) return rc;
    if (persistent_write) {
        SideScope whole(st);
        hipStream_t side = nullptr;
        if ((rc = launch_bitmask_boxes(boxes, B, N, counts, P.nms_threshold, ws, L, st))) return rc;
        if ((rc = whole.fork(&side, 0))) return rc;
        // (the scan on at most kBesideChainWGs workgroups per image: their CUs are the ones the persistent write leaves free)
        if ((rc = launch_tail<true>(boxes, B, N, N, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, st, 1, kBesideChainWGs))) return rc;
        if ((rc = launch_write_staged(boxes, boxes, B, N, N, mw->out, mw->ld, B * leaders_chain_wgs(N, 1, kBesideChainWGs), side))) return rc;
        return whole.join();
    }
    if (mw && mw->one_launch) {
        if ((rc = launch_bitmask_boxes(boxes, B, N, counts, P.nms_threshold, ws, L, st))) return rc;
        if ((rc = launch_tail_write<kFromBoxes>(boxes, boxes, B, N, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, mw->out, mw->ld, st))) return rc;
        if (!P.mask_group_boxes) {                                    // unmasked groups: the chain stopped behind K5's group structure
            const size_t lds = kSolveGroupsLds;
            if ((rc = allow_lds(solve_groups_kernel<false, true>, lds))) return rc;
            solve_groups_kernel<false, true><<<dim3(solve_groups_wgs(B, device_cu_count()), B), 1024, lds, st>>>(boxes, N, (long)N, counts, P, ws, L, nullptr, nullptr);
            GNMS_CHECK_LAUNCH();
            GNMS_DISPATCH_SORT(P2, {
                if ((rc = allow_lds(finalize_kernel<E>, sort_lds))) return rc;
                finalize_kernel<E><<<B, sort_threads, sort_lds, st>>>(N, counts, P, ws, L, P2, prob, (long long*)valid, (long long*)invalid, nvalid, ninvalid);
            });
            GNMS_CHECK_LAUNCH();
        }
        return GNMS_OK;
    }
    SideScope beside(st);
    const int r1 = mw ? split_rows(N, 20) : 0;
    if (r1 > 0) {                        // first part of the write beside the bit-matrix kernel (small workgroups: they interleave)
        hipStream_t side = nullptr;
        if ((rc = beside.fork(&side, 0))) return rc;
        if ((rc = gnms_internal_iou2d_rows(boxes, B, N, mw->out, mw->ld, side, 0, r1))) return rc;
    }
    if ((rc = launch_bitmask_boxes(boxes, B, N, counts, P.nms_threshold, ws, L, st))) return rc;
    if (mw) {
        // Large images: the rest of the layer is one workgroup per image (K3..K6 in one launch) and the matrix write runs beside
        // it on the side stream.  The write is forked HERE and not earlier because its workgroups take every free wave slot: a
        // 16-wave, 128-KiB-LDS workgroup that becomes ready while the write runs is not placed before the write drains (measured:
        // sort_merge_kernel waited 1.85 ms).  Forked here, the tail's B workgroups are resident before the side stream has seen
        // the event (same-queue successor ~2 us, cross-queue event ~25 us), and then keep their CUs until the layer is done.
        hipStream_t side = nullptr;
        if ((rc = beside.fork(&side, 1))) return rc;
        if ((rc = launch_tail<true>(boxes, B, N, N, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, st, 1, kBesideChainWGs))) return rc;
        if ((rc = gnms_internal_iou2d_rows(boxes, B, N, mw->out, mw->ld, side, r1, N))) return rc;
        return beside.join();
    }
    if (P.mask_group_boxes && use_tail_kernel(N, 1))
        return launch_tail<true>(boxes, B, N, N, counts, P, ws, L, prob, valid, invalid, nvalid, ninvalid, st, 1);
    const size_t llds = leaders_lds_bytes(N);
    if ((rc = allow_lds(leaders_kernel, llds))) return rc;
    { const int spw = leaders_chain_wgs(N, 1); leaders_kernel<<<B * spw, 1024, llds, st>>>(N, counts, ws, L, 1, B, spw); }   // bitmask_boxes_kernel wrote full symmetric rows
    GNMS_CHECK_LAUNCH();
    attribute_kernel<true><<<dim3(L.NB, B), 64, 0, st>>>(boxes, (long)N, N, counts, P.nms_threshold, ws, L, 1);
    GNMS_CHECK_LAUNCH();
    GNMS_DISPATCH_SORT(P2, {
        if ((rc = allow_lds(groups_kernel<E, true>, sort_lds))) return rc;
        groups_kernel<E, true><<<B, sort_threads, sort_lds, st>>>(boxes, N, (long)N, counts, P, ws, L, P2);
    });
    GNMS_CHECK_LAUNCH();
    if (!P.mask_group_boxes) {
        const size_t lds = kSolveGroupsLds;
        if ((rc = allow_lds(solve_groups_kernel<false, true>, lds))) return rc;
        solve_groups_kernel<false, true><<<dim3(solve_groups_wgs(B, device_cu_count()), B), 1024, lds, st>>>(boxes, N, (long)N, counts, P, ws, L, nullptr, nullptr);
        GNMS_CHECK_LAUNCH();
    }
    GNMS_DISPATCH_SORT(P2, {
        if ((rc = allow_lds(finalize_kernel<E>, sort_lds))) return rc;
        finalize_kernel<E><<<B, sort_threads, sort_lds, st>>>(N, counts, P, ws, L, P2, prob, (long long*)valid, (long long*)invalid,
                                                               nvalid, ninvalid);
    });
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
}  // namespace

extern "C" int gnms_forward_from_boxes(const float* boxes, const float* scores, int B, int N, const int32_t* counts,
                                       const gnms_params* params, float* prob, int64_t* order, int64_t* valid, int64_t* invalid,
                                       int32_t* nvalid, int32_t* ninvalid, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_boxes_impl(boxes, scores, B, N, counts, params, prob, order, valid, invalid, nvalid, ninvalid, workspace,
                              workspace_bytes, stream, false);
}

extern "C" int gnms_backward_from_boxes(const float* grad_prob, const float* boxes, const float* scores, int B, int N,
                                        const int32_t* counts, const gnms_params* params, float* grad_scores, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    int rc = check_common("gnms_backward_from_boxes", B, N, N, params, workspace, workspace_bytes);
    if (rc) return rc;
    if (!params->group_boxes || params->presorted) {
        gnms_set_error("gnms_backward_from_boxes: only the grouped, hard-sorted modes run without the matrix");
        return GNMS_ERR_UNSUPPORTED;
    }
    if (B == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(grad_prob && boxes && scores && grad_scores, "gnms_backward_from_boxes: null pointer");
    if (params->mask_group_boxes)      // the masked backward never touches the overlaps
        return gnms_backward(grad_prob, scores, boxes, B, N, N, counts, params, grad_scores, nullptr, workspace, workspace_bytes, stream);
    hipStream_t st = (hipStream_t)stream;
    const gnms_params P = *params;
    const gnms_ws_layout L = gnms_make_layout(N);
    char* ws = (char*)workspace;
    bwd_gx_kernel<<<dim3(gnms_div_up(N, 256), B), 256, 0, st>>>(grad_prob, N, counts, P, ws, L);
    GNMS_CHECK_LAUNCH();
    const size_t lds = kSolveGroupsLds;
    if ((rc = allow_lds(solve_groups_kernel<true, true>, lds))) return rc;
    solve_groups_kernel<true, true><<<dim3(solve_groups_wgs(B, device_cu_count()), B), 1024, lds, st>>>(boxes, N, (long)N, counts, P, ws, L, grad_scores, nullptr);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// Profiling hook: re-runs ONLY the threshold bit-matrix kernel (K2, the one full read of the matrix) on a
// workspace that a previous gnms_forward filled (it needs `order`).  bench.py times it for the roofline line.
extern "C" int gnms_profile_bitmask(const float* iou, int B, int N, int64_t ld, const int32_t* counts, float nms_threshold,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    gnms_params P;
    gnms_default_params(&P);
    int rc = check_common("gnms_profile_bitmask", B, N, ld, &P, workspace, workspace_bytes);
    if (rc) return rc;
    if (B == 0 || N == 0) return GNMS_OK;
    const gnms_ws_layout L = gnms_make_layout(N);
    return launch_bitmask(iou, B, N, ld, counts, nms_threshold, (char*)workspace, L, (hipStream_t)stream);
}

extern "C" int gnms_profile_bitmask_boxes(const float* boxes, int B, int N, const int32_t* counts, float nms_threshold, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    gnms_params P;
    gnms_default_params(&P);
    int rc = check_common("gnms_profile_bitmask_boxes", B, N, N, &P, workspace, workspace_bytes);
    if (rc) return rc;
    if (B == 0 || N == 0) return GNMS_OK;
    const gnms_ws_layout L = gnms_make_layout(N);
    return launch_bitmask_boxes(boxes, B, N, counts, nms_threshold, (char*)workspace, L, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// get_groups as a stand-alone entry (lib/groomed_nms.py:208-270)
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void export_groups_kernel(int N, char* ws, gnms_ws_layout L, int* __restrict__ group_of, int* __restrict__ pos_in_group,
                                     int* __restrict__ ngroups) {
    // group id = index of the group's leader among the leaders in creation order, minus the empty groups before it
    using namespace gnms;
    ImgPtrs I = img_ptrs(ws, L, 0);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) *ngroups = I.misc[0];
    if (k >= N) return;
    const int c = I.order[k];
    const int h = I.head[k];
    int gid = -1;
    if (h >= 0) {
        const int lr = I.rem[k];                              // the group's leader
        gid = I.leadpfx[lr >> 6] + __builtin_popcountll(I.leadw[lr >> 6] & ((1ull << (lr & 63)) - 1ull));
    }
    group_of[c] = gid;
    pos_in_group[c] = I.gpos[k];
}
}  // namespace

extern "C" int gnms_get_groups(const float* scores, const float* iou, int N, int64_t ld, float group_threshold, int group_size,
                               int32_t* group_of, int32_t* pos_in_group, int32_t* ngroups_out, void* workspace,
                               size_t workspace_bytes, void* stream) {
    gnms_params P;
    gnms_default_params(&P);
    P.nms_threshold = group_threshold;
    P.group_size = group_size;
    P.mask_group_boxes = 0;   // grouping only: skip the fused rescoring
    int rc = check_common("gnms_get_groups", 1, N, ld, &P, workspace, workspace_bytes);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    GNMS_CHECK_ARG(ngroups_out != nullptr, "gnms_get_groups: ngroups_out is NULL");
    if (N == 0) { GNMS_CHECK_HIP(hipMemsetAsync(ngroups_out, 0, sizeof(int32_t), st)); return GNMS_OK; }
    GNMS_CHECK_ARG(scores && iou && group_of && pos_in_group, "gnms_get_groups: null pointer");
    const gnms_ws_layout L = gnms_make_layout(N);
    char* ws = (char*)workspace;
    const int P2 = next_pow2(N);
    const size_t sort_lds = (size_t)P2 * 8;
    const int sort_threads = P2 <= 1024 ? P2 : 1024;
    GNMS_DISPATCH_SORT(P2, {
        if ((rc = allow_lds(sort_scores_kernel<E>, sort_lds))) return rc;
        sort_scores_kernel<E><<<1, sort_threads, sort_lds, st>>>(scores, N, nullptr, ws, L, P2, nullptr, nullptr, 0);
    });
    GNMS_CHECK_LAUNCH();
    if ((rc = run_grouping(iou, 1, N, ld, nullptr, group_threshold, ws, L, st))) return rc;
    GNMS_DISPATCH_SORT(P2, {
        if ((rc = allow_lds(groups_kernel<E, false>, sort_lds))) return rc;
        groups_kernel<E, false><<<1, sort_threads, sort_lds, st>>>(iou, N, (long)ld, nullptr, P, ws, L, P2);
    });
    GNMS_CHECK_LAUNCH();
    export_groups_kernel<<<gnms_div_up(N, 256), 256, 0, st>>>(N, ws, L, group_of, pos_in_group, ngroups_out);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

// ------------------------------------------------------------------------------------------------
// pruning_function, elementwise (lib/groomed_nms.py:167-189)
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void prune_kernel(const float* __restrict__ x, long long count, float thr, float temp, int method, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
        out[i] = gnms_prune(x[i], thr, temp, method);
}
}  // namespace

namespace {
__global__ void prune_grad_kernel(const float* __restrict__ x, const float* __restrict__ g, long long count, float thr, float temp, int method,
                                  float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
        out[i] = g[i] * gnms_prune_grad(x[i], thr, temp, method);
}
}  // namespace

extern "C" int gnms_pruning_function_backward(const float* iou, const float* grad_out, int64_t count, float nms_threshold, float temperature,
                                              int pruning_method, float* grad_iou, void* stream) {
    if (pruning_method < 0 || pruning_method > 2) {
        gnms_set_error("gnms_pruning_function_backward: Pruning method not implemented! (pruning_method=%d)", pruning_method);
        return GNMS_ERR_UNSUPPORTED;
    }
    GNMS_CHECK_ARG(count >= 0, "gnms_pruning_function_backward: negative count");
    if (count == 0) return GNMS_OK;
    GNMS_CHECK_ARG(iou && grad_out && grad_iou, "gnms_pruning_function_backward: null pointer");
    long long blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    prune_grad_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(iou, grad_out, (long long)count, nms_threshold, temperature,
                                                                         pruning_method, grad_iou);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

extern "C" int gnms_pruning_function(const float* iou, int64_t count, float nms_threshold, float temperature, int pruning_method,
                                     float* out, void* stream) {
    if (pruning_method < 0 || pruning_method > 2) {
        gnms_set_error("gnms_pruning_function: Pruning method not implemented! (pruning_method=%d)", pruning_method);
        return GNMS_ERR_UNSUPPORTED;
    }
    GNMS_CHECK_ARG(count >= 0, "gnms_pruning_function: negative count");
    if (count == 0) return GNMS_OK;
    GNMS_CHECK_ARG(iou && out, "gnms_pruning_function: null pointer");
    long long blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    prune_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(iou, (long long)count, nms_threshold, temperature, pruning_method, out);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}
