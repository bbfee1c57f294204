// tools/bw_variants.hip -- bandwidth experiments for the two streaming kernels (developer tool, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;

template <typename F> float time_us(F f, int reps = 30) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1000.f / reps;
}

// plain streaming read (sum) and write (fill) and copy baselines
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ in, size_t n4, float* out) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *out = acc;
}
__global__ __launch_bounds__(256) void write_kernel(float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// bitmask variants: RB rows per load batch, NT nontemporal loads, SCATTER 0 none(coalesced col store) 1 scatter via perm
template <int RB, bool NT, int SCATTER, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bitmask_v(const float* __restrict__ iou, int N, const int* __restrict__ order, const int* __restrict__ rankof,
                                                       float thr, u64* __restrict__ W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, kb = blockIdx.y;
    const int k0 = kb * 64;
    const int c0 = (blockIdx.x * WAVES + wave) * 256;
    if (c0 >= N) return;
    const float* m = iou + (size_t)b * N * N;
    const int myrow = order[(size_t)b * N + k0 + lane];
    const int col0 = c0 + 4 * lane;
    unsigned lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int rb = 0; rb < 64; rb += RB) {
        float4 v[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int row = __builtin_amdgcn_readlane(myrow, rb + u);
            const float4* p = reinterpret_cast<const float4*>(m + (size_t)row * N + col0);
            if (NT) { v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y); v[u].z = __builtin_nontemporal_load(&p->z); v[u].w = __builtin_nontemporal_load(&p->w); }
            else v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = rb + u;
            const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { const bool nl = !(vv[j] <= thr); if (r < 32) lo[j] |= nl ? (1u << r) : 0u; else hi[j] |= nl ? (1u << (r - 32)) : 0u; }
        }
    }
    u64* Wk = W + ((size_t)b * (N / 64) + kb) * N;
    if (SCATTER == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) Wk[col0 + j] = ((u64)hi[j] << 32) | lo[j];
    } else {
        const int4 t = *reinterpret_cast<const int4*>(rankof + (size_t)b * N + col0);
        const int rk[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) Wk[rk[j]] = ((u64)hi[j] << 32) | lo[j];
    }
}

// rolled variant: RB loads per trip, real loops (no 64-load hoisting) -> few VGPRs, high occupancy
template <int RB, bool NT, int SCATTER, int WAVES, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void bitmask_r(const float* __restrict__ iou, int N, const int* __restrict__ order, const int* __restrict__ rankof,
                                                       float thr, u64* __restrict__ W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, kb = blockIdx.y;
    const int k0 = kb * 64;
    const int c0 = (blockIdx.x * WAVES + wave) * 256;
    if (c0 >= N) return;
    const float* m = iou + (size_t)b * N * N;
    const int myrow = order[(size_t)b * N + k0 + lane];
    const int col0 = c0 + 4 * lane;
    unsigned wd[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll 1
        for (int rb = 0; rb < 32; rb += RB) {
            float4 v[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int row = __builtin_amdgcn_readlane(myrow, half * 32 + rb + u);
                const float4* p = reinterpret_cast<const float4*>(m + (size_t)row * N + col0);
                if (NT) { v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y); v[u].z = __builtin_nontemporal_load(&p->z); v[u].w = __builtin_nontemporal_load(&p->w); }
                else v[u] = *p;
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const unsigned bit = 1u << (rb + u);
                const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) wd[half][j] |= !(vv[j] <= thr) ? bit : 0u;
            }
        }
    }
    u64* Wk = W + ((size_t)b * (N / 64) + kb) * N;
    if (SCATTER == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) Wk[col0 + j] = ((u64)wd[1][j] << 32) | wd[0][j];
    } else {
        const int4 t = *reinterpret_cast<const int4*>(rankof + (size_t)b * N + col0);
        const int rk[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) Wk[rk[j]] = ((u64)wd[1][j] << 32) | wd[0][j];
    }
}

// IoU write-kernel variants
__device__ __forceinline__ float bcastf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
template <bool NT, int WAVES, int TROWS>
__global__ __launch_bounds__(WAVES * 64) void iou_v(const float* __restrict__ A, int N, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int img = blockIdx.z, i0 = blockIdx.y * TROWS, c0 = (blockIdx.x * WAVES + wave) * 256;
    if (c0 >= N) return;
    const float* a = A + (size_t)img * N * 4;
    float* o = out + (size_t)img * N * N;
    float bx1[4], by1[4], bx2[4], by2[4], ba[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { float4 v = *reinterpret_cast<const float4*>(a + (size_t)(c0 + 4 * lane + j) * 4); bx1[j] = v.x; by1[j] = v.y; bx2[j] = v.z; by2[j] = v.w; ba[j] = (v.z - v.x) * (v.w - v.y); }
    for (int rr = 0; rr < TROWS; rr += 64) {
        float4 ra = *reinterpret_cast<const float4*>(a + (size_t)(i0 + rr + lane) * 4);
        const float rarea = (ra.z - ra.x) * (ra.w - ra.y);
        for (int r = 0; r < 64; ++r) {
            const float ax1 = bcastf(ra.x, r), ay1 = bcastf(ra.y, r), ax2 = bcastf(ra.z, r), ay2 = bcastf(ra.w, r), aa = bcastf(rarea, r);
            float res[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { float w = fmaxf(fminf(ax2, bx2[j]) - fmaxf(ax1, bx1[j]), 0.f), h = fmaxf(fminf(ay2, by2[j]) - fmaxf(ay1, by1[j]), 0.f); float in = w * h; res[j] = in / ((aa + ba[j]) - in); }
            float* dst = o + (size_t)(i0 + rr + r) * N + c0 + 4 * lane;
            if (NT) { __builtin_nontemporal_store(res[0], dst); __builtin_nontemporal_store(res[1], dst + 1); __builtin_nontemporal_store(res[2], dst + 2); __builtin_nontemporal_store(res[3], dst + 3); }
            else *reinterpret_cast<float4*>(dst) = make_float4(res[0], res[1], res[2], res[3]);
        }
    }
}
__global__ __launch_bounds__(256) void write_nt_kernel(float* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float* d = out + i * 4; __builtin_nontemporal_store(1.f, d); __builtin_nontemporal_store(2.f, d + 1); __builtin_nontemporal_store(3.f, d + 2); __builtin_nontemporal_store(4.f, d + 3); }
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 4096, B = argc > 2 ? atoi(argv[2]) : 8;
    size_t nel = (size_t)B * N * N;
    float* iou; CK(hipMalloc(&iou, nel * 4));
    std::vector<float> h(1 << 20); for (auto& v : h) v = (float)rand() / RAND_MAX;
    for (size_t off = 0; off < nel; off += h.size()) CK(hipMemcpy(iou + off, h.data(), std::min(h.size(), nel - off) * 4, hipMemcpyHostToDevice));
    std::vector<int> ord((size_t)B * N), rk((size_t)B * N);
    for (int b = 0; b < B; ++b) { for (int i = 0; i < N; ++i) ord[(size_t)b * N + i] = i; for (int i = N - 1; i > 0; --i) { int j = rand() % (i + 1); std::swap(ord[(size_t)b * N + i], ord[(size_t)b * N + j]); } for (int i = 0; i < N; ++i) rk[(size_t)b * N + ord[(size_t)b * N + i]] = i; }
    int *d_ord, *d_rk; CK(hipMalloc(&d_ord, ord.size() * 4)); CK(hipMalloc(&d_rk, rk.size() * 4));
    CK(hipMemcpy(d_ord, ord.data(), ord.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_rk, rk.data(), rk.size() * 4, hipMemcpyHostToDevice));
    u64* W; CK(hipMalloc(&W, (size_t)B * (N / 64) * N * 8));
    float* dummy; CK(hipMalloc(&dummy, 4));
    const double gb = nel * 4.0 / 1e9;
    auto rep = [&](const char* name, float us) { printf("%-44s %8.1f us  %7.1f GB/s\n", name, us, gb / (us * 1e-6)); };
    rep("read baseline (grid-stride float4 sum)", time_us([&] { read_kernel<<<256 * 8, 256>>>((const float4*)iou, nel / 4, dummy); }));
    rep("write baseline (grid-stride float4 fill)", time_us([&] { write_kernel<<<256 * 8, 256>>>((float4*)iou, nel / 4); }));
    for (size_t off = 0; off < nel; off += h.size()) CK(hipMemcpy(iou + off, h.data(), std::min(h.size(), nel - off) * 4, hipMemcpyHostToDevice));
#define RUN(RB, NT, SC, WV) { dim3 g((N + WV * 256 - 1) / (WV * 256), N / 64, B); char nm[96]; snprintf(nm, 96, "bitmask RB=%d NT=%d SCATTER=%d WAVES=%d", RB, NT, SC, WV); \
    rep(nm, time_us([&] { bitmask_v<RB, NT, SC, WV><<<g, WV * 64>>>(iou, N, d_ord, d_rk, 0.4f, W); })); }
    {
        std::vector<float> hb((size_t)B * N * 4);
        for (size_t i = 0; i < hb.size(); i += 4) { float cx = 1760.f * rand() / RAND_MAX, cy = 512.f * rand() / RAND_MAX, w = 16 + 120.f * rand() / RAND_MAX, hh = 16 + 120.f * rand() / RAND_MAX; hb[i] = cx - w / 2; hb[i + 1] = cy - hh / 2; hb[i + 2] = cx + w / 2; hb[i + 3] = cy + hh / 2; }
        float* boxes; CK(hipMalloc(&boxes, hb.size() * 4)); CK(hipMemcpy(boxes, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        float* out2; CK(hipMalloc(&out2, nel * 4));
        rep("write NT baseline", time_us([&] { write_nt_kernel<<<256 * 8, 256>>>(out2, nel / 4); }));
#define RUNI(NT, WV, TR) { dim3 g((N + WV * 256 - 1) / (WV * 256), N / TR, B); char nm[96]; snprintf(nm, 96, "iou NT=%d WAVES=%d TROWS=%d", NT, WV, TR); rep(nm, time_us([&] { iou_v<NT, WV, TR><<<g, WV * 64>>>(boxes, N, out2); })); }
        RUNI(false, 4, 64) RUNI(true, 4, 64) RUNI(false, 8, 64) RUNI(true, 8, 64) RUNI(true, 4, 128) RUNI(true, 4, 256) RUNI(true, 2, 64) RUNI(true, 16, 64) RUNI(false, 4, 256)
        CK(hipFree(out2));
    }
#define RUNR(RB, NT, SC, WV, MW) { dim3 g((N + WV * 256 - 1) / (WV * 256), N / 64, B); char nm[96]; snprintf(nm, 96, "rolled  RB=%d NT=%d SCATTER=%d WAVES=%d MINW=%d", RB, NT, SC, WV, MW); \
    rep(nm, time_us([&] { bitmask_r<RB, NT, SC, WV, MW><<<g, WV * 64>>>(iou, N, d_ord, d_rk, 0.4f, W); })); }
    RUNR(4, false, 1, 4, 1) RUNR(4, true, 1, 4, 1) RUNR(8, false, 1, 4, 1) RUNR(8, true, 1, 4, 1) RUNR(16, false, 1, 4, 1) RUNR(16, true, 1, 4, 1)
    RUNR(8, true, 1, 4, 8) RUNR(4, true, 1, 4, 8) RUNR(2, true, 1, 4, 8) RUNR(8, true, 1, 8, 4) RUNR(8, true, 1, 2, 4) RUNR(8, true, 0, 4, 1) RUNR(8, false, 0, 4, 1)
    return 0;
}
