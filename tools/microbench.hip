// tools/microbench.hip -- developer microbenchmarks of the single-workgroup NMS kernels (not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I groomed_nms_amd/csrc tools/microbench.hip -o /tmp/microbench -L groomed_nms_amd -lgroomed_nms_hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "nms_kernels.h"
using namespace gnms;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int E>
__global__ __launch_bounds__(1024) void sort_only_bitonic_kernel(const u64* in, u64* out, int P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    u64 r[E];
    const u64* src = in + (size_t)blockIdx.x * P;
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = src[threadIdx.x * E + e];
    block_sort_bitonic<E, u64>(r, keys, P);
#pragma unroll
    for (int e = 0; e < E; ++e) out[(size_t)blockIdx.x * P + threadIdx.x * E + e] = r[e];
}

template <int E>
__global__ __launch_bounds__(1024) void sort_only_kernel(const u64* in, u64* out, int P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    u64 r[E];
    const u64* src = in + (size_t)blockIdx.x * P;
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = src[threadIdx.x * E + e];
    block_sort<E, u64>(r, keys, P);
#pragma unroll
    for (int e = 0; e < E; ++e) out[(size_t)blockIdx.x * P + threadIdx.x * E + e] = r[e];
}

// experiment: the from-boxes bit-matrix kernel with the division replaced (0 exact, 1 rcp-multiply, 2 cross-multiply)
template <int DIVMODE, int UNR>
__global__ __launch_bounds__(256) void bitmask_boxes_exp(const float* __restrict__ boxes, int N, float thr, char* ws, gnms_ws_layout L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, kb = blockIdx.y;
    const int n = N;
    const int k0 = kb * 64;
    const int c0 = (blockIdx.x * 4 + wave) * 256;
    if (k0 >= n || c0 >= n || c0 >= k0 + 64) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    const float4* bx = reinterpret_cast<const float4*>(boxes) + (size_t)b * N;
    float4 cb[4]; float carea[4]; int col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { col[j] = c0 + 4 * lane + j; cb[j] = bx[I.order[col[j] < n ? col[j] : n - 1]]; carea[j] = (cb[j].z - cb[j].x) * (cb[j].w - cb[j].y); }
    const float4 rb = bx[I.order[min(k0 + lane, n - 1)]];
    const float rarea = (rb.z - rb.x) * (rb.w - rb.y);
    unsigned wd[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll UNR
        for (int rr = 0; rr < 32; ++rr) {
            const int r = half * 32 + rr;
            const float ax1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.x), r));
            const float ay1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.y), r));
            const float ax2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.z), r));
            const float ay2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.w), r));
            const float aa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rarea), r));
            const unsigned bit = 1u << rr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = fmaxf(fminf(ax2, cb[j].z) - fmaxf(ax1, cb[j].x), 0.0f);
                const float h = fmaxf(fminf(ay2, cb[j].w) - fmaxf(ay1, cb[j].y), 0.0f);
                const float inter = w * h;
                const float uni = (aa + carea[j]) - inter;
                bool nl;
                if (DIVMODE == 0) nl = !(inter / uni <= thr);
                else if (DIVMODE == 1) nl = !(inter * __builtin_amdgcn_rcpf(uni) <= thr);
                else nl = !(inter <= thr * uni);
                wd[half][j] |= nl ? bit : 0u;
            }
        }
    }
    u64* Wk = I.W + (size_t)kb * L.NC;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (col[j] < n) Wk[col[j]] = (((u64)wd[1][j] << 32) | wd[0][j]);
}

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }

template <typename F>
float time_us(F f, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 4096;
    int B = argc > 2 ? atoi(argv[2]) : 8;
    int per = argc > 3 ? atoi(argv[3]) : 64;     // boxes per cluster (0 = uniform)
    srand(1);
    auto U = []() { return (float)rand() / RAND_MAX; };
    std::vector<float> boxes((size_t)B * N * 4), scores((size_t)B * N);
    for (int b = 0; b < B; ++b) {
        int K = per > 0 ? (N / per > 0 ? N / per : 1) : N;
        std::vector<float> base(K * 4);
        for (int k = 0; k < K; ++k) { float cx = U() * 1760, cy = U() * 512, w = 16 + U() * 120, h = 16 + U() * 120; base[k*4]=cx; base[k*4+1]=cy; base[k*4+2]=w; base[k*4+3]=h; }
        for (int i = 0; i < N; ++i) {
            int k = per > 0 ? i % K : i;
            float cx = base[k*4], cy = base[k*4+1], w = base[k*4+2], h = base[k*4+3];
            if (per > 0) { cx += (U() - 0.5f) * 0.35f * w; cy += (U() - 0.5f) * 0.35f * h; w *= 0.85f + 0.3f * U(); h *= 0.85f + 0.3f * U(); }
            float* p = &boxes[((size_t)b * N + i) * 4];
            p[0] = cx - w / 2; p[1] = cy - h / 2; p[2] = cx + w / 2; p[3] = cy + h / 2;
            scores[(size_t)b * N + i] = U() + 1e-7f * i;
        }
    }
    float *d_boxes, *d_scores, *d_iou, *d_prob; char* ws;
    CK(hipMalloc(&d_boxes, boxes.size() * 4)); CK(hipMalloc(&d_scores, scores.size() * 4));
    CK(hipMalloc(&d_iou, (size_t)B * N * N * 4)); CK(hipMalloc(&d_prob, (size_t)B * N * 4));
    CK(hipMemcpy(d_boxes, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_scores, scores.data(), scores.size() * 4, hipMemcpyHostToDevice));
    gnms_params P; gnms_default_params(&P);
    size_t wsb = gnms_workspace_bytes(B, N, &P);
    CK(hipMalloc(&ws, wsb));
    gnms_ws_layout L = gnms_make_layout(N);
    if (gnms_iou2d(d_boxes, d_boxes, B, N, N, d_iou, N, nullptr)) { printf("iou2d failed %s\n", gnms_last_error()); return 1; }
    if (gnms_forward(d_scores, d_iou, B, N, N, nullptr, &P, d_prob, nullptr, nullptr, nullptr, nullptr, nullptr, ws, wsb, nullptr)) { printf("fwd failed %s\n", gnms_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    int nlead[8]; for (int b = 0; b < B && b < 8; ++b) { CK(hipMemcpy(&nlead[b], img_ptrs(ws, L, b).misc, 4, hipMemcpyDeviceToHost)); }
    printf("N=%d B=%d per=%d leaders(img0)=%d\n", N, B, per, nlead[0]); setvbuf(stdout, NULL, _IONBF, 0);

    int P2 = 64; while (P2 < N) P2 <<= 1;
    const size_t sort_lds = (size_t)P2 * 8;
    const int T = P2 <= 1024 ? P2 : 1024;
    printf("empty kernel            %8.1f us\n", time_us([&] { empty_kernel<<<B, 1024>>>(nullptr); }));
    printf("iou2d                   %8.1f us\n", time_us([&] { gnms_iou2d(d_boxes, d_boxes, B, N, N, d_iou, N, nullptr); }));
    printf("forward (all)           %8.1f us\n", time_us([&] { gnms_forward(d_scores, d_iou, B, N, N, nullptr, &P, d_prob, nullptr, nullptr, nullptr, nullptr, nullptr, ws, wsb, nullptr); }));
    const size_t llds = leaders_lds_size(L.NB);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(leaders_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
    printf("leaders                 %8.1f us\n", time_us([&] { leaders_kernel<<<B, 1024, llds>>>(N, nullptr, ws, L); }));
#ifdef GNMS_TIMING
    {
        long long z[8] = {0}; CK(hipMemcpy(img_ptrs(ws, L, 0).gx, z, sizeof(z), hipMemcpyHostToDevice));
        leaders_kernel<<<B, 1024, llds>>>(N, nullptr, ws, L); CK(hipDeviceSynchronize());
        CK(hipMemcpy(z, img_ptrs(ws, L, 0).gx, sizeof(z), hipMemcpyDeviceToHost));
        printf("  leaders phases (cycles, wave0): prologue %lld | resolve %lld | barrierA %lld | push+book %lld | barrierB %lld\n", z[0], z[1], z[2], z[3], z[4]);
    }
#endif
    {
        const bool vec = true;
        dim3 gm((N + kMaskWaves * 256 - 1) / (kMaskWaves * 256), L.NB, B);
        float tb = time_us([&] { bitmask_kernel<true><<<gm, kMaskWaves * 64>>>(d_iou, N, N, nullptr, 0.4f, ws, L); });
        float tbl = time_us([&] { bitmask_kernel<true><<<gm, kMaskWaves * 64>>>(d_iou, N, N, nullptr, 0.4f, ws, L); leaders_kernel<<<B, 1024, llds>>>(N, nullptr, ws, L); });
        printf("bitmask                 %8.1f us ; bitmask+leaders %8.1f us -> leaders cold %8.1f us\n", tb, tbl, tbl - tb);
        (void)vec;
#ifdef GNMS_TIMING
        long long z[8] = {0}; CK(hipMemcpy(img_ptrs(ws, L, 0).gx, z, sizeof(z), hipMemcpyHostToDevice));
        bitmask_kernel<true><<<gm, kMaskWaves * 64>>>(d_iou, N, N, nullptr, 0.4f, ws, L);
        leaders_kernel<<<B, 1024, llds>>>(N, nullptr, ws, L); CK(hipDeviceSynchronize());
        CK(hipMemcpy(z, img_ptrs(ws, L, 0).gx, sizeof(z), hipMemcpyDeviceToHost));
        printf("  COLD leaders phases (cycles, wave0): prologue %lld | resolve %lld | barrierA %lld | push+book %lld | barrierB %lld\n", z[0], z[1], z[2], z[3], z[4]);
#endif
    }
    {
        dim3 gb((N + 1023) / 1024, L.NB, B);
        printf("bitmask_boxes exact  u4 %8.1f us\n", time_us([&] { bitmask_boxes_exp<0, 4><<<gb, 256>>>(d_boxes, N, 0.4f, ws, L); }));
        printf("bitmask_boxes exact  u1 %8.1f us\n", time_us([&] { bitmask_boxes_exp<0, 1><<<gb, 256>>>(d_boxes, N, 0.4f, ws, L); }));
        printf("bitmask_boxes exact u32 %8.1f us\n", time_us([&] { bitmask_boxes_exp<0, 32><<<gb, 256>>>(d_boxes, N, 0.4f, ws, L); }));
        printf("bitmask_boxes rcp    u4 %8.1f us\n", time_us([&] { bitmask_boxes_exp<1, 4><<<gb, 256>>>(d_boxes, N, 0.4f, ws, L); }));
        printf("bitmask_boxes xmul   u4 %8.1f us\n", time_us([&] { bitmask_boxes_exp<2, 4><<<gb, 256>>>(d_boxes, N, 0.4f, ws, L); }));
        // restore W for the kernels below
        bitmask_kernel<true><<<dim3((N + kMaskWaves * 256 - 1) / (kMaskWaves * 256), L.NB, B), kMaskWaves * 64>>>(d_iou, N, N, nullptr, 0.4f, ws, L);
    }
    printf("attribute               %8.1f us\n", time_us([&] { attribute_kernel<false><<<dim3(L.NB, B), 64>>>(d_iou, (long)N, N, nullptr, ws, L); }));
    if (P2 == 4096) {
        printf("sort_scores<4>          %8.1f us\n", time_us([&] { sort_scores_kernel<4><<<B, T, sort_lds>>>(d_scores, N, nullptr, ws, L, P2, nullptr, nullptr); }));
        // groups_kernel consumes the leader ordinals attribute_kernel leaves in gpos (and overwrites gpos): time the pair
        {
            const float ta = time_us([&] { attribute_kernel<false><<<dim3(L.NB, B), 64>>>(d_iou, (long)N, N, nullptr, ws, L); });
            const float tag = time_us([&] { attribute_kernel<false><<<dim3(L.NB, B), 64>>>(d_iou, (long)N, N, nullptr, ws, L); groups_kernel<4, false><<<B, T, sort_lds>>>(d_iou, N, N, nullptr, P, ws, L, P2); });
            printf("groups<4>               %8.1f us (attribute + groups %.1f)\n", tag - ta, tag);
        }
#ifdef GNMS_TIMING
        {
            long long z[16] = {0}; CK(hipMemcpy(img_ptrs(ws, L, 0).gx, z, sizeof(z), hipMemcpyHostToDevice));
            attribute_kernel<false><<<dim3(L.NB, B), 64>>>(d_iou, (long)N, N, nullptr, ws, L);
            groups_kernel<4, false><<<B, T, sort_lds>>>(d_iou, N, N, nullptr, P, ws, L, P2); CK(hipDeviceSynchronize());
            CK(hipMemcpy(z, img_ptrs(ws, L, 0).gx, sizeof(z), hipMemcpyDeviceToHost));
            printf("  groups phases (cycles, thread0): keys %lld | sort %lld | runs %lld | rescoring %lld\n", z[8], z[9], z[10], z[11]);
        }
#endif
        printf("finalize<4>             %8.1f us\n", time_us([&] { finalize_kernel<4><<<B, T, sort_lds>>>(N, nullptr, P, ws, L, P2, d_prob, nullptr, nullptr, nullptr, nullptr); }));
#ifdef GNMS_TIMING
        {
            long long z[16] = {0}; CK(hipMemcpy(img_ptrs(ws, L, 0).gx, z, sizeof(z), hipMemcpyHostToDevice));
            finalize_kernel<4><<<B, T, sort_lds>>>(N, nullptr, P, ws, L, P2, d_prob, nullptr, nullptr, nullptr, nullptr); CK(hipDeviceSynchronize());
            CK(hipMemcpy(z, img_ptrs(ws, L, 0).gx, sizeof(z), hipMemcpyDeviceToHost));
            printf("  finalize phases (cycles, thread0): classify+compact %lld | sort %lld | write %lld\n", z[12], z[13], z[14]);
        }
#endif
        std::vector<u64> hk((size_t)B * P2);
        for (auto& v : hk) v = ((u64)rand() << 32) ^ (u64)rand() ^ ((u64)rand() << 17);
        u64 *din, *dout; CK(hipMalloc(&din, hk.size() * 8)); CK(hipMalloc(&dout, hk.size() * 8));
        CK(hipMemcpy(din, hk.data(), hk.size() * 8, hipMemcpyHostToDevice));
        printf("block_sort<4> only      %8.1f us\n", time_us([&] { sort_only_kernel<4><<<B, 1024, sort_lds>>>(din, dout, P2); }));
#ifdef GNMS_TIMING
        { long long z[4] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_sort_ticks), z, sizeof(z)));
          sort_only_kernel<4><<<B, 1024, sort_lds>>>(din, dout, P2); CK(hipDeviceSynchronize());
          CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_sort_ticks), sizeof(z)));
          printf("  block_sort<4,u64> cycles: LDS stages %lld | shuffle stages %lld | register stages %lld | final %lld\n", z[0], z[1], z[2], z[3]); }
#endif
        printf("bitonic<4> only         %8.1f us\n", time_us([&] { sort_only_bitonic_kernel<4><<<B, 1024, sort_lds>>>(din, dout, P2); }));
        printf("block_sort<8> only      %8.1f us\n", time_us([&] { sort_only_kernel<8><<<B, 512, sort_lds>>>(din, dout, P2); }));
        printf("block_sort<16> only     %8.1f us\n", time_us([&] { sort_only_kernel<16><<<B, 256, sort_lds>>>(din, dout, P2); }));
        std::vector<u64> ho(hk.size()); CK(hipMemcpy(ho.data(), dout, ho.size() * 8, hipMemcpyDeviceToHost));
        bool ok = true; for (int b = 0; b < B; ++b) for (int i = 1; i < P2; ++i) if (ho[(size_t)b * P2 + i - 1] > ho[(size_t)b * P2 + i]) ok = false;
        printf("sorted ok: %d\n", (int)ok);
    }
    if (P2 == 16384) {
        long long* d_valid; CK(hipMalloc(&d_valid, (size_t)B * N * 8 * 2));
        int* d_nv; CK(hipMalloc(&d_nv, B * 8));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_scores_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(groups_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
        printf("sort_scores<16>         %8.1f us\n", time_us([&] { sort_scores_kernel<16><<<B, T, sort_lds>>>(d_scores, N, nullptr, ws, L, P2, nullptr, nullptr); }));
        {
            const float ta = time_us([&] { attribute_kernel<false><<<dim3(L.NB, B), 64>>>(d_iou, (long)N, N, nullptr, ws, L); });
            const float tag = time_us([&] { attribute_kernel<false><<<dim3(L.NB, B), 64>>>(d_iou, (long)N, N, nullptr, ws, L); groups_kernel<16, false><<<B, T, sort_lds>>>(d_iou, N, N, nullptr, P, ws, L, P2); });
            printf("groups<16>              %8.1f us (attribute + groups %.1f)\n", tag - ta, tag);
        }
        printf("finalize<16>            %8.1f us\n", time_us([&] { finalize_kernel<16><<<B, T, sort_lds>>>(N, nullptr, P, ws, L, P2, d_prob, d_valid, d_valid + (size_t)B * N, d_nv, d_nv + B); }));
#ifdef GNMS_TIMING
        {
            long long z[16] = {0}; CK(hipMemcpy(img_ptrs(ws, L, 0).gx, z, sizeof(z), hipMemcpyHostToDevice));
            attribute_kernel<false><<<dim3(L.NB, B), 64>>>(d_iou, (long)N, N, nullptr, ws, L);
            groups_kernel<16, false><<<B, T, sort_lds>>>(d_iou, N, N, nullptr, P, ws, L, P2);
            finalize_kernel<16><<<B, T, sort_lds>>>(N, nullptr, P, ws, L, P2, d_prob, d_valid, d_valid + (size_t)B * N, d_nv, d_nv + B); CK(hipDeviceSynchronize());
            CK(hipMemcpy(z, img_ptrs(ws, L, 0).gx, sizeof(z), hipMemcpyDeviceToHost));
            printf("  groups phases (cycles, thread0): keys %lld | sort %lld | runs %lld | rescoring %lld\n", z[8], z[9], z[10], z[11]);
            printf("  finalize phases (cycles, thread0): classify+compact %lld | sort %lld | write %lld\n", z[12], z[13], z[14]);
        }
#endif
    }
    return 0;
}
