// tools/usolve_ticks.hip -- developer microbenchmark (not part of the product): phase ticks (s_memtime: ~2.3 per ns on this part) of the inversion of one
// 128 x 128 diagonal tile as ungrouped_solve_forward_kernel does it, and a check of the result against a substitution on the host.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I groomed_nms_amd/csrc tools/usolve_ticks.hip -o build/usolve_ticks && build/usolve_ticks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <string.h>
__device__ long long g_ticks[16];
#define GNMS_UTICK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_ticks[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
__device__ long long g_hop[64 * 8];
#define GNMS_UHOP(blk, slot) do { if (blockIdx.y == 0 && (threadIdx.x == 0 || threadIdx.x == 512)) g_hop[(blk) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#include "nms_kernels.h"
#include "nms_solve_kernels.h"
using namespace gnms;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(512) void invert_kernel(const float* __restrict__ T, float* __restrict__ D) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Tl = reinterpret_cast<float*>(smem);
    float* Dl = Tl + kUB * kUP;
    const int t = threadIdx.x;
    for (int e = t; e < kUB * kUB; e += 512) { const int i = e / kUB, j = e % kUB; Tl[i * kUP + j] = j < i ? T[(size_t)blockIdx.x * kUB * kUB + e] : 0.0f; Dl[i * kUP + j] = 0.0f; }
    __syncthreads();
    GNMS_UTICK(0);
    ungrouped_invert_diag32(Tl, Dl, t);
    ungrouped_invert_offdiag(Tl, Dl, t);
    for (int e = t; e < kUB * kUB; e += 512) { const int i = e / kUB, j = e % kUB; D[(size_t)blockIdx.x * kUB * kUB + e] = Dl[i * kUP + j]; }
}

int main() {
    const int nb = 256;
    std::vector<float> T((size_t)nb * kUB * kUB);
    srand(1);
    for (auto& v : T) v = (rand() % 100 < 30) ? (float)rand() / RAND_MAX : 0.0f;
    float *dT, *dD;
    CK(hipMalloc(&dT, T.size() * 4)); CK(hipMalloc(&dD, T.size() * 4));
    CK(hipMemcpy(dT, T.data(), T.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)2 * kUB * kUP * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(invert_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 3; ++rep) {
        invert_kernel<<<nb, 512, lds>>>(dT, dD);
        CK(hipDeviceSynchronize());
        long long h[16];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ticks), sizeof(h)));
        printf("ticks: diag32 %lld | barrier %lld | 32 -> 64 %lld | barrier + product 1 %lld | barrier + product 2 %lld | barrier %lld | total %lld\n", h[1] - h[0], h[2] - h[1],
               h[6] - h[2], h[3] - h[6], h[4] - h[3], h[5] - h[4], h[5] - h[0]);
    }
    // the forward solve itself: when each block of image 0 sees its last source (poller), has its right-hand side, has passed the
    // barrier, has published
    for (int N : {512, 4096}) {
        const int B = 8;
        const gnms_ws_layout L = gnms_make_layout(N);
        const size_t ldp = ungrouped_ld(N);
        const size_t wsb = (size_t)B * L.per_image + ungrouped_scratch_bytes(B, N);
        char* ws; float* sc;
        CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&sc, (size_t)B * N * 4));
        std::vector<float> Ps((size_t)B * N * ldp), scores((size_t)B * N);
        for (auto& v : Ps) v = (rand() % 100 < 3) ? 0.5f * rand() / RAND_MAX : 0.0f;
        for (auto& v : scores) v = (float)rand() / RAND_MAX;
        gnms_params P; memset(&P, 0, sizeof(P)); P.presorted = 1; P.nms_threshold = 0.4f;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ungrouped_solve_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUngroupedFwdLds));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ws, 0, (size_t)B * L.per_image));
            CK(hipMemcpy(ws + (size_t)B * L.per_image, Ps.data(), Ps.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(sc, scores.data(), scores.size() * 4, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            ungrouped_solve_forward_kernel<<<dim3(N / kUB, B), kUThreads, kUngroupedFwdLds>>>(sc, N, nullptr, P, ws, L, reinterpret_cast<float*>(ws + (size_t)B * L.per_image));
            CK(hipDeviceSynchronize());
        }
        long long h[64 * 8];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_hop), sizeof(h)));
        const int nb = N / kUB;
        printf("N = %d: block: seen-by-poller after the source's publish | rhs ready | barrier | published (ticks, ~2.3 per ns)\n", N);
        for (int k = 1; k < nb; k += (nb > 8 ? 5 : 1))
            printf("  block %2d: %6lld | %6lld | %6lld | %6lld   (hop %lld)\n", k, h[k * 8] - h[(k - 1) * 8 + 3], h[k * 8 + 1] - h[k * 8], h[k * 8 + 2] - h[k * 8 + 1],
                   h[k * 8 + 3] - h[k * 8 + 2], h[k * 8 + 3] - h[(k - 1) * 8 + 3]);
        CK(hipFree(ws)); CK(hipFree(sc));
    }
    std::vector<float> D(T.size());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    // check block 0: (I + T) D = I
    double worst = 0;
    for (int blk = 0; blk < 2; ++blk)
        for (int i = 0; i < kUB; ++i)
            for (int j = 0; j < kUB; ++j) {
                double acc = D[(size_t)blk * kUB * kUB + i * kUB + j];
                for (int k = 0; k < i; ++k) acc += (double)T[(size_t)blk * kUB * kUB + i * kUB + k] * D[(size_t)blk * kUB * kUB + k * kUB + j];
                worst = fmax(worst, fabs(acc - (i == j ? 1.0 : 0.0)));
            }
    printf("max |(I + T) D - I| = %.3e\n", worst);
    return 0;
}
