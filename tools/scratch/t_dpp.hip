#include <hip/hip_runtime.h>
__device__ __forceinline__ unsigned lane_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned lane_xor2(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned lane_xor8(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned lane_xor4(unsigned v, int lane) {
    unsigned a = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x104, 0xF, 0xF, true);   // row_shl:4 : lane i <- i+4
    unsigned b = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x114, 0xF, 0xF, true);   // row_shr:4 : lane i <- i-4
    return (lane & 4) ? b : a;
}
__device__ __forceinline__ unsigned lane_xor16(unsigned v, int lane) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (lane & 16) ? r[0] : r[1];
}
__device__ __forceinline__ unsigned lane_xor32(unsigned v, int lane) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? r[0] : r[1];
}
__global__ void k(unsigned* out) {
    int lane = threadIdx.x & 63;
    unsigned v = threadIdx.x * 7 + 1;
    out[threadIdx.x * 8 + 0] = lane_xor1(v);
    out[threadIdx.x * 8 + 1] = lane_xor2(v);
    out[threadIdx.x * 8 + 2] = lane_xor4(v, lane);
    out[threadIdx.x * 8 + 3] = lane_xor8(v);
    out[threadIdx.x * 8 + 4] = lane_xor16(v, lane);
    out[threadIdx.x * 8 + 5] = lane_xor32(v, lane);
    out[threadIdx.x * 8 + 6] = __shfl_xor(v, 4, 64);
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8 * 4); k<<<1, 64>>>(d);
    unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0; const int xs[6] = {1, 2, 4, 8, 16, 32};
    for (int t = 0; t < 64; ++t) for (int j = 0; j < 6; ++j) { unsigned want = (t ^ xs[j]) * 7 + 1; if (h[t * 8 + j] != want) { if (bad < 10) printf("lane %d xor %d got %u want %u\n", t, xs[j], h[t * 8 + j], want); ++bad; } }
    printf("bad=%d\n", bad); return bad != 0;
}
