// does s_atomic_add (scalar memory atomic, returns through lgkmcnt) work on gfx950?  every workgroup claims tickets until the
// counter passes `total`; each ticket must be handed out exactly once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void claim_kernel(int* counter, int total, int* seen) {
    for (;;) {
        int t = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(counter) : "memory");
        if (t >= total) break;
        if (threadIdx.x == 0) atomicAdd(&seen[t], 1);
    }
}
int main() {
    int *counter, *seen; const int total = 100000;
    hipMalloc(&counter, 4); hipMalloc(&seen, total * 4);
    hipMemset(counter, 0, 4); hipMemset(seen, 0, total * 4);
    claim_kernel<<<1024, 256>>>(counter, total, seen);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    std::vector<int> h(total); int c;
    hipMemcpy(h.data(), seen, total * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, counter, 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int v : h) bad += (v != 1);
    printf("counter %d (expect >= %d), tickets not handed out exactly once: %d\n", c, total, bad);
    return bad != 0;
}
