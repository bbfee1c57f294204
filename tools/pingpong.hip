// tools/pingpong.hip -- developer microbenchmark (not part of the product): the round trip of a flag between two workgroups,
// by the workgroups' XCDs and by the cache-control bits of the poll.  A workgroup's XCD is read from XCC_ID; workgroup i of a
// launch goes to XCD i % 8 (checked here, not assumed).
//   store: agent-scope atomic (sc1, write-through)        poll: agent-scope atomic (sc1)           -- what the chains do today
//   store: agent-scope atomic                             poll: sc0 only (misses the L1, may hit the XCD's L2)
//   store: plain + s_waitcnt                              poll: sc0 only
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/pingpong.hip -o /tmp/pingpong && /tmp/pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ u64 load_sc0(const u64* p) {
    u64 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u64 load_sc1(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_sc1(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_plain(u64* p, u64 v) {
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_sc0(u64* p, u64 v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}

// MODE 0: sc1 / sc1; 1: store sc1, poll sc0; 2: store plain, poll sc0; 3: store sc1, poll alternating sc0 / sc1; 4: store sc0, poll sc0
template <int MODE>
__device__ __forceinline__ u64 poll(const u64* p, u64 want) {
    u64 v;
    for (int it = 0; it < 200000; ++it) {                 // bounded: a poll that can never see the store must not hang the box
        if (MODE == 0) { v = load_sc1(p); if (v >= want) return v; }
        else if (MODE == 3) { v = load_sc0(p); if (v >= want) return v; v = load_sc1(p); if (v >= want) return v; }
        else { v = load_sc0(p); if (v >= want) return v; }
    }
    return ~0ull;                                          // gave up
}
template <int MODE>
__device__ __forceinline__ void put(u64* p, u64 v) {
    if (MODE == 0 || MODE == 1 || MODE == 3) store_sc1(p, v);
    else if (MODE == 2) store_plain(p, v);
    else store_sc0(p, v);
}

// workgroups wa and wb play; every other workgroup leaves at once.  flag[0]: a -> b, flag[32]: b -> a (different 128-byte lines)
template <int MODE>
__global__ void pingpong_kernel(u64* flag, int wa, int wb, int iters, long long* ticks, int* xcc) {
    const int w = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (w == wa || w == wb) xcc[w == wa ? 0 : 1] = __builtin_amdgcn_s_getreg(63508) & 15;
    if (w == wa) {
        const long long t0 = wall_clock64();
        for (int i = 1; i <= iters; ++i) {
            put<MODE>(flag, (u64)i);
            if (poll<MODE>(flag + 32, (u64)i) == ~0ull) { ticks[1] = i; store_sc1(flag, ~0ull - 1); break; }
        }
        ticks[0] = wall_clock64() - t0;
    } else if (w == wb) {
        for (int i = 1; i <= iters; ++i) {
            if (poll<MODE>(flag, (u64)i) == ~0ull) { ticks[2] = i; store_sc1(flag + 32, ~0ull - 1); break; }
            put<MODE>(flag + 32, (u64)i);
        }
    }
}

// one producer, many consumers polling the same word (a grid barrier's fan-out): consumers report the delay from the producer's store
template <int MODE>
__global__ void fanout_kernel(u64* flag, int nwg, long long* ticks) {
    const int w = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (w == 0) {
        // wait until everybody is polling
        for (int it = 0; it < 1000000 && __hip_atomic_load(flag + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u64)(nwg - 1); ++it) {}
        for (int k = 0; k < 2000; ++k) __builtin_amdgcn_s_sleep(10);
        const long long t0 = wall_clock64();
        ticks[0] = t0;
        put<MODE>(flag, 1ull);
    } else {
        __hip_atomic_fetch_add(flag + 64, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        poll<MODE>(flag, 1ull);
        ticks[w] = wall_clock64();
    }
}

template <int MODE>
void run(const char* name, u64* flag, long long* ticks, int* xcc, int wa, int wb, int nwg) {
    const int iters = 2000;
    CK(hipMemset(flag, 0, 1024)); CK(hipMemset(ticks, 0, 64));
    pingpong_kernel<MODE><<<nwg, 64>>>(flag, wa, wb, iters, ticks, xcc);
    CK(hipDeviceSynchronize());
    long long t, tt[3]; int x[2];
    CK(hipMemcpy(tt, ticks, 24, hipMemcpyDeviceToHost)); t = tt[0];
    if (tt[1] || tt[2]) { printf("%-34s wg %3d <-> wg %3d: GAVE UP (a at %lld, b at %lld)\n", name, wa, wb, tt[1], tt[2]); fflush(stdout); return; }
    CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
    printf("%-34s wg %3d (xcc %d) <-> wg %3d (xcc %d): round trip %.3f us (one way %.3f)\n", name, wa, x[0], wb, x[1], t / 100.0 / iters, t / 200.0 / iters); fflush(stdout);
}

int main() {
    u64* flag; long long* ticks; int* xcc;
    CK(hipMalloc(&flag, 4096)); CK(hipMalloc(&ticks, 8 * 1024)); CK(hipMalloc(&xcc, 64));
    const int nwg = 64;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("store sc1, poll sc1", flag, ticks, xcc, 0, 8, nwg);
        run<0>("store sc1, poll sc1", flag, ticks, xcc, 0, 1, nwg);
        run<1>("store sc1, poll sc0", flag, ticks, xcc, 0, 8, nwg);
        run<3>("store sc1, poll sc0/sc1 alternating", flag, ticks, xcc, 0, 8, nwg);
        run<3>("store sc1, poll sc0/sc1 alternating", flag, ticks, xcc, 0, 1, nwg);
        run<2>("store plain, poll sc0", flag, ticks, xcc, 0, 8, nwg);
        run<4>("store sc0, poll sc0", flag, ticks, xcc, 0, 8, nwg);
        run<2>("store plain, poll sc0", flag, ticks, xcc, 0, 16, nwg);
    }
    // fan-out: 255 pollers
    for (int mode = 0; mode < 2; ++mode) {
        const int n = 256;
        CK(hipMemset(flag, 0, 1024)); CK(hipMemset(ticks, 0, 8 * 1024));
        if (mode == 0) fanout_kernel<0><<<n, 64>>>(flag, n, ticks); else fanout_kernel<3><<<n, 64>>>(flag, n, ticks);
        CK(hipDeviceSynchronize());
        long long h[256];
        CK(hipMemcpy(h, ticks, 8 * n, hipMemcpyDeviceToHost));
        long long mx = 0, mn = 1ll << 60; double sum = 0;
        for (int i = 1; i < n; ++i) { long long d = h[i] - h[0]; if (d > mx) mx = d; if (d < mn) mn = d; sum += d; }
        printf("fan-out to 255 pollers, %s: min %.3f avg %.3f max %.3f us\n", mode == 0 ? "poll sc1" : "poll sc0/sc1", mn / 100.0, sum / 255 / 100.0, mx / 100.0); fflush(stdout);
    }
    return 0;
}
