// tools/mall_probe.hip -- does the 256-MiB Infinity Cache keep what a streaming WRITE left behind, and does a read that starts with the
// youngest bytes get them back faster than HBM delivers?  (developer tool, not product: the question behind reading the images of the
// two-call path -- iou2d writes B matrices, differentiable_nms reads them -- in REVERSE order.)
// Build: hipcc --offload-arch=gfx950 -O3 -o build/mall_probe tools/mall_probe.hip ; run: build/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// segment s of nseg (64 MiB each, "an image"); forward or reversed segment order; NT or ordinary accesses
template <bool NT> __global__ __launch_bounds__(1024) void fill_kernel(float* out, size_t seg_floats, int reverse) {
    const int s = reverse ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    float* p = out + (size_t)s * seg_floats;
    const size_t per = seg_floats / gridDim.x;                       // a block's contiguous share
    float* q = p + (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x * 4; i < per; i += 4096) {
        if (NT) { __builtin_nontemporal_store(1.f, q + i); __builtin_nontemporal_store(2.f, q + i + 1); __builtin_nontemporal_store(3.f, q + i + 2); __builtin_nontemporal_store(4.f, q + i + 3); }
        else *reinterpret_cast<float4*>(q + i) = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
template <bool NT> __global__ __launch_bounds__(1024) void read_kernel(const float* in, size_t seg_floats, int reverse, float* sink) {
    const int s = reverse ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    const float* p = in + (size_t)s * seg_floats;
    const size_t per = seg_floats / gridDim.x;
    const float* q = p + (size_t)blockIdx.x * per;
    float acc = 0.f;
    for (size_t i = threadIdx.x * 4; i < per; i += 4096) {
        if (NT) acc += __builtin_nontemporal_load(q + i) + __builtin_nontemporal_load(q + i + 1) + __builtin_nontemporal_load(q + i + 2) + __builtin_nontemporal_load(q + i + 3);
        else { const float4 v = *reinterpret_cast<const float4*>(q + i); acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 123.456f) *sink = acc;
}

int main() {
    const int nseg = 8;
    const size_t seg_floats = (size_t)4096 * 4096;                     // 64 MiB
    float *buf, *other, *sink;
    CK(hipMalloc(&buf, nseg * seg_floats * 4)); CK(hipMalloc(&other, nseg * seg_floats * 4)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const dim3 grid(128, nseg), blk(1024);
    printf("8 segments x 64 MiB; write then read the same buffer; us per pass (median of 9), bytes = 512 MiB each\n");
    for (int wnt = 0; wnt < 2; ++wnt) for (int rnt = 0; rnt < 2; ++rnt) for (int rev = 0; rev < 2; ++rev) for (int live = 1; live <= 8; live *= 2) {
        // `live` = how many segments the pass covers (live * 64 MiB): 1, 2, 4 fit the cache, 8 does not
        float tw[9], tr[9];
        for (int it = 0; it < 9; ++it) {
            // flush: stream through the other buffer
            fill_kernel<true><<<grid, blk>>>(other, seg_floats, 0);
            CK(hipEventRecord(e0));
            if (wnt) fill_kernel<true><<<dim3(128, live), blk>>>(buf, seg_floats, 0); else fill_kernel<false><<<dim3(128, live), blk>>>(buf, seg_floats, 0);
            CK(hipEventRecord(e1));
            if (rnt) read_kernel<true><<<dim3(128, live), blk>>>(buf, seg_floats, rev, sink); else read_kernel<false><<<dim3(128, live), blk>>>(buf, seg_floats, rev, sink);
            CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
            float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2)); tw[it] = a * 1000.f; tr[it] = b * 1000.f;
        }
        for (int i = 0; i < 9; ++i) for (int j = i + 1; j < 9; ++j) { if (tw[j] < tw[i]) { float t = tw[i]; tw[i] = tw[j]; tw[j] = t; } if (tr[j] < tr[i]) { float t = tr[i]; tr[i] = tr[j]; tr[j] = t; } }
        const double bytes = (double)live * seg_floats * 4;
        printf("write %-3s read %-3s order %-7s %d seg: write %7.1f us %6.0f GB/s   read %7.1f us %6.0f GB/s\n", wnt ? "nt" : "st", rnt ? "nt" : "ld", rev ? "reverse" : "forward", live,
               tw[4], bytes / tw[4] * 1e-3, tr[4], bytes / tr[4] * 1e-3);
    }
    return 0;
}
