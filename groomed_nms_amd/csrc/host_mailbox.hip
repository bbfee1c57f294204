// host_mailbox.hip -- the layer's two per-image counts on the host without a device-to-host copy.
//
// The reference returns `valid_boxes_index` / `invalid_boxes_index` as tensors whose LENGTH is data dependent
// (lib/groomed_nms.py:120-127): one host round trip per call is part of its boundary.  A `hipMemcpy` (torch's `.tolist()` / `.item()`)
// pays for that trip with a copy submission, the copy itself and a stream synchronisation -- ~20 us behind ~20 us of kernels at the
// reference's own size (N = 500).  Here a one-wave kernel, stream-ordered behind the layer, stores the counts and then a call tag
// into a slot of fine-grained (coherent, host-mapped) pinned memory, and the host polls the tag: the trip is one PCIe write.
//
// One mailbox per device (64 slots of 1 KiB, allocated on first use); a call takes the slot of its tag (a per-device counter), so
// calls of different host threads do not meet unless more than 64 of them wait at once -- a call that does not see its tag within
// two seconds synchronises the stream and copies the counts the plain way (also the path for B > kMaxImages).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>

#include "../../include/groomed_nms_hip.h"
#include "gnms_common.h"

namespace {

constexpr int kSlots = 64, kSlotWords = 256, kMaxImages = (kSlotWords - 2) / 2, kMaxDevices = 64;

struct Mailbox {
    int32_t* host = nullptr;           // kSlots x kSlotWords words, hipHostMallocMapped | hipHostMallocCoherent
    int32_t* dev = nullptr;            // the same memory as the device addresses it
    std::atomic<uint32_t> seq{0};
    std::once_flag once;
    hipError_t err = hipSuccess;
};
Mailbox g_box[kMaxDevices];

// one wave: counts -> slot[2 ..], then (release, system scope) the tag -> slot[0]
__global__ __launch_bounds__(64) void counts_to_host_kernel(const int32_t* __restrict__ nvalid, const int32_t* __restrict__ ninvalid, int B,
                                                            int32_t* slot, int32_t tag) {
    for (int i = threadIdx.x; i < 2 * B; i += 64) {
        const int32_t v = i < B ? nvalid[i] : ninvalid[i - B];
        __hip_atomic_store(slot + 2 + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the fence is a wave-wide wait for every lane's stores above (one wave: no barrier needed), then the tag may leave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");               // ("": system scope)
    if (threadIdx.x == 0) __hip_atomic_store(slot, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int plain_copy(const int32_t* nvalid, const int32_t* ninvalid, int B, int32_t* host_out, hipStream_t st) {
    GNMS_CHECK_HIP(hipMemcpyAsync(host_out, nvalid, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
    GNMS_CHECK_HIP(hipMemcpyAsync(host_out + B, ninvalid, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
    GNMS_CHECK_HIP(hipStreamSynchronize(st));
    return GNMS_OK;
}

Mailbox* mailbox(int dev) {
    if (dev < 0 || dev >= kMaxDevices) return nullptr;
    Mailbox& M = g_box[dev];
    std::call_once(M.once, [&M] {
        void* p = nullptr;
        M.err = hipHostMalloc(&p, sizeof(int32_t) * kSlots * kSlotWords, hipHostMallocMapped | hipHostMallocCoherent);
        if (M.err != hipSuccess) return;
        std::memset(p, 0, sizeof(int32_t) * kSlots * kSlotWords);
        void* d = nullptr;
        M.err = hipHostGetDevicePointer(&d, p, 0);
        M.host = (int32_t*)p;
        M.dev = (int32_t*)d;
    });
    return (M.err == hipSuccess && M.dev) ? &M : nullptr;
}

}  // namespace

extern "C" int gnms_counts_to_host(const int32_t* nvalid, const int32_t* ninvalid, int B, int32_t* host_out, void* stream) {
    GNMS_CHECK_ARG(B >= 0 && (B == 0 || (nvalid && ninvalid && host_out)), "gnms_counts_to_host: null pointer");
    if (B == 0) return GNMS_OK;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    GNMS_CHECK_HIP(hipGetDevice(&dev));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    GNMS_CHECK_HIP(hipStreamIsCapturing(st, &cap));
    GNMS_CHECK_ARG(cap == hipStreamCaptureStatusNone, "gnms_counts_to_host: a host round trip cannot be captured into a graph");
    if (B > kMaxImages) return plain_copy(nvalid, ninvalid, B, host_out, st);
    Mailbox* Mp = mailbox(dev);
    if (!Mp) return plain_copy(nvalid, ninvalid, B, host_out, st);
    Mailbox& M = *Mp;
    uint32_t s = M.seq.fetch_add(1, std::memory_order_relaxed) + 1;
    const int32_t tag = (int32_t)(s | 0x40000000u);                      // never 0, the slots' initial content
    const int slot = (int)(s % kSlots);
    counts_to_host_kernel<<<1, 64, 0, st>>>(nvalid, ninvalid, B, M.dev + (size_t)slot * kSlotWords, tag);
    GNMS_CHECK_LAUNCH();
    volatile int32_t* h = M.host + (size_t)slot * kSlotWords;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n((const int32_t*)h, __ATOMIC_ACQUIRE) == tag) break;
        __builtin_ia32_pause();
        if ((spins & 0xfff) == 0xfff) {
            // a failed launch upstream never delivers the tag: ask the stream now and then, give up on the mailbox after two seconds
            hipError_t q = hipStreamQuery(st);
            if (q != hipSuccess && q != hipErrorNotReady) {
                gnms_set_error("gnms_counts_to_host: %s", hipGetErrorString(q));
                return GNMS_ERR_HIP;
            }
            if (q == hipSuccess && __atomic_load_n((const int32_t*)h, __ATOMIC_ACQUIRE) != tag)
                return plain_copy(nvalid, ninvalid, B, host_out, st);   // the stream is done and the slot belongs to somebody else's tag
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return plain_copy(nvalid, ninvalid, B, host_out, st);
        }
    }
    for (int i = 0; i < 2 * B; ++i) host_out[i] = h[2 + i];
    return GNMS_OK;
}

// The same trip without the extra kernel: the forward call's `nvalid` / `ninvalid` arguments may point INTO a slot (the kernels only ever
// store to them, once, at the end of an image's chain), preset to -1 by the host; the host then polls until every word is a count.
extern "C" int gnms_host_counts_slot(int B, int32_t** device_view, const int32_t** host_view) {
    GNMS_CHECK_ARG(B >= 1 && device_view && host_view, "gnms_host_counts_slot: bad arguments");
    if (B > kMaxImages) {
        gnms_set_error("gnms_host_counts_slot: B=%d exceeds %d images per slot", B, kMaxImages);
        return GNMS_ERR_UNSUPPORTED;
    }
    int dev = 0;
    GNMS_CHECK_HIP(hipGetDevice(&dev));
    Mailbox* M = mailbox(dev);
    if (!M) {
        gnms_set_error("gnms_host_counts_slot: no fine-grained pinned memory on device %d", dev);
        return GNMS_ERR_HIP;
    }
    const uint32_t s = M->seq.fetch_add(1, std::memory_order_relaxed) + 1;
    const size_t off = (size_t)(s % kSlots) * kSlotWords + 2;
    for (int i = 0; i < 2 * B; ++i) __atomic_store_n(M->host + off + i, -1, __ATOMIC_RELAXED);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);                               // the presets are out before any launch that follows
    *device_view = M->dev + off;
    *host_view = M->host + off;
    return GNMS_OK;
}

extern "C" int gnms_host_counts_wait(const int32_t* host_view, int B, int32_t* host_out, void* stream) {
    GNMS_CHECK_ARG(host_view && host_out && B >= 1 && B <= kMaxImages, "gnms_host_counts_wait: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const auto t0 = std::chrono::steady_clock::now();
    int i = 0;
    bool synced = false;
    for (unsigned spins = 0; i < 2 * B; ++spins) {
        const int32_t v = __atomic_load_n(host_view + i, __ATOMIC_ACQUIRE);
        if (v >= 0) { host_out[i++] = v; continue; }
        if (synced) {
            gnms_set_error("gnms_host_counts_wait: the stream is idle and count %d was never written (was the slot passed to the forward call?)", i);
            return GNMS_ERR_INVALID_ARGUMENT;
        }
        __builtin_ia32_pause();
        if ((spins & 0xfff) == 0xfff) {
            hipError_t q = hipStreamQuery(st);
            if (q != hipSuccess && q != hipErrorNotReady) {
                gnms_set_error("gnms_host_counts_wait: %s", hipGetErrorString(q));
                return GNMS_ERR_HIP;
            }
            if (q == hipSuccess || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                GNMS_CHECK_HIP(hipStreamSynchronize(st));                  // everything the stream held has been written by now
                synced = true;
                i = 0;
            }
        }
    }
    return GNMS_OK;
}
