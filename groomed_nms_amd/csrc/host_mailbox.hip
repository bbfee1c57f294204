// host_mailbox.hip -- the layer's two per-image counts on the host without a device-to-host copy.
//
// The reference returns `valid_boxes_index` / `invalid_boxes_index` as tensors whose LENGTH is data dependent
// (lib/groomed_nms.py:120-127): one host round trip per call is part of its boundary.  A `hipMemcpy` (torch's `.tolist()` / `.item()`)
// pays for that trip with a copy submission, the copy itself and a stream synchronisation -- ~20 us behind ~20 us of kernels at the
// reference's own size (N = 500).  Here the counts are stored into a slot of fine-grained (coherent, host-mapped) pinned memory -- by a
// one-wave kernel behind the layer (gnms_counts_to_host) or by the forward call's own kernels (gnms_host_counts_slot / _wait) -- and the
// host polls: the trip is one PCIe write.
//
// OWNERSHIP (round 6).  One mailbox per device: kSlots slots of 1 KiB, allocated on first use.  A slot has exactly
// one owner at a time: `busy[slot]` is taken with a compare-exchange (a thread starts its search at the slot it used last, so a thread that
// calls in a loop keeps meeting its own slot and never contends) and is given back
//   * by gnms_host_counts_wait on EVERY return path, and only when the device can no longer store into the slot: all 2 B counts have been seen
//     (the kernels store each exactly once), or the stream has been synchronised, or the device has failed;
//   * by gnms_host_counts_release (a caller that took a slot and then did not make / could not make the forward call): it synchronises the stream first;
//   * by gnms_counts_to_host itself before it returns (same rule: the tag has been seen, or the fallback copy has synchronised the stream).
// No free slot (every one is owned: > kSlots calls in flight on the device, or callers that leaked theirs): gnms_host_counts_slot returns
// GNMS_ERR_UNSUPPORTED and the caller takes the plain path (a counts tensor + gnms_counts_to_host, whose own fallback is the copy) -- a
// slot is NEVER handed to a second call while the first may still be written or read.  Word [0] of a slot is the tag the one-wave kernel
// writes last, word [1] the owner's generation (host only; gnms_host_counts_wait refuses a view whose slot is not owned: a second wait on
// the same view, or a view that was never handed out, is GNMS_ERR_INVALID_ARGUMENT instead of somebody else's counts).
//
// POLLING (gnms_poll_backoff, gnms_common.h): ~4 k `pause`s (what a call at the reference's size needs), then `sched_yield` between the
// polls, then sleeps that double from 2 to 128 us -- a wait behind milliseconds of kernels (N = 16384) no longer burns a host core.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>

#include <sched.h>
#include <time.h>

#include "../../include/groomed_nms_hip.h"
#include "gnms_common.h"

bool gnms_poll_backoff::wait() {
    ++n;
    if (n <= 4096) { gnms_cpu_relax(); return (n & 1023) == 0; }
    if (n <= 8192) { sched_yield(); return (n & 1023) == 0; }
    timespec ts{0, (long)sleep_us * 1000};
    nanosleep(&ts, nullptr);
    if (sleep_us < 128) sleep_us *= 2;
    return true;
}

namespace {

constexpr int kSlots = 64, kSlotWords = 256, kMaxImages = (kSlotWords - 2) / 2, kMaxDevices = 64;

struct Mailbox {
    int32_t* host = nullptr;           // kSlots x kSlotWords words, hipHostMallocMapped | hipHostMallocCoherent
    int32_t* dev = nullptr;            // the same memory as the device addresses it
    std::atomic<uint32_t> seq{0};      // tags / generations
    std::atomic<uint32_t> busy[kSlots];// 0: free, else the owner's generation
    std::once_flag once;
    hipError_t err = hipSuccess;
};
Mailbox g_box[kMaxDevices];
std::atomic<int> g_slot_limit{kSlots};     // gnms_test_mailbox_slots: a test makes the mailbox small to reach the no-free-slot path

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
        for (auto& b : M.busy) b.store(0, std::memory_order_relaxed);
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

// takes a free slot for generation `gen` (never 0); -1: every slot is owned
int acquire(Mailbox& M, uint32_t gen) {
    static thread_local int last = -1;
    const int n = g_slot_limit.load(std::memory_order_relaxed);
    int s = last >= 0 && last < n ? last : (int)(gen % (uint32_t)n);
    for (int i = 0; i < n; ++i, s = s + 1 == n ? 0 : s + 1) {
        uint32_t expect = 0;
        if (M.busy[s].compare_exchange_strong(expect, gen, std::memory_order_acquire, std::memory_order_relaxed)) { last = s; return s; }
    }
    return -1;
}
inline void release(Mailbox& M, int slot) { M.busy[slot].store(0, std::memory_order_release); }

// the mailbox and slot a host view points into (any device's); false: not a view this library handed out
bool locate(const int32_t* host_view, Mailbox** Mout, int* slot) {
    for (Mailbox& M : g_box) {
        if (!M.host || host_view < M.host || host_view >= M.host + (size_t)kSlots * kSlotWords) continue;
        const size_t off = (size_t)(host_view - M.host);
        if (off % kSlotWords != 2) return false;
        *Mout = &M;
        *slot = (int)(off / kSlotWords);
        return true;
    }
    return false;
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
    const uint32_t s = M.seq.fetch_add(1, std::memory_order_relaxed) + 1;
    const int32_t tag = (int32_t)(s | 0x40000000u);                      // never 0, the slots' initial content
    const int slot = acquire(M, (uint32_t)tag);
    if (slot < 0) return plain_copy(nvalid, ninvalid, B, host_out, st);  // every slot is owned
    struct Owner { Mailbox& M; int slot; ~Owner() { release(M, slot); } } owner{M, slot};   // every return below has seen the tag or synchronised the stream
    counts_to_host_kernel<<<1, 64, 0, st>>>(nvalid, ninvalid, B, M.dev + (size_t)slot * kSlotWords, tag);
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { gnms_set_error("gnms_counts_to_host: launch failed: %s", hipGetErrorString(e)); return GNMS_ERR_HIP; }   // (nothing was enqueued)
    }
    volatile int32_t* h = M.host + (size_t)slot * kSlotWords;
    const auto t0 = std::chrono::steady_clock::now();
    gnms_poll_backoff bo;
    for (;;) {
        if (__atomic_load_n((const int32_t*)h, __ATOMIC_ACQUIRE) == tag) break;
        if (bo.wait()) {
            // a failed launch upstream never delivers the tag: ask the stream now and then, give up on the mailbox after two seconds
            hipError_t q = hipStreamQuery(st);
            if (q != hipSuccess && q != hipErrorNotReady) {
                gnms_set_error("gnms_counts_to_host: %s", hipGetErrorString(q));
                return GNMS_ERR_HIP;                                    // (a failed device stores nothing any more)
            }
            if (q == hipSuccess && __atomic_load_n((const int32_t*)h, __ATOMIC_ACQUIRE) == tag) break;
            if (q == hipSuccess || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return plain_copy(nvalid, ninvalid, B, host_out, st);
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
    const uint32_t gen = (M->seq.fetch_add(1, std::memory_order_relaxed) + 1) | 0x40000000u;
    const int slot = acquire(*M, gen);
    if (slot < 0) {
        gnms_set_error("gnms_host_counts_slot: all %d slots of device %d are owned (calls in flight, or slots that were never waited for / released)",
                       g_slot_limit.load(), dev);
        return GNMS_ERR_UNSUPPORTED;
    }
    const size_t base = (size_t)slot * kSlotWords;
    M->host[base + 1] = (int32_t)gen;
    for (int i = 0; i < 2 * B; ++i) __atomic_store_n(M->host + base + 2 + i, -1, __ATOMIC_RELAXED);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);                               // the presets are out before any launch that follows
    *device_view = M->dev + base + 2;
    *host_view = M->host + base + 2;
    return GNMS_OK;
}

extern "C" int gnms_host_counts_wait(const int32_t* host_view, int B, int32_t* host_out, void* stream) {
    GNMS_CHECK_ARG(host_view && host_out && B >= 1 && B <= kMaxImages, "gnms_host_counts_wait: bad arguments");
    Mailbox* M = nullptr;
    int slot = -1;
    GNMS_CHECK_ARG(locate(host_view, &M, &slot), "gnms_host_counts_wait: not a view gnms_host_counts_slot handed out");
    const uint32_t gen = (uint32_t)M->host[(size_t)slot * kSlotWords + 1];
    GNMS_CHECK_ARG(gen != 0 && M->busy[slot].load(std::memory_order_acquire) == gen,
                   "gnms_host_counts_wait: the slot is not owned (one wait or release per gnms_host_counts_slot call)");
    struct Owner { Mailbox& M; int slot; ~Owner() { release(M, slot); } } owner{*M, slot};
    hipStream_t st = (hipStream_t)stream;
    const auto t0 = std::chrono::steady_clock::now();
    int i = 0;
    bool synced = false;
    gnms_poll_backoff bo;
    while (i < 2 * B) {
        const int32_t v = __atomic_load_n(host_view + i, __ATOMIC_ACQUIRE);
        if (v >= 0) { host_out[i++] = v; continue; }
        if (synced) {
            gnms_set_error("gnms_host_counts_wait: the stream is idle and count %d was never written (was the slot passed to the forward call?)", i);
            return GNMS_ERR_INVALID_ARGUMENT;
        }
        if (bo.wait()) {
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

extern "C" int gnms_host_counts_release(const int32_t* host_view, void* stream) {
    GNMS_CHECK_ARG(host_view, "gnms_host_counts_release: null view");
    Mailbox* M = nullptr;
    int slot = -1;
    GNMS_CHECK_ARG(locate(host_view, &M, &slot), "gnms_host_counts_release: not a view gnms_host_counts_slot handed out");
    const uint32_t gen = (uint32_t)M->host[(size_t)slot * kSlotWords + 1];
    GNMS_CHECK_ARG(gen != 0 && M->busy[slot].load(std::memory_order_acquire) == gen, "gnms_host_counts_release: the slot is not owned");
    // a forward call that did get enqueued may still store its counts: the slot goes back only behind it
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    release(*M, slot);
    if (e != hipSuccess) { gnms_set_error("gnms_host_counts_release: %s", hipGetErrorString(e)); return GNMS_ERR_HIP; }
    return GNMS_OK;
}

extern "C" int gnms_test_mailbox_slots(int n) {
    const int old = g_slot_limit.load();
    if (n >= 1 && n <= kSlots) g_slot_limit.store(n);
    return old;
}
