// gnms_prof.h -- per-launch timing of the library's HBM-bound launches (gnms_profile_events / gnms_profile_collect).
// Armed, such a launch goes through hipExtLaunchKernel with a start and a stop event: the events then carry the dispatch's own
// begin / end timestamps -- the interval a kernel trace (rocprofv3 --kernel-trace) reports for it -- without any marker packet in
// the stream.  Disarmed (the default) a launch costs one relaxed atomic load more than a plain <<<>>>.  State: nms_layer.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

enum { kProfMatrixWrite = 0, kProfMatrixRead = 1, kProfPlainStream = 2, kProfSlots = 3 };

bool gnms_prof_armed();
// a fresh (start, stop) event pair registered under `slot`; false if events could not be created (the launch then goes unprofiled)
bool gnms_prof_pair(int slot, hipEvent_t* start, hipEvent_t* stop);

template <typename... KArgs, typename... Args>
inline void gnms_launch_prof(int slot, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (gnms_prof_armed() && gnms_prof_pair(slot, &e0, &e1)) {
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, st, e0, e1, 0u, static_cast<KArgs>(args)...);
        return;
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, st, static_cast<KArgs>(args)...);
}
