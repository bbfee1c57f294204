// gnms_prof.h -- per-launch HIP-event timing of the library's HBM-bound launches (gnms_profile_events / gnms_profile_collect).
// Disarmed (the default) a scope costs one relaxed atomic load.  State lives in nms_layer.hip.
#pragma once
#include <hip/hip_runtime.h>

enum { kProfMatrixWrite = 0, kProfMatrixRead = 1, kProfSlots = 2 };

bool gnms_prof_armed();
void gnms_prof_begin(int slot, hipStream_t st);
void gnms_prof_end(int slot, hipStream_t st);

// brackets the launches issued on `st` during its lifetime with one event pair of slot `slot`
struct GnmsProfScope {
    int slot;
    hipStream_t st;
    bool on;
    GnmsProfScope(int s, hipStream_t stream) : slot(s), st(stream), on(gnms_prof_armed()) { if (on) gnms_prof_begin(slot, st); }
    ~GnmsProfScope() { if (on) gnms_prof_end(slot, st); }
};
