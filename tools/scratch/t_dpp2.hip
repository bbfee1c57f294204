#include <hip/hip_runtime.h>
#include <stdio.h>
// OR over all 64 lanes via DPP (row_shr 1,2,4,8; row_bcast:15; row_bcast:31); total lands in lane 63
__device__ __forceinline__ unsigned dpp_or_total(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);   // row_bcast:15 -> rows 1,3
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);   // row_bcast:31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__global__ void k(const unsigned* in, unsigned* out, unsigned* scan) {
    unsigned v = in[threadIdx.x];
    unsigned t = dpp_or_total(v);
    out[threadIdx.x] = t;
    // also export the inclusive scan for inspection
    unsigned s = v;
    s |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)s, 0x111, 0xF, 0xF, true);
    s |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)s, 0x112, 0xF, 0xF, true);
    s |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xF, 0xF, true);
    s |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xF, 0xF, true);
    s |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xA, 0xF, true);
    s |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xC, 0xF, true);
    scan[threadIdx.x] = s;
}
int main() {
    unsigned h[64], *d, *o, *sc; hipMalloc(&d, 256); hipMalloc(&o, 256); hipMalloc(&sc, 256);
    int bad = 0;
    for (int trial = 0; trial < 70; ++trial) {
        unsigned want = 0;
        for (int i = 0; i < 64; ++i) { h[i] = (trial < 64) ? (i == trial ? (1u << (i & 31)) | 0x80000000u : 0u) : (unsigned)rand(); want |= h[i]; }
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice); k<<<1, 64>>>(d, o, sc);
        unsigned r[64], s[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost); hipMemcpy(s, sc, 256, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; ++i) if (r[i] != want) { if (bad < 5) printf("trial %d lane %d got %08x want %08x\n", trial, i, r[i], want); ++bad; }
        unsigned run = 0; for (int i = 0; i < 64; ++i) { run |= h[i]; if (s[i] != run) { if (bad < 10) printf("scan trial %d lane %d got %08x want %08x\n", trial, i, s[i], run); ++bad; } }
    }
    printf("bad=%d\n", bad); return bad != 0;
}
