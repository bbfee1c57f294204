// soft_sort.hip -- SoftSort relaxation used by differentiable_nms(sorting_method="soft") on gfx950.
//
// Reference: lib/groomed_nms.py:131-165 soft_sort (Prillo & Eisenschlos, ICML 2020):
//   A[i][j] = -|s_j - shat_i|            shat = hard-sorted scores (:145)
//   E[i][j] = exp((A[i][j] - max_j A[i][j]) / T)                  (:149-152)
//   Z[i]    = sum_j E[i][j] + 1e-3                                (:154)
//   C[i][j] = E[i][j] / Z[j]     <- (n,n)/(n,) broadcasts over the LAST axis in the reference (:155); replicated
//   soft_scores = C s (:158) ;  soft_matrix = C iou (:163)
// The C @ iou product is the one dense GEMM on the whole GrooMeD path (2 N^3 flop): it runs on the
// matrix cores with v_mfma_f32_32x32x2_f32 (exact fp32, = an fmaf chain), 128x128x16 LDS tiles,
// one wave per 64x64 quadrant (2x2 MFMA tiles, 64 accumulator registers).
#include "nms_kernels.h"

namespace {

using namespace gnms;
typedef float floatx16 __attribute__((ext_vector_type(16)));

// one workgroup per row i: E row, Z_i
__global__ __launch_bounds__(256) void softsort_rows_kernel(const float* __restrict__ scores, int N, float T, char* ws, gnms_ws_layout L,
                                                            float* __restrict__ C, float* __restrict__ Z) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    ImgPtrs I = img_ptrs(ws, L, 0);
    const float shat = I.sscore[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < N; j += 256) mx = fmaxf(mx, -fabsf(scores[j] - shat));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.0f;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float e = expf((-fabsf(scores[j] - shat) - mx) / T);
        C[(size_t)i * N + j] = e;
        sum += e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) Z[i] = ((red[0] + red[1]) + (red[2] + red[3])) + 1e-3f;
}

// C[i][j] = E[i][j] / Z[j];  soft_scores[i] = sum_j C[i][j] s_j
__global__ __launch_bounds__(256) void softsort_normalize_kernel(const float* __restrict__ scores, int N, const float* __restrict__ Z,
                                                                 float* __restrict__ C, float* __restrict__ soft_scores) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.0f;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float c = C[(size_t)i * N + j] / Z[j];
        C[(size_t)i * N + j] = c;
        acc += c * scores[j];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) soft_scores[i] = (red[0] + red[1]) + (red[2] + red[3]);
}

// D[M x Nn] = A[M x K] * B[K x Nn], fp32, row-major, leading dims lda/ldb/ldd.
// 128x128 block tile, K step 32, 4 waves each owning a 64x64 quadrant = 2x2 MFMA 32x32 tiles (64 accumulator registers).
// The next K tile is fetched from HBM/L2 into registers (16-byte loads) while the current one is multiplied out of LDS
// (register-staged double buffering): v_mfma_f32_32x32x2_f32 is 64 cycles per issue, 16 per K tile and wave pair, which
// covers the ~1 us global latency with two workgroups per CU.
constexpr int BM = 128, BN = 128, BK = 32, LDP = 132;   // LDS row pitch (floats): 128 + 4 keeps 16-B alignment and spreads banks

template <bool ALIGNED>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ D,
                                                         int M, int Nn, int K, long lda, long ldb, long ldd) {
    __shared__ __attribute__((aligned(16))) float As[BK][LDP];   // As[k][m]
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDP];   // Bs[k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // staging registers: A tile 128 rows x 32 k = 1024 float4 -> 4 per thread (row = tid/2 + 0/.., k chunk);  B tile 32 k x 128 n -> 4 per thread
    float4 ra[4], rb[4];
    const int a_row = tid >> 3, a_k4 = (tid & 7) * 4;            // 32 rows x 8 float4 per pass, 4 passes of 32 rows
    const int b_k = tid >> 5, b_n4 = (tid & 31) * 4;             // 8 k rows x 32 float4 per pass, 4 passes of 8 k
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int gm = m0 + a_row + 32 * p, gk = k0 + a_k4;
            if (ALIGNED && gm < M && gk + 3 < K) ra[p] = *reinterpret_cast<const float4*>(A + (size_t)gm * lda + gk);
            else {
                float t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = (gm < M && gk + u < K) ? A[(size_t)gm * lda + gk + u] : 0.0f;
                ra[p] = make_float4(t[0], t[1], t[2], t[3]);
            }
            const int gkb = k0 + b_k + 8 * p, gn = n0 + b_n4;
            if (ALIGNED && gkb < K && gn + 3 < Nn) rb[p] = *reinterpret_cast<const float4*>(Bm + (size_t)gkb * ldb + gn);
            else {
                float t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = (gkb < K && gn + u < Nn) ? Bm[(size_t)gkb * ldb + gn + u] : 0.0f;
                rb[p] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = a_row + 32 * p;
            As[a_k4 + 0][row] = ra[p].x; As[a_k4 + 1][row] = ra[p].y; As[a_k4 + 2][row] = ra[p].z; As[a_k4 + 3][row] = ra[p].w;
            *reinterpret_cast<float4*>(&Bs[b_k + 8 * p][b_n4]) = rb[p];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        stage();
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);                          // in flight while the MFMAs below run
#pragma unroll
        for (int ks = 0; ks < BK; ks += 2) {
            const int kr = ks + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = As[kr][wm + t * 32 + (lane & 31)];
                b[t] = Bs[kr][wn + t * 32 + (lane & 31)];
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = n0 + wn + tb * 32 + (lane & 31);
                if (row < M && col < Nn) D[(size_t)row * ldd + col] = acc[ta][tb][r];
            }
}

}  // namespace

extern "C" int gnms_sgemm(const float* A, const float* B, float* D, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldd,
                          void* stream) {
    GNMS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "gnms_sgemm: negative size");
    if (M == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(A && B && D, "gnms_sgemm: null pointer");
    dim3 grid(gnms_div_up(N, BN), gnms_div_up(M, BM));
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0);
    if (aligned) sgemm_mfma_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(A, B, D, M, N, K, (long)lda, (long)ldb, (long)ldd);
    else sgemm_mfma_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(A, B, D, M, N, K, (long)lda, (long)ldb, (long)ldd);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

extern "C" int gnms_soft_sort(const float* scores, const float* iou, int N, int64_t ld, float temperature, float* C,
                              float* soft_scores, float* soft_matrix, void* workspace, size_t workspace_bytes, void* stream) {
    GNMS_CHECK_ARG(N >= 0, "gnms_soft_sort: negative N");
    if (N == 0) return GNMS_OK;
    if (N > GNMS_MAX_BOXES) { gnms_set_error("gnms_soft_sort: N=%d exceeds GNMS_MAX_BOXES", N); return GNMS_ERR_UNSUPPORTED; }
    GNMS_CHECK_ARG(scores && C && soft_scores && workspace, "gnms_soft_sort: null pointer");
    GNMS_CHECK_ARG((iou == nullptr) == (soft_matrix == nullptr), "gnms_soft_sort: iou and soft_matrix go together");
    GNMS_CHECK_ARG(!iou || ld >= N, "gnms_soft_sort: ld < N");
    const gnms_ws_layout L = gnms_make_layout(N);
    if (workspace_bytes < L.per_image) { gnms_set_error("gnms_soft_sort: workspace too small"); return GNMS_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int P2 = 64;
    while (P2 < N) P2 <<= 1;
    const size_t sort_lds = (size_t)P2 * 8;
    const int sort_threads = P2 <= 1024 ? P2 : 1024;
#define GNMS_SS_SORT(EE)                                                                                                         \
    do {                                                                                                                         \
        if (sort_lds > 64 * 1024)                                                                                                \
            GNMS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_scores_kernel<EE>),                             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));                      \
        sort_scores_kernel<EE><<<1, sort_threads, sort_lds, st>>>(scores, N, nullptr, ws, L, P2, nullptr, nullptr);                     \
    } while (0)
    switch (P2 <= 1024 ? 1 : P2 / 1024) {
        case 1: GNMS_SS_SORT(1); break;
        case 2: GNMS_SS_SORT(2); break;
        case 4: GNMS_SS_SORT(4); break;
        case 8: GNMS_SS_SORT(8); break;
        default: GNMS_SS_SORT(16); break;
    }
#undef GNMS_SS_SORT
    GNMS_CHECK_LAUNCH();
    float* Z = img_ptrs(ws, L, 0).xsol;
    softsort_rows_kernel<<<N, 256, 0, st>>>(scores, N, temperature, ws, L, C, Z);
    GNMS_CHECK_LAUNCH();
    softsort_normalize_kernel<<<N, 256, 0, st>>>(scores, N, Z, C, soft_scores);
    GNMS_CHECK_LAUNCH();
    if (iou) return gnms_sgemm(C, iou, soft_matrix, N, N, N, N, ld, N, stream);
    return GNMS_OK;
}
