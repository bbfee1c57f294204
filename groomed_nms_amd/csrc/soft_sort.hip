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
#include <dlfcn.h>
#include <array>
#include <map>
#include <mutex>
#include <hipblaslt/hipblaslt.h>   // (types and enumerators only: the entry points are resolved with dlsym, below)
#include <type_traits>
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
                                                         int M, int Nn, int Kall, long lda, long ldb, long ldd, int accumulate, int kslice,
                                                         long slice_stride) {
    // split K (round 4): slice z multiplies the k range [z kslice, (z + 1) kslice) into its own M x Nn panel, D + z slice_stride
    // (launch_sgemm's scratch; sgemm_reduce_kernel adds the panels in slice order); one slice: the whole K, straight into D
    const int kbeg = (int)blockIdx.z * kslice;
    const int K = min(Kall, kbeg + kslice);
    D += (long)blockIdx.z * slice_stride;
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
    fetch(kbeg);
    for (int k0 = kbeg; k0 < K; k0 += BK) {
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
                if (row < M && col < Nn) {
                    float* d = D + (size_t)row * ldd + col;
                    *d = accumulate ? (*d + acc[ta][tb][r]) : acc[ta][tb][r];         // D += A B (the adjoint's dC accumulation) or D = A B
                }
            }
}

// ------------------------------------------------------------------------------------------------
// The large-matrix GEMM (round 3): 256 x 128 block tile, K step 16, 4 waves each owning a 128 x 64 quadrant = 4 x 2 MFMA 32x32 tiles
// (128 accumulator registers; with the staging and fragment registers a wave stays under 256, so TWO workgroups share a CU and one
// covers the other's barrier).  What the 128 x 128 kernel above leaves on the table (64 % of the fp32 MFMA peak at 4096^3) is traffic
// per MFMA: a wave there reads 4 LDS fragments for every 4 MFMAs and the workgroup 32 KiB from L2 per K tile of 64 MFMAs; here
// 6 fragments feed 8 MFMAs (512 matrix-pipe cycles per K step of 2) and a CU pulls 25 % fewer bytes per flop.
//   * LDS is double buffered (one barrier per K tile): tile t+1 is staged into the other buffer while tile t is multiplied;
//   * the global loads of tile t+2 are issued before the MFMAs of tile t (registers): two tiles of slack for the L2 / HBM latency;
//   * the fragments of K step s+1 are read before the 8 MFMAs of step s issue.
// (A 256 x 256 tile with 256 accumulators per wave was tried first: the compiler reads all accumulators into VGPRs for the epilogue,
// and its register allocator then spills the main loop's prefetch registers -- a load, a wait and a scratch store per 16 bytes.)
// Interior tiles only (M multiple of 256, N of 128, K of 16, 16-byte aligned rows): everything else takes the kernel above.
// D = A B or D += A B; fp32 exact products and sums in MFMA order (fmaf chain along K), like the kernel above.
// ------------------------------------------------------------------------------------------------
constexpr int GM = 256, GN = 128, GK = 16, GLA = 20, GLB = 132;      // LDS row pitches (floats): As[m][k] rows of 16 + 4, Bs[k][n] rows of 128 + 4

// K order inside a tile: MFMA step j multiplies k = j (lanes 0-31) and k = 8 + j (lanes 32-63) -- any pairing that covers the 16 k of a
// tile once is a valid order of the sum, and this one lets a lane take its eight A values of a row as TWO 16-byte LDS reads from a
// row-major copy of the A tile, which in turn is staged with plain 16-byte LDS writes of the 16-byte global loads: no element of a
// loaded vector is ever moved (with a k-major copy the compiler shuffled the components right behind the loads and waited for them
// there: the prefetch of tile t+2 stalled every tile).
// Split K (round 4b): slice z = blockIdx.z multiplies the k range [z K, (z + 1) K) into its own panel D + z slice_stride (launch_sgemm's
// scratch, summed in slice order by sgemm_reduce_kernel) -- for the sizes whose tiles alone leave the machine empty (1024^3: 32 tiles).
__global__ __launch_bounds__(256, 2) void sgemm_mfma_big_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ D,
                                                                int K, long lda, long ldb, long ldd, int accumulate, long slice_stride, int dephase = 0) {
    // (round 6 experiment, gnms_profile_sgemm variant 3: the second half of the grid -- the CUs' second workgroups -- starts half a K tile late, so
    // that the two workgroups of a CU do not sit in their barriers at the same time)
    if (dephase && ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2 >= (long)gridDim.x * gridDim.y) __builtin_amdgcn_s_sleep(32);
    A += (long)blockIdx.z * K;                                        // (K = the slice's length: a multiple of 32)
    Bm += (long)blockIdx.z * K * ldb;
    D += (long)blockIdx.z * slice_stride;
    __shared__ __attribute__((aligned(16))) float As[2][GM][GLA];     // As[buf][m][k]
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][GLB];     // Bs[buf][k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long m0 = (long)blockIdx.y * GM, n0 = (long)blockIdx.x * GN;
    const int wm = (wave >> 1) * 128, wn = (wave & 1) * 64;
    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // global -> registers: A: row tid of the tile, its 16 k values (4 float4); B: k rows tid / 32 and + 8, columns 4 (tid % 32) .. (2 float4)
    const float* ap = A + (m0 + tid) * lda;
    const float* bp = Bm + (long)(tid >> 5) * ldb + n0 + 4 * (tid & 31);
    float4 ra0, ra1, ra2, ra3, rb0, rb1;
    auto fetch = [&](int k0) {
        ra0 = *reinterpret_cast<const float4*>(ap + k0);
        ra1 = *reinterpret_cast<const float4*>(ap + k0 + 4);
        ra2 = *reinterpret_cast<const float4*>(ap + k0 + 8);
        ra3 = *reinterpret_cast<const float4*>(ap + k0 + 12);
        rb0 = *reinterpret_cast<const float4*>(bp + (long)k0 * ldb);
        rb1 = *reinterpret_cast<const float4*>(bp + (long)(k0 + 8) * ldb);
    };
    const int kh = lane >> 5, l31 = lane & 31;
    auto tile = [&](auto bufc, int t, int nk) {
        constexpr int buf = decltype(bufc)::value;
        // tile t+1 (in registers since the previous tile) goes to the other LDS buffer; tile t+2 is requested
        if (t + 1 < nk) {
            float4* arow = reinterpret_cast<float4*>(&As[buf ^ 1][tid][0]);
            arow[0] = ra0; arow[1] = ra1; arow[2] = ra2; arow[3] = ra3;
            *reinterpret_cast<float4*>(&Bs[buf ^ 1][tid >> 5][4 * (tid & 31)]) = rb0;
            *reinterpret_cast<float4*>(&Bs[buf ^ 1][(tid >> 5) + 8][4 * (tid & 31)]) = rb1;
        }
        if (t + 2 < nk) fetch((t + 2) * GK);
        // the lane's eight k values (k = 8 kh .. 8 kh + 7) of its row in each of the four row tiles: two 16-byte reads each
        float4 fa[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4* arow = reinterpret_cast<const float4*>(&As[buf][wm + 32 * u + l31][8 * kh]);
            fa[u][0] = arow[0]; fa[u][1] = arow[1];
        }
        auto a_of = [&](int u, int j) { const float4 v = fa[u][j >> 2]; return (j & 3) == 0 ? v.x : (j & 3) == 1 ? v.y : (j & 3) == 2 ? v.z : v.w; };
        float fb[2][2];
        fb[0][0] = Bs[buf][8 * kh][wn + l31]; fb[0][1] = Bs[buf][8 * kh][wn + 32 + l31];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cur = j & 1;
            if (j + 1 < 8) { fb[cur ^ 1][0] = Bs[buf][8 * kh + j + 1][wn + l31]; fb[cur ^ 1][1] = Bs[buf][8 * kh + j + 1][wn + 32 + l31]; }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_of(ta, j), fb[cur][tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();                                              // everyone is done with `buf`; buffer buf ^ 1 is complete
    };
    const int nk = K / GK;                                            // even (K % 32 == 0): tiles in pairs, the LDS buffer of every access a constant
    fetch(0);
    {
        float4* arow = reinterpret_cast<float4*>(&As[0][tid][0]);
        arow[0] = ra0; arow[1] = ra1; arow[2] = ra2; arow[3] = ra3;
        *reinterpret_cast<float4*>(&Bs[0][tid >> 5][4 * (tid & 31)]) = rb0;
        *reinterpret_cast<float4*>(&Bs[0][(tid >> 5) + 8][4 * (tid & 31)]) = rb1;
    }
    fetch(GK);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < nk; t += 2) {
        tile(std::integral_constant<int, 0>{}, t, nk);
        tile(std::integral_constant<int, 1>{}, t + 1, nk);
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* dbase = D + (m0 + wm + 4 * kh) * ldd + n0 + wn + l31;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            float* dt = dbase + (long)(ta * 32) * ldd + tb * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* d = dt + (long)((r & 3) + 8 * (r >> 2)) * ldd;
                *d = accumulate ? (*d + acc[ta][tb][r]) : acc[ta][tb][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Adjoint of soft_sort (the reference differentiates lib/groomed_nms.py:145-164 with autograd).  With
//   A[i][j] = -|s_j - shat_i|,  E = exp((A - rowmax A) / T),  Z[i] = sum_j E[i][j] + 1e-3,  C[i][j] = E[i][j] / Z[j]  (:155),
//   soft_scores = C s,  soft_matrix = C M
// and upstream gradients g_soft, g_C, g_mat:
//   dC      = g_C + g_soft s^T + g_mat M^T                       (MFMA GEMM, accumulating)
//   dM      = C^T g_mat                                          (MFMA GEMM)
//   dZ[j]   = -(sum_i dC[i][j] C[i][j]) / Z[j]                   (column pass; the same pass yields (C^T g_soft)[j])
//   dE[i][j]= dC[i][j] / Z[j] + dZ[i];   dArg = dE E / T          (row pass: E recomputed with the forward's own expression, bit for bit)
//   dA      = dArg, minus the row sum of dArg at the row's arg max (the rowmax term)
//   ds[j]   = (C^T g_soft)[j] - sum_i dA[i][j] sign(s_j - shat_i) + dshat[rank of j],   dshat[i] = sum_j dA[i][j] sign(s_j - shat_i)
// Deterministic: column sums walk the rows in order, no atomics.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int rows, int cols, long ld_in, float* __restrict__ out,
                                                        long ld_out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads, 4 passes
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + ty + 8 * p, c = c0 + tx;
        tile[ty + 8 * p][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = c0 + ty + 8 * p, r = r0 + tx;                  // out[c][r] = in[r][c]
        if (c < cols && r < rows) out[(size_t)c * ld_out + r] = tile[tx][ty + 8 * p];
    }
}

// dC[i][j] = g_C[i][j] + g_soft[i] * s[j]   (either gradient may be absent)
__global__ __launch_bounds__(256) void softsort_bwd_dc_kernel(const float* __restrict__ g_C, const float* __restrict__ g_soft,
                                                              const float* __restrict__ scores, int N, float* __restrict__ dC) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float gs = g_soft ? g_soft[i] : 0.0f;
    dC[(size_t)i * N + j] = (g_C ? g_C[(size_t)i * N + j] : 0.0f) + gs * scores[j];
}

// per column j (one thread each, rows walked in order: coalesced and deterministic):
//   dZ[j] = -(sum_i dC[i][j] C[i][j]) / Z[j],   base[j] = sum_i C[i][j] g_soft[i]
__global__ __launch_bounds__(256) void softsort_bwd_cols1_kernel(const float* __restrict__ dC, const float* __restrict__ C,
                                                                 const float* __restrict__ g_soft, const float* __restrict__ Z, int N,
                                                                 float* __restrict__ dZ, float* __restrict__ base) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    float s1 = 0.0f, s2 = 0.0f;
    for (int i = 0; i < N; ++i) {
        const float c = C[(size_t)i * N + j];
        s1 += dC[(size_t)i * N + j] * c;
        if (g_soft) s2 += c * g_soft[i];
    }
    dZ[j] = -(s1 / Z[j]);
    base[j] = s2;
}

// one workgroup per row i: dA row; writes  -dA[i][j] * sign(s_j - shat_i)  over dC[i][j] and dshat[i]
__global__ __launch_bounds__(256) void softsort_bwd_rows_kernel(const float* __restrict__ scores, int N, float T, char* ws, gnms_ws_layout L,
                                                                const float* __restrict__ Z, const float* __restrict__ dZ,
                                                                float* __restrict__ dC, float* __restrict__ dshat) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int i = blockIdx.x;
    ImgPtrs I = img_ptrs(ws, L, 0);
    const float shat = I.sscore[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // row maximum of A and its first position (the forward's softsort_rows_kernel expression)
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float a = -fabsf(scores[j] - shat);
        if (a > mx) { mx = a; am = j; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float om = __shfl_xor(mx, off, 64);
        const int oa = __shfl_xor(am, off, 64);
        if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    if (lane == 0) { red[wave] = mx; redi[wave] = am; }
    __syncthreads();
    mx = red[0]; am = redi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (red[w] > mx || (red[w] == mx && redi[w] < am)) { mx = red[w]; am = redi[w]; }
    __syncthreads();
    const float dz = dZ[i];
    float* row = dC + (size_t)i * N;
    float rs = 0.0f;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float e = expf((-fabsf(scores[j] - shat) - mx) / T);
        const float darg = ((row[j] / Z[j] + dz) * e) / T;
        row[j] = darg;
        rs += darg;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rs += __shfl_xor(rs, off, 64);
    if (lane == 0) red[wave] = rs;
    __syncthreads();
    rs = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    float dsh = 0.0f;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float da = row[j] - ((j == am) ? rs : 0.0f);
        const float d = scores[j] - shat;
        const float sg = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
        dsh += da * sg;
        row[j] = -(da * sg);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dsh += __shfl_xor(dsh, off, 64);
    if (lane == 0) red[wave] = dsh;
    __syncthreads();
    if (threadIdx.x == 0) dshat[i] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ds[j] = base[j] + sum_i M2[i][j] + dshat[rank of j]
__global__ __launch_bounds__(256) void softsort_bwd_cols2_kernel(const float* __restrict__ M2, const float* __restrict__ base,
                                                                 const float* __restrict__ dshat, int N, char* ws, gnms_ws_layout L,
                                                                 float* __restrict__ d_scores) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    ImgPtrs I = img_ptrs(ws, L, 0);
    float acc = 0.0f;
    for (int i = 0; i < N; ++i) acc += M2[(size_t)i * N + j];
    d_scores[j] = (base[j] + acc) + dshat[I.rankof[j]];
}

// D (+)= sum over the S split-K panels, slice order
__global__ __launch_bounds__(256) void sgemm_reduce_kernel(const float* __restrict__ part, int S, int M, int N, float* __restrict__ D, long ldd,
                                                           int accumulate) {
    const long total = (long)M * N;
    if ((N % 4 == 0) && (ldd % 4 == 0) && ((uintptr_t)D % 16 == 0)) {                  // 16-byte lanes (the panels are dense M x N: aligned)
        const float4* p4 = reinterpret_cast<const float4*>(part);
        const long tot4 = total / 4;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot4; e += (long)gridDim.x * 256) {
            float4 acc = p4[e];
            for (int z = 1; z < S; ++z) { const float4 v = p4[(long)z * tot4 + e]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            float4* d = reinterpret_cast<float4*>(D + (e * 4 / N) * ldd + (e * 4 % N));
            if (accumulate) { const float4 o = *d; acc.x = o.x + acc.x; acc.y = o.y + acc.y; acc.z = o.z + acc.z; acc.w = o.w + acc.w; }
            *d = acc;
        }
        return;
    }
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        float acc = part[e];
        for (int z = 1; z < S; ++z) acc += part[(long)z * total + e];
        float* d = D + (e / N) * ldd + (e % N);
        *d = accumulate ? (*d + acc) : acc;
    }
}

// ------------------------------------------------------------------------------------------------
// The PLAIN product on the vendor libraries (round 6).  C @ iou of the soft sort is a plain fp32 GEMM with nothing to fuse into it, and the
// libraries' tuned assembly kernels for gfx950 stay ahead of the kernels above at most sizes (profiles/r06_sgemm_variants.txt, TF, same box:
// 512^3 own 12.5 / rocBLAS 33.8 / torch.matmul 14.4; 1024^3 54.6 / 106.0 / 99.2; 2048^3 104 / 119 / 139; 4096^3 140 / 139 / 149; 8192^3
// 143 / 140 / 153 -- torch.matmul is hipBLASLt; all of them the same v_mfma_f32_32x32x2_f32, exact fp32 products).  So from 512^3 on the
// product goes to rocBLAS, from 2048^3 on to hipBLASLt, when the library can be found; the hand-written kernels stay for everything else
// (small or odd shapes, a stream that is being captured, a process without the libraries) and behind gnms_profile_sgemm for comparison.
// Resolved with dlopen / dlsym at first use -- libgroomed_nms_hip.so has no link-time dependency on it; inside a PyTorch process the name
// resolves to the copy PyTorch has already mapped.  One handle per device, created once; atomics off (deterministic sums).
// ------------------------------------------------------------------------------------------------
struct RocblasApi {
    typedef int (*create_t)(void**);
    typedef int (*set_stream_t)(void*, hipStream_t);
    typedef int (*set_atomics_t)(void*, int);
    typedef int (*sgemm_t)(void*, int, int, int, int, int, const float*, const float*, int, const float*, int, const float*, float*, int);
    create_t create = nullptr;
    set_stream_t set_stream = nullptr;
    set_atomics_t set_atomics = nullptr;
    sgemm_t sgemm = nullptr;
    void* handle[64] = {};
    std::mutex mu;
    bool tried = false, ok = false;
};
RocblasApi g_rocblas;
// true when the product was enqueued on the library
bool rocblas_sgemm_rowmajor(const float* A, const float* B, float* D, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldd, int accumulate,
                            hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;   // (the library may allocate a workspace)
    if (lda > 0x7fffffff || ldb > 0x7fffffff || ldd > 0x7fffffff) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    RocblasApi& R = g_rocblas;
    std::lock_guard<std::mutex> lock(R.mu);
    if (!R.tried) {
        R.tried = true;
        void* h = nullptr;
        for (const char* name : {"librocblas.so", "librocblas.so.5", "librocblas.so.4", "/opt/rocm/lib/librocblas.so"}) {
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        }
        if (h) {
            R.create = (RocblasApi::create_t)dlsym(h, "rocblas_create_handle");
            R.set_stream = (RocblasApi::set_stream_t)dlsym(h, "rocblas_set_stream");
            R.set_atomics = (RocblasApi::set_atomics_t)dlsym(h, "rocblas_set_atomics_mode");
            R.sgemm = (RocblasApi::sgemm_t)dlsym(h, "rocblas_sgemm");
            R.ok = R.create && R.set_stream && R.sgemm;
        }
    }
    if (!R.ok) return false;
    if (!R.handle[dev]) {
        void* hd = nullptr;
        if (R.create(&hd) != 0 || !hd) { R.ok = false; return false; }
        if (R.set_atomics) (void)R.set_atomics(hd, 0);               // rocblas_atomics_not_allowed
        R.handle[dev] = hd;
    }
    if (R.set_stream(R.handle[dev], st) != 0) return false;
    // row-major D = A B  ==  column-major D^T = B^T A^T: (N x M) = (N x K)(K x M), operands swapped, no transposes
    const float alpha = 1.0f, beta = accumulate ? 1.0f : 0.0f;
    return R.sgemm(R.handle[dev], 111, 111, N, M, K, &alpha, B, (int)ldb, A, (int)lda, &beta, D, (int)ldd) == 0;   // 111 = rocblas_operation_none
}

// hipBLASLt, the library behind torch.matmul on this platform: ahead of rocBLAS's own sgemm and of the kernels here from 2048^3 on
// (r06_sgemm_variants.txt).  The copy that matches the headers this file was compiled against is loaded by its path (PyTorch maps an older one
// under the bare name; the two live side by side), one handle and one 32-MiB workspace per device, the heuristic's first algorithm per shape
// remembered.  Row-major D = A B as the column-major product D^T = B^T A^T.
struct LtApi {
    decltype(&hipblasLtCreate) create = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
    decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
    decltype(&hipblasLtMatmul) matmul = nullptr;
    decltype(&hipblasLtMatmulDescDestroy) desc_destroy = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy) layout_destroy = nullptr;
    struct Plan { hipblasLtMatmulDesc_t desc; hipblasLtMatrixLayout_t a, b, d; hipblasLtMatmulAlgo_t algo; size_t ws; };
    hipblasLtHandle_t handle[64] = {};
    void* wsp[64] = {};
    hipblasLtMatmulPreference_t pref = nullptr;
    std::map<std::array<int64_t, 7>, Plan> plans;
    std::mutex mu;
    bool tried = false, ok = false;
};
LtApi g_lt;
constexpr size_t kLtWorkspace = (size_t)32 << 20;
bool hipblaslt_sgemm_rowmajor(const float* A, const float* B, float* D, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldd, int accumulate,
                              hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    LtApi& R = g_lt;
    std::lock_guard<std::mutex> lock(R.mu);
    if (!R.tried) {
        R.tried = true;
        void* h = nullptr;
        for (const char* name : {"/opt/rocm/lib/libhipblaslt.so.1", "libhipblaslt.so.1"}) {
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        }
        if (h) {
            R.create = (decltype(R.create))dlsym(h, "hipblasLtCreate");
            R.layout_create = (decltype(R.layout_create))dlsym(h, "hipblasLtMatrixLayoutCreate");
            R.desc_create = (decltype(R.desc_create))dlsym(h, "hipblasLtMatmulDescCreate");
            R.pref_create = (decltype(R.pref_create))dlsym(h, "hipblasLtMatmulPreferenceCreate");
            R.pref_set = (decltype(R.pref_set))dlsym(h, "hipblasLtMatmulPreferenceSetAttribute");
            R.heuristic = (decltype(R.heuristic))dlsym(h, "hipblasLtMatmulAlgoGetHeuristic");
            R.matmul = (decltype(R.matmul))dlsym(h, "hipblasLtMatmul");
            R.desc_destroy = (decltype(R.desc_destroy))dlsym(h, "hipblasLtMatmulDescDestroy");
            R.layout_destroy = (decltype(R.layout_destroy))dlsym(h, "hipblasLtMatrixLayoutDestroy");
            R.ok = R.create && R.layout_create && R.desc_create && R.pref_create && R.pref_set && R.heuristic && R.matmul;
            if (R.ok) {
                const uint64_t wsz = kLtWorkspace;
                R.ok = R.pref_create(&R.pref) == HIPBLAS_STATUS_SUCCESS &&
                       R.pref_set(R.pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)) == HIPBLAS_STATUS_SUCCESS;
            }
        }
    }
    if (!R.ok) return false;
    if (!R.handle[dev]) {
        hipblasLtHandle_t hd = nullptr;
        void* w = nullptr;
        if (R.create(&hd) != HIPBLAS_STATUS_SUCCESS || !hd) { R.ok = false; return false; }
        if (hipMalloc(&w, kLtWorkspace) != hipSuccess) { R.ok = false; return false; }
        R.handle[dev] = hd;
        R.wsp[dev] = w;
    }
    const std::array<int64_t, 7> key = {(int64_t)dev, M, N, K, lda, ldb, ldd};
    auto it = R.plans.find(key);
    if (it == R.plans.end()) {
        if (R.plans.size() >= 256) {                                  // (a caller that walks through many shapes: the cache starts over instead of growing)
            for (auto& kv : R.plans) {
                if (R.desc_destroy) (void)R.desc_destroy(kv.second.desc);
                if (R.layout_destroy) { (void)R.layout_destroy(kv.second.a); (void)R.layout_destroy(kv.second.b); (void)R.layout_destroy(kv.second.d); }
            }
            R.plans.clear();
        }
        LtApi::Plan P{};
        // column-major: "A" = B^T as stored (N x K, ld ldb), "B" = A^T as stored (K x M, ld lda), C = D = D^T as stored (N x M, ld ldd)
        if (R.desc_create(&P.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS ||
            R.layout_create(&P.a, HIP_R_32F, (uint64_t)N, (uint64_t)K, ldb) != HIPBLAS_STATUS_SUCCESS ||
            R.layout_create(&P.b, HIP_R_32F, (uint64_t)K, (uint64_t)M, lda) != HIPBLAS_STATUS_SUCCESS ||
            R.layout_create(&P.d, HIP_R_32F, (uint64_t)N, (uint64_t)M, ldd) != HIPBLAS_STATUS_SUCCESS) return false;
        hipblasLtMatmulHeuristicResult_t res[1];
        int got = 0;
        if (R.heuristic(R.handle[dev], P.desc, P.a, P.b, P.d, P.d, R.pref, 1, res, &got) != HIPBLAS_STATUS_SUCCESS || got < 1 ||
            res[0].state != HIPBLAS_STATUS_SUCCESS || res[0].workspaceSize > kLtWorkspace) return false;
        P.algo = res[0].algo;
        P.ws = res[0].workspaceSize;
        it = R.plans.emplace(key, P).first;
    }
    const LtApi::Plan& P = it->second;
    const float alpha = 1.0f, beta = accumulate ? 1.0f : 0.0f;
    return R.matmul(R.handle[dev], P.desc, &alpha, B, P.a, A, P.b, &beta, D, P.d, D, P.d, &P.algo, R.wsp[dev], kLtWorkspace, st) == HIPBLAS_STATUS_SUCCESS;
}

// variant: 0 = the product path (hipBLASLt from 2048^3 on, rocBLAS from 512^3 on, else -- and whenever a library is missing -- the kernels
// here); 1 = the kernels here only; 2 = rocBLAS only, 4 = hipBLASLt only (an error if it is not there); 3 = the kernels here, the large one
// de-phased (an experiment, see sgemm_mfma_big_kernel)
int launch_sgemm(const float* A, const float* B, float* D, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldd, int accumulate,
                 hipStream_t st, int variant = 0) {
    if (M == 0 || N == 0) return GNMS_OK;
    if (variant == 4 || (variant == 0 && M >= 2048 && N >= 2048 && K >= 2048)) {
        if (hipblaslt_sgemm_rowmajor(A, B, D, M, N, K, lda, ldb, ldd, accumulate, st)) return GNMS_OK;
        if (variant == 4) { gnms_set_error("gnms_profile_sgemm: hipBLASLt is not available"); return GNMS_ERR_UNSUPPORTED; }
    }
    if (variant == 2 || (variant == 0 && M >= 512 && N >= 512 && K >= 512)) {
        if (rocblas_sgemm_rowmajor(A, B, D, M, N, K, lda, ldb, ldd, accumulate, st)) return GNMS_OK;
        if (variant == 2) { gnms_set_error("gnms_profile_sgemm: rocBLAS is not available"); return GNMS_ERR_UNSUPPORTED; }
    }
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0);
    if (aligned && M % GM == 0 && N % GN == 0 && K % (2 * GK) == 0 && K >= 2 * GK) {
        const long big_tiles = (long)(M / GM) * (N / GN);
        if (big_tiles >= 256) {
            sgemm_mfma_big_kernel<<<dim3(N / GN, M / GM), 256, 0, st>>>(A, B, D, K, (long)lda, (long)ldb, (long)ldd, accumulate, 0L, variant == 3 ? 1 : 0);
            GNMS_CHECK_LAUNCH();
            return GNMS_OK;
        }
        // the same kernel over S slices of K when its tiles alone do not fill the machine: S doubles until there are two workgroups per CU
        // (the kernel is built for two per CU), a slice keeps >= 128 k and a multiple of 32.  1024^3: 32 tiles x 8 slices; 2048^3: 128 x 4.
        const int cus = gnms_device_cu_count();
        int S = 1;
        while (S < 16 && big_tiles * S < 2L * cus && K % (2 * S * 2 * GK) == 0 && K / (2 * S) >= 128) S *= 2;
        if (S > 1 && big_tiles * S >= cus) {
            const int kslice = K / S;
            gnms_async_buffer part;
            GNMS_CHECK_HIP(part.alloc((size_t)S * M * N * sizeof(float), st));
            sgemm_mfma_big_kernel<<<dim3(N / GN, M / GM, S), 256, 0, st>>>(A, B, part.as<float>(), kslice, (long)lda, (long)ldb, (long)N, 0, (long)M * N);
            GNMS_CHECK_LAUNCH();
            const long total = (long)M * N;
            sgemm_reduce_kernel<<<(unsigned)std::min<long>((total / 4 + 255) / 256, 8192), 256, 0, st>>>(part.as<float>(), S, M, N, D, (long)ldd, accumulate);
            GNMS_CHECK_LAUNCH();
            return GNMS_OK;
        }
    }
    dim3 grid(gnms_div_up(N, BN), gnms_div_up(M, BM));
    // SPLIT K when the tiles alone leave the machine empty (soft sort's own sizes: N <= 500 boxes is 16 tiles on 256 CUs; 1024^3: 64):
    // S slices of K, each a grid of its own writing an M x N panel of a stream-ordered scratch buffer, then one pass adds the panels
    // in slice order -- deterministic like the unsplit sum, but a different association of it (S partial fmaf chains).  S doubles
    // until the launch has two workgroups per CU, a slice keeps at least 64 k.
    const long tiles = (long)grid.x * grid.y;
    const int cus = gnms_device_cu_count();
    int S = 1;
    while (S < 16 && tiles * S < 2L * cus && K / (2 * S) >= 64) S *= 2;
    if (S > 1) {
        const int kslice = gnms_div_up(gnms_div_up(K, S), BK) * BK;
        S = gnms_div_up(K, kslice);
        gnms_async_buffer part;
        GNMS_CHECK_HIP(part.alloc((size_t)S * M * N * sizeof(float), st));
        grid.z = (unsigned)S;
        if (aligned) sgemm_mfma_kernel<true><<<grid, 256, 0, st>>>(A, B, part.as<float>(), M, N, K, (long)lda, (long)ldb, (long)N, 0, kslice, (long)M * N);
        else sgemm_mfma_kernel<false><<<grid, 256, 0, st>>>(A, B, part.as<float>(), M, N, K, (long)lda, (long)ldb, (long)N, 0, kslice, (long)M * N);
        GNMS_CHECK_LAUNCH();
        const long total = (long)M * N;
        sgemm_reduce_kernel<<<(unsigned)std::min<long>((total + 255) / 256, 4096), 256, 0, st>>>(part.as<float>(), S, M, N, D, (long)ldd, accumulate);
        GNMS_CHECK_LAUNCH();
        return GNMS_OK;
    }
    if (aligned) sgemm_mfma_kernel<true><<<grid, 256, 0, st>>>(A, B, D, M, N, K, (long)lda, (long)ldb, (long)ldd, accumulate, K, 0L);
    else sgemm_mfma_kernel<false><<<grid, 256, 0, st>>>(A, B, D, M, N, K, (long)lda, (long)ldb, (long)ldd, accumulate, K, 0L);
    GNMS_CHECK_LAUNCH();
    return GNMS_OK;
}

}  // namespace

extern "C" int gnms_sgemm(const float* A, const float* B, float* D, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldd,
                          void* stream) {
    GNMS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "gnms_sgemm: negative size");
    if (M == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(A && B && D, "gnms_sgemm: null pointer");
    return launch_sgemm(A, B, D, M, N, K, lda, ldb, ldd, 0, (hipStream_t)stream);
}

extern "C" int gnms_profile_sgemm(const float* A, const float* B, float* D, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldd,
                                  int variant, void* stream) {
    GNMS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && variant >= 0 && variant <= 4, "gnms_profile_sgemm: bad argument");
    if (M == 0 || N == 0) return GNMS_OK;
    GNMS_CHECK_ARG(A && B && D, "gnms_profile_sgemm: null pointer");
    return launch_sgemm(A, B, D, M, N, K, lda, ldb, ldd, 0, (hipStream_t)stream, variant);
}

extern "C" size_t gnms_soft_sort_backward_scratch_bytes(int N, int K) {
    if (N <= 0) return 0;
    const size_t n = (size_t)N, k = (size_t)(K > 0 ? K : 0);
    return (2 * n * n + k * n + 3 * n) * sizeof(float) + 256;
}

extern "C" int gnms_soft_sort_backward(const float* scores, const float* matrix, int N, int K, int64_t ld, float temperature, const float* C,
                                       const float* g_soft, const float* g_C, const float* g_mat, float* d_scores, float* d_matrix,
                                       void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, void* stream) {
    GNMS_CHECK_ARG(N >= 0 && K >= 0, "gnms_soft_sort_backward: negative size");
    if (N == 0) return GNMS_OK;
    if (N > GNMS_MAX_BOXES) { gnms_set_error("gnms_soft_sort_backward: N=%d exceeds GNMS_MAX_BOXES", N); return GNMS_ERR_UNSUPPORTED; }
    GNMS_CHECK_ARG(scores && C && d_scores && workspace && scratch, "gnms_soft_sort_backward: null pointer");
    const bool with_matrix = matrix && g_mat && K > 0;
    GNMS_CHECK_ARG(!with_matrix || ld >= K, "gnms_soft_sort_backward: ld < K");
    GNMS_CHECK_ARG(!d_matrix || with_matrix, "gnms_soft_sort_backward: d_matrix needs matrix and g_mat");
    const gnms_ws_layout L = gnms_make_layout(N);
    if (workspace_bytes < L.per_image) { gnms_set_error("gnms_soft_sort_backward: workspace too small"); return GNMS_ERR_WORKSPACE; }
    if (scratch_bytes < gnms_soft_sort_backward_scratch_bytes(N, with_matrix ? K : 0)) {
        gnms_set_error("gnms_soft_sort_backward: scratch too small");
        return GNMS_ERR_WORKSPACE;
    }
    GNMS_CHECK_ARG((uintptr_t)scratch % 16 == 0, "gnms_soft_sort_backward: scratch must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const float* Z = img_ptrs(ws, L, 0).xsol;                       // row sums left by gnms_soft_sort
    const size_t n2 = (size_t)N * N;
    float* dC = (float*)scratch;                                    // [N][N]
    float* T1 = dC + n2;                                            // [N][N]: C^T; before that (first K*N floats) matrix^T... kept apart below
    float* MT = T1 + n2;                                            // [K][N]: matrix^T
    float* dZ = MT + (with_matrix ? (size_t)K * N : 0);
    float* base = dZ + N;
    float* dshat = base + N;
    int rc;
    softsort_bwd_dc_kernel<<<dim3(gnms_div_up(N, 256), N), 256, 0, st>>>(g_C, g_soft, scores, N, dC);
    GNMS_CHECK_LAUNCH();
    if (with_matrix) {
        transpose_kernel<<<dim3(gnms_div_up(K, 32), gnms_div_up(N, 32)), 256, 0, st>>>(matrix, N, K, (long)ld, MT, (long)N);   // MT [K][N]
        GNMS_CHECK_LAUNCH();
        if ((rc = launch_sgemm(g_mat, MT, dC, N, N, K, K, N, N, 1, st))) return rc;                 // dC += g_mat M^T
        if (d_matrix) {
            transpose_kernel<<<dim3(gnms_div_up(N, 32), gnms_div_up(N, 32)), 256, 0, st>>>(C, N, N, (long)N, T1, (long)N);      // C^T
            GNMS_CHECK_LAUNCH();
            if ((rc = launch_sgemm(T1, g_mat, d_matrix, N, K, N, N, K, ld, 0, st))) return rc;      // dM = C^T g_mat
        }
    }
    softsort_bwd_cols1_kernel<<<gnms_div_up(N, 256), 256, 0, st>>>(dC, C, g_soft, Z, N, dZ, base);
    GNMS_CHECK_LAUNCH();
    softsort_bwd_rows_kernel<<<N, 256, 0, st>>>(scores, N, temperature, ws, L, Z, dZ, dC, dshat);
    GNMS_CHECK_LAUNCH();
    softsort_bwd_cols2_kernel<<<gnms_div_up(N, 256), 256, 0, st>>>(dC, base, dshat, N, ws, L, d_scores);
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
        sort_scores_kernel<EE><<<1, sort_threads, sort_lds, st>>>(scores, N, nullptr, ws, L, P2, nullptr, nullptr, 0);                     \
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
    if (iou) return launch_sgemm(C, iou, soft_matrix, N, N, N, N, ld, N, 0, st);
    return GNMS_OK;
}
