// nms_backward_kernels.h -- backward of the GrooMeD-NMS layer (gfx950).
//
// The reference has no hand-written backward: autograd differentiates
//   prob = [threshold](clamp(M s, 0, 1))            lib/groomed_nms.py:111-127
// with M assembled from I - P (masked groups, :105), inverse(I_g + P_g) (:107) or inverse(I + P) (:110).
// Closed forms (restated in oracle/gnms_oracle.c):
//   gx_q   = dL/dprob routed to NMS position q, zeroed where the returned tensor was thresholded
//            in place (:115) or where the clamp saturated (torch.clamp passes the gradient AT the bounds)
//   masked : dL/ds_head = gx_head - sum_{i in group, i != head} P_i gx_i ; dL/ds_i = gx_i ;
//            dL/diou[i][head] = -gx_i s_head f'(iou[i][head])
// Everything is gather-form and deterministic (no float atomics): bit-identical run to run.
#pragma once
#include "nms_kernels.h"

namespace gnms {

// gx by NMS position.  One thread per returned element j.
__global__ __launch_bounds__(256) void bwd_gx_kernel(const float* __restrict__ grad_prob, int N, const int* __restrict__ counts,
                                                     gnms_params P, char* ws, gnms_ws_layout L) {
    const int b = blockIdx.y;
    const int n = counts ? counts[b] : N;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    if (j >= n) { I.gx[j] = 0.0f; return; }
    const int q = P.return_sorted_prob ? I.sidx[j] : j;              // prob[j] = r[sorted_indices[j]]  (:117)
    float g = grad_prob[(size_t)b * N + j];
    const float pre = I.pre[q];
    const float r2 = I.r2[q];
    const bool thresholded = P.return_sorted_prob || !P.group_boxes;  // which tensor was returned (:117,:124-127)
    if (thresholded && r2 < P.valid_box_prob_threshold) g = 0.0f;
    if (!(pre >= 0.0f && pre <= 1.0f)) g = 0.0f;
    I.gx[q] = g;
}

// masked groups: one thread per rank.
__global__ __launch_bounds__(256) void bwd_masked_kernel(int N, const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L,
                                                         float* __restrict__ grad_scores) {
    const int b = blockIdx.y;
    const int n = counts ? counts[b] : N;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    float* gs = grad_scores + (size_t)b * N;
    if (k >= n) { gs[k] = 0.0f; return; }
    const int h = I.head[k];
    float g = 0.0f;
    if (h >= 0) {
        const int q = P.presorted ? I.order[k] : k;
        g = I.gx[q];
        if (h == k) {
            const int start = I.gstart[k], len = I.glen[k];
            int t = 1;
            for (; t + 8 <= len; t += 8) {                           // 8 independent gathers in flight, summed in rank order
                float pl[8], gv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int mk = I.gsorted[start + t + u];
                    pl[u] = I.plead[mk];
                    gv[u] = I.gx[P.presorted ? I.order[mk] : mk];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) g -= pl[u] * gv[u];
            }
            for (; t < len; ++t) {
                const int mk = I.gsorted[start + t];
                g -= I.plead[mk] * I.gx[P.presorted ? I.order[mk] : mk];
            }
        }
    }
    gs[I.order[k]] = g;
}

// sparse part of dL/diou for masked groups (the dense zero fill is a hipMemsetAsync before this kernel)
__global__ __launch_bounds__(256) void bwd_masked_iou_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                             gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ grad_iou) {
    const int b = blockIdx.y;
    const int n = counts ? counts[b] : N;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    const int h = I.head[k];
    if (h < 0 || h == k) return;
    const int ck = I.order[k], ch = I.order[h];
    if (P.presorted && !(ch < ck)) return;                           // entry killed by tril
    const int q = P.presorted ? ck : k;
    const size_t e = (size_t)b * N * ld + (size_t)ck * ld + ch;
    const float d = gnms_prune_grad(iou[e], P.nms_threshold, P.temperature, P.pruning_method);
    grad_iou[e] = (-(I.gx[q] * I.sscore[h])) * d;
}

}  // namespace gnms
