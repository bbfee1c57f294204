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
    const int n = gnms_count(counts, b, N);
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

// masked groups, default path, ONE launch with two kinds of workgroups (hard sort, unsorted output: position q = rank):
//   blockIdx.x <  elem_blocks : one thread per rank: gx = dL/dprob where the clamp let it through; dL/ds = gx for every box
//                               in a group, 0 for boxes in no group -- except the heads of multi-member groups, which belong to
//   blockIdx.x >= elem_blocks : one WAVE per multi-member group (hlist): dL/ds_head = gx_head - sum_i P_i gx_i.  All lanes fetch
//                               the members' products in parallel (gx recomputed from dL/dprob, so the two kinds of workgroups
//                               do not depend on each other); the sum runs SEQUENTIALLY in member order (v_readlane + v_sub):
//                               deterministic and bit-identical to the left-to-right matmul row of the reference.
__device__ __forceinline__ float bwd_gx_of(const float* __restrict__ grad_prob_img, const ImgPtrs& I, int k) {
    const float pre = I.pre[k];
    return (pre >= 0.0f && pre <= 1.0f) ? grad_prob_img[k] : 0.0f;    // clamp passes the gradient at the bounds only
}

__global__ __launch_bounds__(256) void bwd_masked_fused_kernel(const float* __restrict__ grad_prob, int N, const int* __restrict__ counts,
                                                               gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ grad_scores,
                                                               int elem_blocks) {
    const int b = blockIdx.y;
    const int n = gnms_count(counts, b, N);
    ImgPtrs I = img_ptrs(ws, L, b);
    float* gs = grad_scores + (size_t)b * N;
    const float* gp = grad_prob + (size_t)b * N;
    if ((int)blockIdx.x >= elem_blocks) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int nheads = I.misc[1];
        const int nhb = (int)gridDim.x - elem_blocks;
        for (int hi = ((int)blockIdx.x - elem_blocks) * 4 + wave; hi < nheads; hi += nhb * 4) {
            const int k = I.hlist[hi];
            const int hstart = I.gstart[k], hlen = I.glen[k];
            float acc = bwd_gx_of(gp, I, k);
            for (int base = 1; base < hlen; base += 64) {
                const int t = base + lane;
                float prod = 0.0f;
                if (t < hlen) {
                    const int mk = I.gsorted[hstart + t];
                    prod = I.plead[mk] * bwd_gx_of(gp, I, mk);
                }
                const int cnt = min(64, hlen - base);
                for (int u = 0; u < cnt; ++u) acc -= __int_as_float(__builtin_amdgcn_readlane(__float_as_int(prod), u));
            }
            if (lane == 0) gs[I.order[k]] = acc;
        }
        return;
    }
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    if (k >= n) { gs[k] = 0.0f; I.gx[k] = 0.0f; return; }
    const float g = bwd_gx_of(gp, I, k);
    I.gx[k] = g;
    const int h = I.head[k];
    if (h == k && I.glen[k] > 1 && I.misc[1] > 0) return;            // written by the head workgroups (hlist holds exactly these)
    gs[I.order[k]] = (h >= 0) ? g : 0.0f;
}

__global__ __launch_bounds__(256) void bwd_masked_kernel(int N, const int* __restrict__ counts, gnms_params P, char* ws, gnms_ws_layout L,
                                                         float* __restrict__ grad_scores) {
    const int b = blockIdx.y;
    const int n = gnms_count(counts, b, N);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    ImgPtrs I = img_ptrs(ws, L, b);
    float* gs = grad_scores + (size_t)b * N;
    if (k >= n) { gs[k] = 0.0f; return; }
    const int c = I.order[k];
    gs[c] = (I.head[k] >= 0) ? I.gx[P.presorted ? c : k] : 0.0f;
}

// masked groups, part 2: one WAVE per multi-member group (hlist).  All lanes fetch the members' products P_i*gx_i in
// parallel (two memory round trips per 64 members); the head's value is then reduced SEQUENTIALLY in member order
// (v_readlane + v_sub, no memory on that path): deterministic, and bit-identical to the left-to-right accumulation of
// the reference's matmul row as restated by the oracle.
__global__ __launch_bounds__(256) void bwd_masked_heads_kernel(int N, gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ grad_scores) {
    const int b = blockIdx.y;
    ImgPtrs I = img_ptrs(ws, L, b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nheads = I.misc[1];
    float* gs = grad_scores + (size_t)b * N;
    for (int hi = blockIdx.x * 4 + wave; hi < nheads; hi += gridDim.x * 4) {
        const int k = I.hlist[hi];
        const int hstart = I.gstart[k], hlen = I.glen[k];
        const int c = I.order[k];
        float acc = I.gx[P.presorted ? c : k];
        for (int base = 1; base < hlen; base += 64) {
            const int t = base + lane;
            float prod = 0.0f;
            if (t < hlen) {
                const int mk = I.gsorted[hstart + t];
                prod = I.plead[mk] * I.gx[P.presorted ? I.order[mk] : mk];
            }
            const int cnt = min(64, hlen - base);
            for (int u = 0; u < cnt; ++u) acc -= __int_as_float(__builtin_amdgcn_readlane(__float_as_int(prod), u));
        }
        if (lane == 0) gs[c] = acc;
    }
}

// sparse part of dL/diou for masked groups (the dense zero fill is a hipMemsetAsync before this kernel)
__global__ __launch_bounds__(256) void bwd_masked_iou_kernel(const float* __restrict__ iou, int N, long ld, const int* __restrict__ counts,
                                                             gnms_params P, char* ws, gnms_ws_layout L, float* __restrict__ grad_iou) {
    const int b = blockIdx.y;
    const int n = gnms_count(counts, b, N);
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
