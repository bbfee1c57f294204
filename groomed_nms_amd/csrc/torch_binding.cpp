// torch_binding.cpp -- the GrooMeD-NMS layer as a C++ torch::autograd::Function over the C ABI (include/groomed_nms_hip.h).
//
// The reference's boundary for this path is a Python call, differentiable_nms (lib/groomed_nms.py:10), whose gradient comes from
// autograd (:111).  groomed_nms_amd/groomed_nms.py mirrors that boundary; THIS file is what it calls into: outputs, workspace and
// saved state are allocated here, the gnms_* entry is called with torch's current HIP stream, and the backward is a C++ autograd
// node -- no ctypes marshalling (17-argument calls) and no Python autograd.Function on the step's host path, which is what bounds the
// reference's own regime (N <= 500 boxes per image: 0.12-0.145 ms per eager step against 0.04-0.05 ms of GPU time, round 2).
// PyTorch is plumbing here (device memory, the current stream, the autograd graph); all arithmetic is in libgroomed_nms_hip.so.
// Built in-tree by groomed_nms_amd/build.py (torch.utils.cpp_extension, host compiler only: no device code in this file).
#include <torch/extension.h>

// PyTorch-ROCm presents its HIP devices under the device type "cuda"; these are the guard / stream accessors for that device type
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <cstring>

#include "../../include/groomed_nms_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc, const char* what) {
    if (rc == GNMS_OK) return;
    const char* m = gnms_last_error();
    if (rc == GNMS_ERR_UNSUPPORTED && m && std::strstr(m, "not implemented"))
        TORCH_CHECK_NOT_IMPLEMENTED(false, "Pruning method not implemented!");                 // lib/groomed_nms.py:178
    TORCH_CHECK(false, "GNMS: ", what, " failed (", rc, "): ", m ? m : "");
}

// which entry runs the forward pass
enum Mode : int64_t { kMatrixIn = 0, kWithIou2d = 1, kWithIou3d = 2, kFromBoxes = 3 };

using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
inline hipStream_t current_stream(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

inline const void* cptr(const Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
inline void* mptr(Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }

// (tensor, ld): unit column stride and image stride N * ld, copying only if needed (groomed_nms.py::_matrix_layout)
std::pair<Tensor, int64_t> matrix_layout(const Tensor& iou) {
    const int64_t B = iou.size(0), N = iou.size(1);
    if (N == 0) return {iou.contiguous(), 1};
    if (iou.stride(2) == 1 && iou.stride(1) >= N && (B == 1 || iou.stride(0) == N * iou.stride(1))) return {iou, iou.stride(1)};
    return {iou.contiguous(), N};
}

// set by single() around Layer::apply (which runs forward() on the calling thread): the forward call then stores its two counts straight into
// a slot of the library's pinned mailbox (gnms_host_counts_slot) instead of the `cnt` tensor, which stays unwritten
thread_local int32_t* tl_counts_slot = nullptr;

struct Layer : public torch::autograd::Function<Layer> {
    // src: the overlap matrix [B,N,N] (kMatrixIn), 2D boxes [B,N,4] (kWithIou2d, kFromBoxes) or cuboid parameters [B,N,7] (kWithIou3d)
    static variable_list forward(AutogradContext* ctx, const Tensor& scores, const Tensor& src, const c10::optional<Tensor>& counts_,
                                 const c10::optional<Tensor>& iou_out_, int64_t mode, double thr, double temp, double vthr, int64_t prune,
                                 bool sorted_prob, bool group, bool mask, int64_t gsize, bool presorted, bool index_lists) {
        gnms_params P;
        P.nms_threshold = (float)thr; P.temperature = (float)temp; P.valid_box_prob_threshold = (float)vthr;
        P.pruning_method = (int32_t)prune; P.return_sorted_prob = sorted_prob; P.group_boxes = group; P.mask_group_boxes = mask;
        P.group_size = (int32_t)gsize; P.presorted = presorted;
        TORCH_CHECK(scores.is_cuda() && scores.dim() == 2 && scores.scalar_type() == at::kFloat, "GNMS: scores must be a CUDA float [B, N] tensor");
        TORCH_CHECK(src.is_cuda() && src.dim() == 3 && src.scalar_type() == at::kFloat && src.size(0) == scores.size(0) && src.size(1) == scores.size(1),
                    "GNMS: second argument must be a CUDA float [B, N, .] tensor");
        const int64_t B = scores.size(0), N = scores.size(1);
        DeviceGuard guard(scores.device());
        hipStream_t st = current_stream(scores);
        Tensor s = scores.contiguous();
        Tensor counts = counts_.has_value() ? *counts_ : Tensor();
        const auto f32 = s.options();
        Tensor prob = at::empty({B, N}, f32);
        Tensor lists = at::empty({index_lists ? 3 : 1, B, N}, f32.dtype(at::kLong));
        Tensor cnt = at::empty({2, B}, f32.dtype(at::kInt));
        Tensor order = lists[0], valid = index_lists ? lists[1] : Tensor(), invalid = index_lists ? lists[2] : Tensor();
        Tensor nvalid = cnt[0], ninvalid = cnt[1];
        int32_t* const nv_out = tl_counts_slot ? tl_counts_slot : (int32_t*)mptr(nvalid);
        int32_t* const ni_out = tl_counts_slot ? tl_counts_slot + B : (int32_t*)mptr(ninvalid);
        const size_t wsb = gnms_workspace_bytes((int)B, (int)N, &P);
        Tensor ws = at::empty({(int64_t)std::max<size_t>(wsb, 256)}, f32.dtype(at::kByte));
        Tensor kept, iou;                     // kept: what the backward reads besides the scores (matrix / boxes), if anything
        bool bwd_boxes = mode == kFromBoxes;  // the backward takes its overlaps from `kept` = boxes
        int64_t ld = std::max<int64_t>(N, 1);
        if (mode == kMatrixIn) {
            TORCH_CHECK(src.size(2) == N, "GNMS: iou must be [B, N, N]");
            auto ml = matrix_layout(src);
            kept = ml.first;
            ld = ml.second;
            check(gnms_forward((const float*)cptr(s), (const float*)cptr(kept), (int)B, (int)N, ld, (const int32_t*)cptr(counts), &P, (float*)mptr(prob),
                               (int64_t*)mptr(order), (int64_t*)mptr(valid), (int64_t*)mptr(invalid), nv_out, ni_out,
                               mptr(ws), (size_t)ws.numel(), st), "gnms_forward");
        } else if (mode == kFromBoxes) {
            TORCH_CHECK(src.size(2) == 4, "GNMS: boxes must be [B, N, 4]");
            kept = src.contiguous();
            check(gnms_forward_from_boxes((const float*)cptr(kept), (const float*)cptr(s), (int)B, (int)N, (const int32_t*)cptr(counts), &P, (float*)mptr(prob),
                                          (int64_t*)mptr(order), (int64_t*)mptr(valid), (int64_t*)mptr(invalid), nv_out,
                                          ni_out, mptr(ws), (size_t)ws.numel(), st), "gnms_forward_from_boxes");
        } else {
            const bool three_d = mode == kWithIou3d;
            TORCH_CHECK(src.size(2) == (three_d ? 7 : 4), "GNMS: expected [B, N, ", three_d ? 7 : 4, "] as the second argument");
            Tensor boxes = src.contiguous();
            iou = iou_out_.has_value() ? *iou_out_ : at::empty({B, N, N}, f32);
            TORCH_CHECK(iou.is_cuda() && iou.scalar_type() == at::kFloat && iou.is_contiguous() && iou.numel() == B * N * N, "GNMS: iou_out must be a contiguous CUDA float [B, N, N] tensor");
            auto entry = three_d ? gnms_forward_with_iou3d : gnms_forward_with_iou2d;
            check(entry((const float*)cptr(boxes), (const float*)cptr(s), (int)B, (int)N, ld, (const int32_t*)cptr(counts), &P, (float*)mptr(iou),
                        (float*)mptr(prob), (int64_t*)mptr(order), (int64_t*)mptr(valid), (int64_t*)mptr(invalid), nv_out,
                        ni_out, mptr(ws), (size_t)ws.numel(), st), three_d ? "gnms_forward_with_iou3d" : "gnms_forward_with_iou2d");
            // The masked-group backward (the default) never reads the overlaps, so the matrix is not kept at all.  The unmasked / ungrouped
            // backward does: there it is saved through autograd, whose version counter then catches a caller that overwrites the
            // (possibly caller-provided) buffer between forward and backward instead of silently producing wrong gradients.
            // Grouped unmasked 2D: the backward solves its groups from the BOXES (gnms_backward_from_boxes: bit-identical overlaps, no matrix
            // reads, a caller may reuse iou_out at once); only the ungrouped mode and the 3D entry still read the matrix back.
            // (the same predicate as gnms_forward_with_iou2d's from-boxes gate, alignment included: for an unaligned view of the boxes the forward took
            // the matrix path, and the backward must not run float4 loads on that pointer -- ADVICE r4)
            if (!three_d && group && !mask && !presorted && (reinterpret_cast<uintptr_t>(boxes.data_ptr()) % 16 == 0)) { kept = boxes; bwd_boxes = true; }
            else if (!(group && mask)) kept = iou;
        }
        ctx->set_materialize_grads(false);                               // no zero-filled "gradients" for the index outputs
        ctx->saved_data["mode"] = mode;
        ctx->saved_data["ld"] = ld;
        ctx->saved_data["params"] = std::vector<double>{thr, temp, vthr, (double)prune, (double)sorted_prob, (double)group, (double)mask, (double)gsize,
                                                        (double)presorted};
        ctx->saved_data["has_counts"] = counts.defined();
        ctx->saved_data["has_kept"] = kept.defined();
        ctx->saved_data["bwd_boxes"] = bwd_boxes;
        variable_list saved{s, ws};
        if (counts.defined()) saved.push_back(counts);
        if (kept.defined()) saved.push_back(kept);
        ctx->save_for_backward(saved);
        // outputs: prob, order, nvalid, ninvalid [, valid, invalid] [, iou] -- defined tensors only (an autograd node cannot return an
        // undefined one); layer() below puts them into the seven slots of the Python-side convention
        variable_list out{prob, order, nvalid, ninvalid};
        if (valid.defined()) { out.push_back(valid); out.push_back(invalid); }
        if (iou.defined()) out.push_back(iou);
        ctx->mark_non_differentiable(variable_list(out.begin() + 1, out.end()));
        return out;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        variable_list none(15);
        if (!grads[0].defined()) return none;
        const auto saved = ctx->get_saved_variables();
        const int64_t mode = ctx->saved_data["mode"].toInt(), ld = ctx->saved_data["ld"].toInt();
        const auto pv = ctx->saved_data["params"].toDoubleVector();
        gnms_params P;
        P.nms_threshold = (float)pv[0]; P.temperature = (float)pv[1]; P.valid_box_prob_threshold = (float)pv[2]; P.pruning_method = (int32_t)pv[3];
        P.return_sorted_prob = (int32_t)pv[4]; P.group_boxes = (int32_t)pv[5]; P.mask_group_boxes = (int32_t)pv[6]; P.group_size = (int32_t)pv[7];
        P.presorted = (int32_t)pv[8];
        size_t k = 0;
        Tensor s = saved[k++], ws = saved[k++];
        Tensor counts = ctx->saved_data["has_counts"].toBool() ? saved[k++] : Tensor();
        Tensor kept = ctx->saved_data["has_kept"].toBool() ? saved[k++] : Tensor();
        const int64_t B = s.size(0), N = s.size(1);
        DeviceGuard guard(s.device());
        hipStream_t st = current_stream(s);
        Tensor g = grads[0].contiguous().to(at::kFloat);
        Tensor gs = at::empty_like(s);
        Tensor gi;
        if (ctx->saved_data["bwd_boxes"].toBool()) {
            check(gnms_backward_from_boxes((const float*)cptr(g), (const float*)cptr(kept), (const float*)cptr(s), (int)B, (int)N, (const int32_t*)cptr(counts), &P,
                                           (float*)mptr(gs), mptr(ws), (size_t)ws.numel(), st), "gnms_backward_from_boxes");
        } else {
            if (mode == kMatrixIn && ctx->needs_input_grad(1)) gi = at::empty({B, N, ld}, s.options());
            // (masked groups never dereference the matrix pointer: any valid device pointer will do when the matrix was not kept)
            const float* m = kept.defined() ? (const float*)cptr(kept) : (const float*)cptr(s);
            check(gnms_backward((const float*)cptr(g), (const float*)cptr(s), m, (int)B, (int)N, ld, (const int32_t*)cptr(counts), &P, (float*)mptr(gs),
                                (float*)mptr(gi), mptr(ws), (size_t)ws.numel(), st), "gnms_backward");
            if (gi.defined() && ld != N) gi = gi.slice(2, 0, N);
        }
        none[0] = gs;
        none[1] = gi;
        return none;
    }
};

// -> (prob, order, valid | None, invalid | None, nvalid, ninvalid, iou | None)
std::vector<c10::optional<Tensor>> layer(const Tensor& scores, const Tensor& src, const c10::optional<Tensor>& counts, const c10::optional<Tensor>& iou_out,
                                         int64_t mode, double thr, double temp, double vthr, int64_t prune, bool sorted_prob, bool group, bool mask,
                                         int64_t gsize, bool presorted, bool index_lists) {
    const variable_list o = Layer::apply(scores, src, counts, iou_out, mode, thr, temp, vthr, prune, sorted_prob, group, mask, gsize, presorted, index_lists);
    std::vector<c10::optional<Tensor>> r(7);
    r[0] = o[0]; r[1] = o[1]; r[4] = o[2]; r[5] = o[3];
    size_t k = 4;
    if (index_lists) { r[2] = o[k++]; r[3] = o[k++]; }
    if (k < o.size()) r[6] = o[k];
    return r;
}

// the layer's two count vectors on the host (gnms_counts_to_host: tag-polled pinned mailbox, no device-to-host copy) -> nvalid[0..B) + ninvalid[0..B)
std::vector<int64_t> counts_to_host(const Tensor& nvalid, const Tensor& ninvalid) {
    TORCH_CHECK(nvalid.is_cuda() && ninvalid.is_cuda() && nvalid.scalar_type() == at::kInt && ninvalid.scalar_type() == at::kInt && nvalid.dim() == 1 &&
                ninvalid.sizes() == nvalid.sizes() && nvalid.is_contiguous() && ninvalid.is_contiguous(), "GNMS: counts_to_host takes two contiguous CUDA int32 [B] tensors");
    DeviceGuard guard(nvalid.device());
    const int64_t B = nvalid.size(0);
    std::vector<int32_t> h((size_t)(2 * B));
    hipStream_t st = current_stream(nvalid);
    int rc;
    {
        pybind11::gil_scoped_release nogil;
        rc = gnms_counts_to_host((const int32_t*)cptr(nvalid), (const int32_t*)cptr(ninvalid), (int)B, h.data(), st);
    }
    check(rc, "gnms_counts_to_host");
    return std::vector<int64_t>(h.begin(), h.end());
}

// lib/groomed_nms.py:10-129 for ONE image of GPU tensors in one host call: scores [N], iou [N, N] -> (valid, invalid, prob [N], nvalid [1], ninvalid [1]).
// lists_now: the counts come to the host (gnms_counts_to_host) and valid / invalid are the reference's index tensors of K and N - K entries;
// otherwise they are the padded [N] lists and the two counts stay on the device (LazyIndexList).  The unsqueeze / select / narrow views that
// groomed_nms.py::differentiable_nms makes one torch call at a time from Python (~2 us a piece at N = 500, where the whole call is ~50 us).
std::vector<Tensor> single(const Tensor& scores, const Tensor& iou, double thr, double temp, double vthr, int64_t prune, bool sorted_prob, bool group, bool mask,
                           int64_t gsize, bool presorted, bool lists_now) {
    TORCH_CHECK(scores.dim() == 1 && iou.dim() == 2 && iou.size(0) == scores.size(0) && iou.size(1) == scores.size(0),
                "iou_unsorted must be (N, N) with N = len(scores_unsorted)");
    TORCH_CHECK(scores.is_cuda() && scores.size(0) > 0, "GNMS: single takes a non-empty CUDA tensor");
    DeviceGuard guard(scores.device());
    int32_t* slot_dev = nullptr;
    const int32_t* slot_host = nullptr;
    if (lists_now) {   // (a capture would bake this call's slot into the graph and then fail in the wait below, with launches already recorded)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        TORCH_CHECK(hipStreamIsCapturing(current_stream(scores), &cap) == hipSuccess && cap == hipStreamCaptureStatusNone,
                    "GNMS: differentiable_nms returns index tensors of a data-dependent length -- a host round trip that cannot be captured into a graph; "
                    "capture differentiable_nms_batched, or set groomed_nms.LAZY_INDEX_LISTS = True");
    }
    // lists_now: the forward call writes its counts into a slot of pinned memory the host polls -- no kernel, no copy, no stream synchronisation
    // behind the layer (gnms_host_counts_slot / _wait); no slot (no fine-grained memory): the counts tensor + gnms_counts_to_host
    // (every slot owned -- more than 64 calls in flight on the device -- is GNMS_ERR_UNSUPPORTED: the plain path below)
    if (lists_now && gnms_host_counts_slot(1, &slot_dev, &slot_host) != GNMS_OK) slot_dev = nullptr;
    // the slot is this call's until the wait below has returned; a forward call that throws gives it back (behind the stream: the launches
    // that did get enqueued may still store their counts)
    struct Reset {
        const int32_t* view;
        hipStream_t st;
        ~Reset() { tl_counts_slot = nullptr; if (view) (void)gnms_host_counts_release(view, st); }
    } reset{slot_dev ? slot_host : nullptr, current_stream(scores)};
    tl_counts_slot = slot_dev;
    const variable_list o = Layer::apply(scores.unsqueeze(0), iou.unsqueeze(0), c10::nullopt, c10::nullopt, (int64_t)kMatrixIn, thr, temp, vthr, prune, sorted_prob,
                                         group, mask, gsize, presorted, true);
    tl_counts_slot = nullptr;
    Tensor valid = o[4].select(0, 0), invalid = o[5].select(0, 0);
    if (lists_now) {
        int64_t k, m;
        if (slot_dev) {
            int32_t c[2];
            int rc;
            hipStream_t st = current_stream(scores);
            {
                pybind11::gil_scoped_release nogil;
                reset.view = nullptr;                                    // the wait gives the slot back on every path
                rc = gnms_host_counts_wait(slot_host, 1, c, st);
            }
            check(rc, "gnms_host_counts_wait");
            k = c[0]; m = c[1];
        } else {
            const std::vector<int64_t> c = counts_to_host(o[2], o[3]);
            k = c[0]; m = c[1];
        }
        return {valid.narrow(0, 0, k), invalid.narrow(0, 0, m), o[0].select(0, 0)};
    }
    return {valid, invalid, o[0].select(0, 0), o[2], o[3]};
}

// lib/core.py:480-508 iou(mode='combinations') for batches: boxes_a [B,M,4], boxes_b [B,N,4] -> [B,M,N]
Tensor iou2d(const Tensor& a, const Tensor& b, const c10::optional<Tensor>& out_) {
    TORCH_CHECK(a.is_cuda() && b.is_cuda() && a.dim() == 3 && b.dim() == 3 && a.size(2) == 4 && b.size(2) == 4 && a.size(0) == b.size(0) &&
                a.scalar_type() == at::kFloat && b.scalar_type() == at::kFloat, "GNMS: iou2d takes CUDA float [B, M, 4] and [B, N, 4]");
    DeviceGuard guard(a.device());
    Tensor ac = a.contiguous(), bc = b.contiguous();
    const int64_t B = ac.size(0), M = ac.size(1), N = bc.size(1);
    if (out_.has_value())
        TORCH_CHECK(out_->is_cuda() && out_->device() == ac.device() && out_->scalar_type() == at::kFloat && out_->dim() == 3 && out_->size(0) == B &&
                    out_->size(1) == M && out_->size(2) == N && out_->is_contiguous(), "GNMS: iou2d `out` must be a contiguous CUDA float [B, M, N] tensor on the boxes' device");
    Tensor out = out_.has_value() ? *out_ : at::empty({B, M, N}, ac.options());
    check(gnms_iou2d((const float*)cptr(ac), (const float*)cptr(bc), (int)B, (int)M, (int)N, (float*)mptr(out), std::max<int64_t>(N, 1),
                     current_stream(ac)), "gnms_iou2d");
    return out;
}

// ------------------------------------------------------------------------------------------------
// The layer's neighbours on the same host path as the layer itself (round 5; they were Python autograd.Functions over ctypes:
// 0.12-0.13 ms per AP-loss step at every size against 17-57 us of kernel time).
// ------------------------------------------------------------------------------------------------
// lib/loss/aploss.py:14-87 for a batch: logits / targets [B, N] -> loss [B]; d loss / d logits comes out of the forward pass (as in the
// reference, :69-78) and the backward node only scales it (:80-85)
struct APLossNode : public torch::autograd::Function<APLossNode> {
    static Tensor forward(AutogradContext* ctx, const Tensor& logits, const Tensor& targets, const c10::optional<Tensor>& counts_, double pos, double neg) {
        TORCH_CHECK(logits.is_cuda() && logits.dim() == 2 && targets.is_cuda() && targets.sizes() == logits.sizes(), "GNMS: aploss takes CUDA [B, N] logits and targets");
        DeviceGuard guard(logits.device());
        Tensor lg = logits.detach().to(at::kFloat).contiguous(), tg = targets.detach().to(at::kFloat).contiguous();
        Tensor counts = counts_.has_value() ? *counts_ : Tensor();
        const int64_t B = lg.size(0), N = lg.size(1);
        Tensor loss = at::empty({B}, lg.options()), grad = at::empty({B, N}, lg.options());
        check(gnms_aploss((const float*)cptr(lg), (const float*)cptr(tg), (int)B, (int)N, (const int32_t*)cptr(counts), (float)pos, (float)neg,
                          (float*)mptr(loss), (float*)mptr(grad), current_stream(lg)), "gnms_aploss");
        ctx->save_for_backward({grad});
        ctx->saved_data["dtype"] = (int64_t)logits.scalar_type();
        return loss;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        variable_list out(5);
        if (!grads[0].defined()) return out;
        const Tensor grad = ctx->get_saved_variables()[0];
        out[0] = (grad * grads[0].reshape({-1, 1})).to((at::ScalarType)ctx->saved_data["dtype"].toInt());
        return out;
    }
};
Tensor aploss(const Tensor& logits, const Tensor& targets, const c10::optional<Tensor>& counts, double pos, double neg) {
    return APLossNode::apply(logits, targets, counts, pos, neg);
}

// lib/loss/rpn_3d.py:801-825 -> (targets [B, N] fp32, best_index [B, M] int64, best_score [B, M])
std::vector<Tensor> best_targets(const Tensor& pred_params, const Tensor& pred_boxes, const Tensor& gt_params, const Tensor& gt_boxes, double beta,
                                 const c10::optional<Tensor>& pred_counts, const c10::optional<Tensor>& gt_counts) {
    TORCH_CHECK(pred_params.is_cuda() && pred_params.dim() == 3 && pred_params.size(2) == 7 && gt_params.dim() == 3 && gt_params.size(2) == 7,
                "GNMS: best_targets takes CUDA [B, N, 7] / [B, M, 7] cuboid parameters");
    DeviceGuard guard(pred_params.device());
    Tensor pp = pred_params.detach().to(at::kFloat).contiguous(), pb = pred_boxes.detach().to(at::kFloat).slice(-1, 0, 4).contiguous();
    Tensor gp = gt_params.detach().to(at::kFloat).contiguous(), gb = gt_boxes.detach().to(at::kFloat).slice(-1, 0, 4).contiguous();
    const int64_t B = pp.size(0), N = pp.size(1), M = gp.size(1);
    Tensor pc = pred_counts.has_value() ? *pred_counts : Tensor(), gc = gt_counts.has_value() ? *gt_counts : Tensor();
    Tensor targets = at::empty({B, N}, pp.options()), idx = at::empty({B, M}, pp.options().dtype(at::kLong)), score = at::empty({B, M}, pp.options());
    check(gnms_best_targets((const float*)cptr(pp), (const float*)cptr(pb), (const float*)cptr(gp), (const float*)cptr(gb), (int)B, (int)N, (int)M,
                            (const int32_t*)cptr(pc), (const int32_t*)cptr(gc), (float)beta, (int64_t*)mptr(idx), (float*)mptr(score), (float*)mptr(targets),
                            current_stream(pp)), "gnms_best_targets");
    return {targets, idx, score};
}

// lib/loss/rpn_3d.py:731-737 / lib/rpn_util.py:1258-1266 -> (index [B, K] int64, count [B] int32, scores [B, K], boxes [B, K, 4] | undefined)
std::vector<c10::optional<Tensor>> select_topk(const Tensor& scores, int64_t K, const c10::optional<Tensor>& candidates, const c10::optional<Tensor>& candidate_counts,
                                               const c10::optional<Tensor>& boxes) {
    TORCH_CHECK(scores.is_cuda() && scores.dim() == 2, "GNMS: select_topk takes CUDA [B, A] scores");
    DeviceGuard guard(scores.device());
    Tensor s = scores.detach().to(at::kFloat).contiguous();
    const int64_t B = s.size(0), A = s.size(1);
    Tensor cand = candidates.has_value() ? candidates->to(at::kInt).contiguous() : Tensor();
    Tensor cnt = (candidates.has_value() && candidate_counts.has_value()) ? candidate_counts->to(at::kInt).contiguous() : Tensor();
    const int64_t F = cand.defined() ? cand.size(1) : A;
    Tensor bx = boxes.has_value() ? boxes->detach().to(at::kFloat).slice(-1, 0, 4).contiguous() : Tensor();
    Tensor idx = at::empty({B, K}, s.options().dtype(at::kLong)), num = at::empty({B}, s.options().dtype(at::kInt)), ssel = at::empty({B, K}, s.options());
    Tensor bsel = bx.defined() ? at::empty({B, K, 4}, s.options()) : Tensor();
    check(gnms_select_topk((const float*)cptr(s), (int)B, (int)A, (const int32_t*)cptr(cand), (int)F, (const int32_t*)cptr(cnt), (int)K, (const float*)cptr(bx),
                           (int64_t*)mptr(idx), (int32_t*)mptr(num), (float*)mptr(ssel), (float*)mptr(bsel), current_stream(s)), "gnms_select_topk");
    std::vector<c10::optional<Tensor>> r(4);
    r[0] = idx; r[1] = num; r[2] = ssel;
    if (bsel.defined()) r[3] = bsel;
    return r;
}

// lib/rpn_util.py:872-934: anchors [A, 4], deltas [B, A, 4] -> [B, A, 4]; means / stds: 4 host floats each (or empty)
Tensor bbox_transform_inv(const Tensor& anchors, const Tensor& deltas, const std::vector<double>& means, const std::vector<double>& stds) {
    TORCH_CHECK(deltas.is_cuda() && deltas.dim() == 3 && deltas.size(2) == 4 && anchors.dim() == 2 && anchors.size(1) == 4 && anchors.size(0) == deltas.size(1),
                "GNMS: bbox_transform_inv takes anchors [A, 4] and CUDA deltas [B, A, 4]");
    DeviceGuard guard(deltas.device());
    Tensor d = deltas.detach().to(at::kFloat).contiguous(), a = anchors.detach().to(deltas.device(), at::kFloat).contiguous();
    float m[4], sd[4];
    for (int i = 0; i < 4; ++i) { m[i] = means.size() == 4 ? (float)means[i] : 0.0f; sd[i] = stds.size() == 4 ? (float)stds[i] : 1.0f; }
    Tensor out = at::empty_like(d);
    check(gnms_bbox_transform_inv((const float*)cptr(a), (const float*)cptr(d), (int)d.size(0), (int)d.size(1), means.size() == 4 ? m : nullptr,
                                  stds.size() == 4 ? sd : nullptr, (float*)mptr(out), current_stream(d)), "gnms_bbox_transform_inv");
    return out;
}

// The training tail of lib/loss/rpn_3d.py:772-825 + :1117-1131 as ONE host call, nothing but stream-ordered launches (HIP-graph capturable):
// scores [B, N] already in descending order per image (as the loss sorts them, :731-737), boxes2d [B, N, 4], params3d [B, N, 7], ground truth
// [B, M, 7] / [B, M, 4]:   GrooMeD-NMS on the 2D overlaps (:772-793) -> best box per ground truth (:801-825) -> after-NMS AP loss on the
// rescored probabilities against those targets (:1123-1128).  -> (loss [B], prob [B, N], targets [B, N])
std::vector<Tensor> training_tail(const Tensor& scores, const Tensor& boxes2d, const Tensor& params3d, const Tensor& gt_params, const Tensor& gt_boxes, double beta,
                                  const c10::optional<Tensor>& counts, const c10::optional<Tensor>& gt_counts, double thr, double temp, double vthr,
                                  int64_t prune, int64_t gsize) {
    const variable_list o = Layer::apply(scores, boxes2d, counts, c10::nullopt, (int64_t)kWithIou2d, thr, temp, vthr, prune, false, true, true, gsize, false, false);
    const std::vector<Tensor> bt = best_targets(params3d, boxes2d, gt_params, gt_boxes, beta, counts, gt_counts);
    // (the scores came in sorted, so rank order == input order and the targets line up with prob as they are)
    Tensor loss = APLossNode::apply(o[0], bt[0], counts, 1.0, 0.0);
    return {loss, o[0], bt[0]};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "GrooMeD-NMS layer: C++ autograd binding of libgroomed_nms_hip.so";
    m.def("layer", &layer, "forward entry + autograd node (mode 0 matrix in, 1 boxes -> matrix + layer, 2 cuboids -> matrix + layer, 3 from boxes)");
    m.def("iou2d", &iou2d, "pairwise 2D IoU matrices");
    m.def("aploss", &aploss, "after-NMS AP loss of a batch, autograd node (lib/loss/aploss.py)");
    m.def("best_targets", &best_targets, "best box per ground truth after the NMS (lib/loss/rpn_3d.py:801-825)");
    m.def("select_topk", &select_topk, "per image the K best-scoring candidates (lib/loss/rpn_3d.py:731-737)");
    m.def("bbox_transform_inv", &bbox_transform_inv, "lib/rpn_util.py:872-934");
    m.def("training_tail", &training_tail, "layer -> best targets -> AP loss in one host call");
    m.def("counts_to_host", &counts_to_host, "nvalid / ninvalid of a forward call on the host through the pinned mailbox (one blocking round trip, no copy)");
    m.def("single", &single, "differentiable_nms of one image of GPU tensors: layer + host round trip + the reference's index tensors in one host call");
    m.def("abi_version", [] { return gnms_abi_version(); });
}
