/*
 * groomed_nms_hip.h -- C ABI of libgroomed_nms_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the GrooMeD-NMS hot path of abhi1kumar/groomed_nms.  Citations are
 * file:line in the reference checkout.  The reference's boundary for this path is a Python call
 * (lib/groomed_nms.py:10 differentiable_nms, imported at lib/loss/rpn_3d.py:14 and
 * lib/rpn_util.py:18) plus ONE true C symbol, `_nms` (lib/nms/gpu_nms.hpp:1-2, bound by Cython at
 * lib/nms/gpu_nms.pyx:13-14).  `_nms` is exported here with the identical signature; the gnms_*
 * functions are what a ctypes/Cython/cffi binding of lib/groomed_nms.py and of the overlap helpers
 * in lib/core.py binds (see INTEGRATION.md for the stubs).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in _host;
 *   - calls are asynchronous and ordered on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream); no hidden synchronisation and no hidden allocation, except `_nms`, which keeps
 *     the reference's blocking host-pointer contract, and the 3D entries (gnms_iou3d_*, gnms_forward_with_iou3d),
 *     which take a stream-ordered temporary (hipMallocAsync/hipFreeAsync) of 48-64 bytes per box for the cuboid records;
 *     and gnms_iou2d of a box set with itself (boxes_a == boxes_b, N <= 4096), whose FIRST call on a device allocates a 1-MiB ring
 *     of claim counters that lives as long as the process (hipMalloc: make that first call outside stream capture; later calls
 *     allocate nothing and are capturable);
 *   - gnms_forward_with_iou2d / _iou3d on images of more than 4096 boxes issue the matrix write on a library-owned stream
 *     (one per device, lowest priority) that forks from `stream` and joins it again before the call returns control of the
 *     ordering to the caller: to the caller everything is still ordered on `stream` (earlier work happens before, later work
 *     after), also under stream capture (the fork/join is captured as a branch of the graph);
 *   - a batch is B images of up to N boxes; image b uses the first counts[b] boxes (counts may be
 *     NULL: every image has N).  Scores are [B][N]; overlap matrices are [B][N][ld] row-major with
 *     row stride ld >= N elements (image stride N*ld);
 *   - return value: 0 on success, negative gnms_status on failure; gnms_last_error() gives the
 *     message of the calling thread's last failure (the reference prints CUDA errors and carries
 *     on, lib/nms/nms_kernel.cu:12-19; this library never does that).
 *   - fp32 arithmetic without FMA contraction, IEEE division: results are bit-identical to the
 *     CPU restatement in oracle/ wherever that one is fp32-sequential (overlaps, default rescoring).
 */
#ifndef GROOMED_NMS_HIP_H
#define GROOMED_NMS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNMS_ABI_VERSION 1
#define GNMS_MAX_BOXES 16384 /* per image; the in-LDS sorts and merges hold 16384 64-bit keys in 128 KiB of the CU's 160 KiB */

typedef enum gnms_status {
    GNMS_OK = 0,
    GNMS_ERR_INVALID_ARGUMENT = -1,
    GNMS_ERR_UNSUPPORTED = -2, /* e.g. N > GNMS_MAX_BOXES, unknown pruning method (reference: NotImplementedError) */
    GNMS_ERR_HIP = -3,
    GNMS_ERR_WORKSPACE = -4
} gnms_status;

/* lib/groomed_nms.py:167-189 pruning_function */
typedef enum gnms_pruning { GNMS_PRUNE_LINEAR = 0, GNMS_PRUNE_SIGMOIDAL = 1, GNMS_PRUNE_SOFT_NMS = 2 } gnms_pruning;

/* keyword arguments of differentiable_nms (lib/groomed_nms.py:10), same names, same defaults */
typedef struct gnms_params {
    float nms_threshold;            /* 0.4  */
    float temperature;              /* 0.01 (unused by "linear") */
    float valid_box_prob_threshold; /* 0.3  */
    int32_t pruning_method;         /* gnms_pruning */
    int32_t return_sorted_prob;     /* 0 */
    int32_t group_boxes;            /* 1 */
    int32_t mask_group_boxes;       /* 1 */
    int32_t group_size;             /* 100 */
    int32_t presorted;              /* 0.  1 = the soft-sort hand-off (lib/groomed_nms.py:42-45): scores/iou are
                                       consumed in INPUT order (prune matrix, rescoring, returned probabilities), only
                                       the grouping re-sorts by score as get_groups does (:213-214). */
} gnms_params;

void gnms_default_params(gnms_params* p);
int gnms_abi_version(void);
const char* gnms_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Pairwise overlaps (lib/core.py)
 * ------------------------------------------------------------------------------------------- */

/* iou(box_a, box_b, mode='combinations')  lib/core.py:480-508 (+ intersect :178-218).
 * boxes_a [B][M][4], boxes_b [B][N][4] as (x1,y1,x2,y2); out[b][i][j] = IoU(a_i, b_j), row stride ld >= N.
 * No +1 pixel; a zero-area pair yields 0/0 = NaN exactly like the reference. */
int gnms_iou2d(const float* boxes_a, const float* boxes_b, int B, int M, int N, float* out, int64_t ld, void* stream);

/* get_corners_of_cuboid  lib/math_3d.py:364-435 (torch branch, iou_3d_convention=True).
 * params [count][7] = (x3d, y3d, z3d, w3d, h3d, l3d, ry3d) -> corners [count][3][8]. */
int gnms_corners_of_cuboid(const float* params, int64_t count, float* corners, void* stream);

/* The float64 NumPy branches of the same two helpers -- what the reference's inference call site runs (lib/rpn_util.py:1292-1320:
 * `aboxes` is float64 after the hstack at :1258): lib/core.py:205-207, 512-513 (iou in float64, rounded to fp32 once by
 * lib/groomed_nms.py:36) and lib/math_3d.py:438-490 (corners in float64 before `.float()`).  Same layouts as above with double
 * elements; the IoU matrix is bit-identical to NumPy's IEEE double arithmetic.  trig_f32 != 0: cos / sin of the yaw in fp32, widened
 * (np.cos on a float32 `ry3d`, the dtype that call site passes). */
int gnms_iou2d_f64(const double* boxes_a, const double* boxes_b, int B, int M, int N, double* out, int64_t ld, void* stream);
int gnms_corners_of_cuboid_f64(const double* params, int64_t count, int trig_f32, double* corners, void* stream);

/* iou3d_approximate(corners_b1, corners_b2, mode="combinations", method=...)  lib/core.py:305-421.
 * corners_a [B][M][3][8], corners_b [B][N][3][8].  Inputs are const (the reference overwrites them, :379-380).
 *   method 0 "normal", 1 "generalized", 2 = 0.5*(1+generalized): what both callers feed the NMS
 *   (lib/loss/rpn_3d.py:781, lib/rpn_util.py:1312).
 * iou_bev may be NULL.  Outputs [B][M][ld]. */
int gnms_iou3d_approximate(const float* corners_a, const float* corners_b, int B, int M, int N, int method,
                           float* iou_bev, float* iou_3d, int64_t ld, void* stream);

/* same, with get_corners_of_cuboid fused as the prologue: params_a [B][M][7], params_b [B][N][7] */
int gnms_iou3d_from_params(const float* params_a, const float* params_b, int B, int M, int N, int method,
                           float* iou_bev, float* iou_3d, int64_t ld, void* stream);

/* The matrix both reference callers hand to the NMS in 3D mode: 0.5 * (1 + GIoU3D) of the cuboids' corner AABBs
 * (lib/loss/rpn_3d.py:778-781, lib/rpn_util.py:1309-1312), params3d [B][N][7] -> out [B][N][ld], when the caller knows the
 * threshold the layer will apply.  Same values as gnms_iou3d_from_params(method 2) to within 2e-6, but HBM-write bound instead of
 * division bound: one reciprocal per pair (0.5 * (i3 vh + u3^2) / (u3 vh)), and every entry within 8e-6 of `nms_threshold` is
 * replaced by the reference's exact operation order, so that `entry > nms_threshold` takes the reference's decision for every
 * pair.  gnms_forward_with_iou3d writes its matrix with the same kernel. */
int gnms_nms_overlap3d_from_params(const float* params3d, int B, int N, float nms_threshold, float* out, int64_t ld, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GrooMeD-NMS layer (lib/groomed_nms.py:10-129), hard sort.  Soft sort = gnms_soft_sort + presorted=1.
 * ------------------------------------------------------------------------------------------- */

/* bytes of the workspace gnms_forward needs for (B, N) with these params (NULL: the defaults).  The same buffer carries the
 * state that gnms_backward reads, so keep it alive and untouched between the two calls.  Grouped modes: ~30 N-sized arrays +
 * the N^2/8-byte bit matrix per image; ungrouped mode (group_boxes = 0) adds a 4 N^2-byte scratch matrix per image (the
 * sorted strictly-lower-triangular copy the reference also makes, lib/groomed_nms.py:48). */
size_t gnms_workspace_bytes(int B, int N, const gnms_params* params);

/* forward.  scores [B][N], iou [B][N][ld].
 *   prob    [B][N]  third return value (:124-129): rescored probabilities in descending-input-score
 *                   order (grouped: un-thresholded clone; ungrouped: thresholded; return_sorted_prob:
 *                   thresholded and sorted).  presorted=1: input order.
 *   order   [B][N]  rank -> input index (the `indices` of :41), int64
 *   valid   [B][N]  first return value, first nvalid[b] entries (input indices, by descending rescored prob)
 *   invalid [B][N]  second return value, first ninvalid[b] entries
 *   nvalid, ninvalid [B] int32 (NaN probabilities are in neither list, :118-123)
 * order/valid/invalid/nvalid/ninvalid may each be NULL if not wanted. */
int gnms_forward(const float* scores, const float* iou, int B, int N, int64_t ld, const int32_t* counts,
                 const gnms_params* params, float* prob, int64_t* order, int64_t* valid, int64_t* invalid,
                 int32_t* nvalid, int32_t* ninvalid, void* workspace, size_t workspace_bytes, void* stream);

/* gnms_iou2d + gnms_forward in one call, as lib/loss/rpn_3d.py:772-791 runs them back to back: boxes [B][N][4] ->
 * iou_out [B][N][ld] (kept for the caller) -> the outputs of gnms_forward.  With the boxes at hand the grouped hard-sort
 * modes take their threshold bits and group overlaps from the boxes (the from-boxes kernels below, bit-identical) instead
 * of reading back the matrix they just wrote: nothing in the layer waits for the matrix, and up to N = 4096 its write
 * shares ONE launch with the per-image chain (masked groups: K3..K6; unmasked: up to the group structure, the per-group
 * solves and K6 follow).  N > 4096 (masked groups): the write runs beside the layer on the library's side stream (see
 * Conventions).  The ungrouped mode builds its pruned triangular matrix from the boxes as well; only pre-sorted scores
 * read the matrix back.  gnms_backward pairs with it unchanged (grouped unmasked: gnms_backward_from_boxes does too, and
 * never reads the matrix). */
int gnms_forward_with_iou2d(const float* boxes, const float* scores, int B, int N, int64_t ld, const int32_t* counts,
                            const gnms_params* params, float* iou_out, float* prob, int64_t* order, int64_t* valid,
                            int64_t* invalid, int32_t* nvalid, int32_t* ninvalid, void* workspace, size_t workspace_bytes,
                            void* stream);

/* The same for the 3D overlap of lib/loss/rpn_3d.py:778-784 (overlap_in_nms == "3d"): params3d [B][N][7] = x y z w h l ry ->
 * iou_out [B][N][ld] = 0.5 * (1 + GIoU3D) (as gnms_nms_overlap3d_from_params with params->nms_threshold) -> the outputs of gnms_forward.  Grouped + masked
 * hard-sort modes take the threshold bits and the single group overlaps from the cuboid records with the arithmetic that wrote
 * the matrix (no read-back of the 4 N^2 bytes; N > 4096: matrix write on the side stream); the other modes read the matrix.
 * gnms_backward pairs with it unchanged. */
int gnms_forward_with_iou3d(const float* params3d, const float* scores, int B, int N, int64_t ld, const int32_t* counts,
                            const gnms_params* params, float* iou_out, float* prob, int64_t* order, int64_t* valid,
                            int64_t* invalid, int32_t* nvalid, int32_t* ninvalid, void* workspace, size_t workspace_bytes,
                            void* stream);

/* The two per-image counts of a forward call on the HOST: what the reference's return convention needs before it can shape
 * `valid_boxes_index` / `invalid_boxes_index` (lib/groomed_nms.py:120-127; their length K is data dependent).  Blocking; ordered behind
 * everything enqueued on `stream` before it.  nvalid / ninvalid: the DEVICE arrays [B] a gnms_forward* call wrote; host_out: HOST array
 * [2 B] = nvalid[0..B) then ninvalid[0..B).  Not a device-to-host copy: a one-wave kernel stores the counts and a call tag into a slot of
 * fine-grained pinned memory the library keeps per device (64 KiB, allocated by the first call) and the host polls the tag (one PCIe
 * write instead of a copy submission + a stream synchronisation: ~20 us less per call at N = 500).  B > 127, or no tag within two
 * seconds: a plain asynchronous copy + stream synchronisation.  Cannot be captured into a graph (GNMS_ERR_INVALID_ARGUMENT). */
int gnms_counts_to_host(const int32_t* nvalid, const int32_t* ninvalid, int B, int32_t* host_out, void* stream);
/* The same trip without any launch behind the layer: gnms_host_counts_slot hands out 2 B words of that pinned memory, preset to -1, as the
 * device (`device_view`) and the host (`host_view`) address them; pass device_view as `nvalid` and device_view + B as `ninvalid` to ONE
 * gnms_forward* call (the kernels store each count exactly once, at the end of the image's chain), then gnms_host_counts_wait polls
 * host_view until all 2 B words are counts (>= 0) and copies them to host_out [2 B]; it falls back to a stream synchronisation when the
 * stream runs empty first or after two seconds.  B <= 127 (GNMS_ERR_UNSUPPORTED above).
 * OWNERSHIP: the slot belongs to the caller from gnms_host_counts_slot until gnms_host_counts_wait returns (whatever it returns) or
 * gnms_host_counts_release is called -- exactly one of the two per slot; nobody else is handed the slot in between, however many calls other
 * threads make on the device.  When all 64 slots of the device are owned, gnms_host_counts_slot returns GNMS_ERR_UNSUPPORTED and the caller
 * takes the plain path (device counts + gnms_counts_to_host).  gnms_host_counts_release is for a caller that took a slot and will not wait
 * (its forward call failed, or was never made): it synchronises `stream` first, so a forward call that was enqueued can no longer store
 * into a slot somebody else owns by then.  A wait / release on a view that is not owned is GNMS_ERR_INVALID_ARGUMENT.
 * differentiable_nms at N = 500, index tensors included: 64 -> 46 us per call. */
int gnms_host_counts_slot(int B, int32_t** device_view, const int32_t** host_view);
int gnms_host_counts_wait(const int32_t* host_view, int B, int32_t* host_out, void* stream);
int gnms_host_counts_release(const int32_t* host_view, void* stream);
/* test hook: the number of mailbox slots gnms_host_counts_slot / gnms_counts_to_host may use per device (1 .. 64); returns the previous value */
int gnms_test_mailbox_slots(int n);

/* backward of L through prob.  grad_prob [B][N] = dL/dprob (same order as prob).
 *   grad_scores [B][N] (input order), overwritten.
 *   grad_iou    [B][N][ld] or NULL.  When given it is fully overwritten (zero fill + the sparse
 *               entries); training detaches the matrix (lib/loss/rpn_3d.py:791), so NULL is the fast path.
 * scores/iou/counts/params/workspace must be the ones the forward call saw. */
int gnms_backward(const float* grad_prob, const float* scores, const float* iou, int B, int N, int64_t ld,
                  const int32_t* counts, const gnms_params* params, float* grad_scores, float* grad_iou,
                  void* workspace, size_t workspace_bytes, void* stream);

/* From-boxes path (2D): the same layer computed straight from boxes [B][N][4] = (x1,y1,x2,y2); the N x N overlap
 * matrix is never materialised -- the bit-matrix kernel recomputes every pair's IoU in registers with the arithmetic of
 * gnms_iou2d (lib/core.py:499-508), so all outputs and gradients are bit-identical to gnms_iou2d + gnms_forward /
 * gnms_backward.  Grouped, hard-sorted modes only (masked or unmasked groups); ungrouped / presorted return
 * GNMS_ERR_UNSUPPORTED (they need the matrix).  HBM traffic drops from 8 N^2 to N^2/8 + O(N) bytes per image; the
 * bound becomes fp32 VALU.  Workspace: gnms_workspace_bytes. */
int gnms_forward_from_boxes(const float* boxes, const float* scores, int B, int N, const int32_t* counts,
                            const gnms_params* params, float* prob, int64_t* order, int64_t* valid, int64_t* invalid,
                            int32_t* nvalid, int32_t* ninvalid, void* workspace, size_t workspace_bytes, void* stream);
int gnms_backward_from_boxes(const float* grad_prob, const float* boxes, const float* scores, int B, int N,
                             const int32_t* counts, const gnms_params* params, float* grad_scores, void* workspace,
                             size_t workspace_bytes, void* stream);

/* profiling hook: re-runs only the threshold bit-matrix kernel (the single full read of the overlap matrix, the
 * dominant kernel of gnms_forward) on a workspace a previous gnms_forward call filled. */
int gnms_profile_bitmask(const float* iou, int B, int N, int64_t ld, const int32_t* counts, float nms_threshold,
                         void* workspace, size_t workspace_bytes, void* stream);

/* same hook for the from-boxes bit-matrix kernel (VALU bound) */
int gnms_profile_bitmask_boxes(const float* boxes, int B, int N, const int32_t* counts, float nms_threshold, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Per-launch timing of the two HBM-bound launches inside whatever call sequence the caller runs (bench.py's roofline line).
 * gnms_profile_events(1) arms it: from then on every launch that writes an N x N overlap matrix (slot GNMS_PROF_MATRIX_WRITE:
 * gnms_iou2d, gnms_iou3d_*, gnms_forward_with_iou2d / _iou3d), every launch of the kernel that reads one (slot GNMS_PROF_MATRIX_READ:
 * gnms_forward's bit-matrix kernel) and the plain streams below (slot GNMS_PROF_PLAIN_STREAM) go through hipExtLaunchKernel with a
 * start and a stop event -- the dispatch's own begin / end timestamps, what a kernel trace reports, no marker packet in the stream.
 * Not capturable in a HIP graph; one profiling thread at a time.  gnms_profile_collect waits for the recorded events
 * and returns the SUM of the launch durations (ms) and their number since the last collect.  gnms_profile_events(0) disarms.
 * gnms_profile_fill / gnms_profile_read: a plain non-temporal float4 store / load stream over `count` floats -- the HBM write /
 * read rate a kernel that does nothing else reaches on this device (the ceiling the roofline fraction is quoted beside). */
#define GNMS_PROF_MATRIX_WRITE 0
#define GNMS_PROF_MATRIX_READ 1
#define GNMS_PROF_PLAIN_STREAM 2
int gnms_profile_events(int enable);
int gnms_profile_collect(int slot, double* ms_sum, int* launches);
const char* gnms_profile_write_kernel_name(int dim, int B, int N); /* the launch that writes the matrix in gnms_forward_with_iou2d (dim 2) / _iou3d (dim 3), as a kernel trace names it */
int gnms_profile_fill(float* dst, size_t count, void* stream);
int gnms_profile_fill_tiles(float* dst, int B, int N, int64_t ld, int rows, int nontemporal, void* stream);
/* the store pattern of a symmetric matrix writer, no arithmetic: upper-triangular tile x tile macro tiles (tile = 128 or 256, dividing
 * N), each written where it is and mirrored; cols_per_lane 2 or 4 floats per lane and store instruction; persist: one workgroup per
 * CU (two at tile 128) walks a contiguous range of tiles, else one workgroup per tile.  Timed like gnms_profile_fill. */
int gnms_profile_fill_sym(float* dst, int B, int N, int64_t ld, int tile, int cols_per_lane, int nontemporal, int persist, void* stream); /* the same store stream in the matrix writers' geometry: persistent 16-wave workgroups, `rows` (4, 8, 16, 32 or 64, dividing N) rows x 1 KiB per wave, rows ld floats apart; non-temporal (as the writers store) or ordinary 16-byte stores */
int gnms_profile_read(const float* src, size_t count, float* sink, void* stream);

/* get_groups(iou_unsorted, group_threshold, scores_unsorted, group_size)  lib/groomed_nms.py:208-270 for one
 * image.  group_of[N]: index of the box's group (groups numbered in creation order) or -1 if the box is in
 * no group (beyond the cap, or NaN overlap with its leader); pos_in_group[N]: position inside the group
 * (0 = first member).  Both int32, indexed by INPUT index.  *ngroups_out is a device int32. */
int gnms_get_groups(const float* scores, const float* iou, int N, int64_t ld, float group_threshold, int group_size,
                    int32_t* group_of, int32_t* pos_in_group, int32_t* ngroups_out, void* workspace,
                    size_t workspace_bytes, void* stream);

/* pruning_function(iou, nms_threshold, temperature, pruning_method)  lib/groomed_nms.py:167-189, elementwise */
int gnms_pruning_function(const float* iou, int64_t count, float nms_threshold, float temperature, int pruning_method,
                          float* out, void* stream);
/* its adjoint (the reference differentiates the same expressions with autograd): grad_iou[i] = grad_out[i] * f'(iou[i]) */
int gnms_pruning_function_backward(const float* iou, const float* grad_out, int64_t count, float nms_threshold, float temperature,
                                   int pruning_method, float* grad_iou, void* stream);

/* soft_sort(scores, full_matrix, temperature)  lib/groomed_nms.py:131-165 for one image.
 * C [N][N] (convex_comb_matrix, including the reference's last-axis broadcast of the row sums, :155),
 * soft_scores [N] = C s, soft_matrix [N][N] = C iou (fp32 MFMA GEMM).  iou/soft_matrix may both be NULL. */
int gnms_soft_sort(const float* scores, const float* iou, int N, int64_t ld, float temperature, float* C,
                   float* soft_scores, float* soft_matrix, void* workspace, size_t workspace_bytes, void* stream);

/* Adjoint of gnms_soft_sort (the reference differentiates lib/groomed_nms.py:145-164 with autograd).  scores, temperature, C and
 * `workspace` are what the forward call saw / produced (the workspace still holds the sorted scores, their order and the row sums);
 * matrix [N][ld] with K columns is the matrix that was multiplied (NULL: none).  Upstream gradients, each may be NULL: g_soft [N]
 * (of soft_scores), g_C [N][N] (of C), g_mat [N][K] (of soft_matrix).  Outputs: d_scores [N]; d_matrix [N][ld] or NULL.
 * scratch: gnms_soft_sort_backward_scratch_bytes(N, K) bytes, 16-byte aligned.  The two products (g_mat M^T, C^T g_mat) run on
 * the MFMA GEMM; row and column passes are deterministic (no atomics). */
size_t gnms_soft_sort_backward_scratch_bytes(int N, int K);
int gnms_soft_sort_backward(const float* scores, const float* matrix, int N, int K, int64_t ld, float temperature, const float* C,
                            const float* g_soft, const float* g_C, const float* g_mat, float* d_scores, float* d_matrix,
                            void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, void* stream);

/* D[M x N] = A[M x K] B[K x N], fp32 row-major with leading dimensions lda/ldb/ldd, on the matrix cores
 * (v_mfma_f32_32x32x2_f32: exact fp32, an fmaf chain over k).  The GEMM behind soft_sort's C @ iou (:163).  Small problems (fewer
 * than two 128 x 128 tiles per CU) split K over up to 16 slices: a stream-ordered temporary of slices x M x N floats, the slices
 * summed in order (deterministic; a different association of the same sum). */
int gnms_sgemm(const float* A, const float* B, float* D, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldd,
               void* stream);
/* Round 6: from 512 x 512 x 512 on the PLAIN product runs on rocBLAS, from 2048^3 on on hipBLASLt, when the library can be found (dlopen at
 * first use; a stream that is being captured keeps the kernels of this library), see csrc/soft_sort.hip.  gnms_profile_sgemm picks the path
 * for comparisons: variant 0 = what gnms_sgemm does, 1 = this library's MFMA kernels only, 2 = rocBLAS only, 4 = hipBLASLt only
 * (GNMS_ERR_UNSUPPORTED when it is missing), 3 = variant 1 with the large kernel's workgroups de-phased (a measured experiment). */
int gnms_profile_sgemm(const float* A, const float* B, float* D, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldd,
                       int variant, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Classical hard NMS (lib/nms)
 * ------------------------------------------------------------------------------------------- */

/* EXACT reference symbol and contract (lib/nms/gpu_nms.hpp:1-2, lib/nms/nms_kernel.cu:91-144):
 * host pointers, boxes_host is boxes_num x boxes_dim fp32 pre-sorted by descending score,
 * keep_out holds boxes_num ints, blocking.  +1-pixel IoU (:24-32), strict '>' (:71).
 * boxes_num <= GNMS_MAX_BOXES runs the layer's leader scan (one workgroup per super-block); larger inputs -- the reference's `use_nms and
 * synced` inference branch (lib/rpn_util.py:1268) feeds every anchor, > 100 k -- are processed in chunks of GNMS_MAX_BOXES (round 6): what the
 * boxes kept so far suppress in the chunk straight from the boxes, the chunk's own bit matrix, the same scan, the kept boxes appended -- the
 * reference's keep list with n^2 / (2 chunks) + n * kept pair decisions and 33 MB of device memory where the n x n mask is 2 GB at 126 720
 * boxes.  Limit: boxes_num <= 262144; above it *num_out = 0 and gnms_last_error() says so (the Python wrapper gpu_nms raises). */
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
          float nms_overlap_thresh, int device_id);

/* device-pointer, stream-ordered variant: boxes [n][boxes_dim] sorted by score; keep [n] int32; num_out device int32.
 * workspace: gnms_nms_workspace_bytes(n). */
size_t gnms_nms_workspace_bytes(int n);
int gnms_nms_sorted(const float* boxes, int n, int boxes_dim, float thresh, int32_t* keep, int32_t* num_out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* the same scan with the pixel convention and the comparison as parameters: areas and overlaps use (x2 - x1 + shift); keep_le = 0
 * suppresses when IoU > thresh (`_nms`), keep_le = 1 keeps only IoU <= thresh -- lib/nms_others.py:119-150 girshick_nms(dets, thresh,
 * shift) on score-sorted boxes (a NaN overlap then suppresses, as `np.where(ovr <= thresh)` does). */
int gnms_nms_sorted_shift(const float* boxes, int n, int boxes_dim, float thresh, float shift, int keep_le, int32_t* keep,
                          int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream);
/* the same for float64 boxes, every operation in double: girshick_nms computes in the dtype of `dets` (lib/nms_others.py:119-150),
 * float64 for the arrays the reference's own test feeds it (test/test_differentiable_nms_forward.py:111-114) */
int gnms_nms_sorted_shift_f64(const double* boxes, int n, int boxes_dim, double thresh, double shift, int keep_le, int32_t* keep,
                              int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream);

/* lib/nms_others.py:6-116 navneeth_soft_nms(boxes, sigma, Nt, threshold, method, shift): Soft-NMS with the reference's slot
 * bookkeeping.  boxes [n][boxes_dim] (x1 y1 x2 y2 score ...), fp64 when is_fp64 else fp32, NOT modified (the reference decays the
 * scores and swaps the rows of its argument in place).  method 0 hard, 1 linear, 2 gaussian.  keep [n] int64: the kept ORIGINAL
 * indices in the reference's slot order (`keep_orig[:N]`), *num_out (device int32) of them.  One workgroup; n <= GNMS_MAX_BOXES.
 * workspace: gnms_soft_nms_workspace_bytes(n), 256-byte aligned. */
size_t gnms_soft_nms_workspace_bytes(int n);
int gnms_soft_nms(const void* boxes, int n, int boxes_dim, int is_fp64, double sigma, double Nt, double threshold, int method, double shift,
                  int64_t* keep, int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * After-NMS AP loss (lib/loss/aploss.py:14-87 backpropAPLoss.forward, called per image on the rescored
 * scores at lib/loss/rpn_3d.py:1117-1131): the immediate consumer of gnms_forward's `prob`.
 *   logits, targets: [B][N] fp32 (image b uses its first counts[b] entries, all N when counts is NULL).
 *   loss: [B] = 1 - mean precision of the positives; grad: [B][N] = d loss / d logits, both produced by the
 *   forward pass as in the reference (:69-78; its backward only scales grad by the incoming gradient, :80-85).
 *   An image without a positive (max(targets) <= 0, :26-28) gets loss 0 and a zero gradient.
 *   delta is 1.0 whatever the caller of the reference passes (:16).  N <= GNMS_MAX_BOXES.  Up to 2047 boxes per image one workgroup per
 *   image does everything in LDS (no scratch); larger images take a stream-ordered temporary of 8 N words per image and spread the
 *   positives over the machine (four launches).
 * ------------------------------------------------------------------------------------------------ */
#define GNMS_APLOSS_MAX_BOXES GNMS_MAX_BOXES
int gnms_aploss(const float* logits, const float* targets, int B, int N, const int32_t* counts, float positive_label,
                float negative_label, float* loss, float* grad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * In front of the layer: decode and selection of the boxes the NMS sees, on the device.
 * ------------------------------------------------------------------------------------------------ */
/* lib/rpn_util.py:872-934 bbox_transform_inv.  anchors [A][4] (x1 y1 x2 y2), deltas [B][A][4] (dx dy dw dh; NOT modified,
 * the reference scales its argument in place, :903-913), means / stds: HOST pointers to 4 floats or NULL; out [B][A][4]. */
int gnms_bbox_transform_inv(const float* anchors, const float* deltas, int B, int A, const float* means, const float* stds,
                            float* out, void* stream);
/* lib/loss/rpn_3d.py:731-737 (and lib/rpn_util.py:1258-1266): per image the candidates' scores sorted descending (stable: ties
 * keep candidate order), the first min(K, #candidates) selected.  scores [B][A]; candidates [B][F] int32 indices into [0, A)
 * with candidate_counts [B] (NULL: all F), or candidates == NULL: every one of the A boxes is a candidate.  More than GNMS_MAX_BOXES
 * candidates (all ~127k anchors of an image at inference): a radix pre-selection first leaves exactly the K the stable sort would put
 * first (then K <= GNMS_MAX_BOXES; a stream-ordered temporary of B * (K + 1) ints).  From 4096 candidates on, while B * ceil(F / 8192)
 * workgroups fit the device and the stream is not being captured, several workgroups per image do the selection together: they use ONE
 * scratch per device that the library keeps between the calls (about 12 MiB + 16 bytes per selected box; grown with a blocking
 * hipFree / hipMalloc when a call needs more), and launches of that kind are ordered across streams by an event per device.
 * Outputs (any may be NULL), padded behind sel_count[b]: sel_index [B][K] int64 (-1), sel_scores [B][K] (0),
 * sel_boxes [B][K][4] gathered from boxes [B][A][4] (0) -- the padded layout gnms_forward_with_iou2d takes with counts. */
int gnms_select_topk(const float* scores, int B, int A, const int32_t* candidates, int F, const int32_t* candidate_counts, int K,
                     const float* boxes, int64_t* sel_index, int32_t* sel_count, float* sel_scores, float* sel_boxes, void* stream);
/* lib/loss/rpn_3d.py:746-768 ("projected" 2D boxes): params [B][N][7] = x y z w h l ry -> cuboid corners (lib/math_3d.py:364-435)
 * -> projected with p2 [B][16] (4x4 row-major, lib/math_3d.py:47-72) -> min/max over the corners -> times scale[b] (NULL: 1). */
int gnms_project_boxes3d(const float* params, const float* p2, const float* scale, int B, int N, float* boxes2d, void* stream);

/* lib/loss/rpn_3d.py:801-825: the best box per ground truth after the NMS.
 * score[i][j] = 0.5 * (1 + GIoU3D(pred_i, gt_j)) * IoU2D(pred_i, gt_j); best[j] = first argmax over the predictions, kept if
 * score > beta (self.best_target_box_beta); targets[b][best] = 1, everything else 0.
 * pred_params [B][N][7], gt_params [B][M][7] = x y z w h l ry; pred_boxes2d [B][N][4], gt_boxes2d [B][M][4]; counts NULL = all.
 * Outputs (any may be NULL): best_index [B][M] int64 (-1: none / below beta), best_score [B][M], targets [B][N] fp32. */
int gnms_best_targets(const float* pred_params, const float* pred_boxes2d, const float* gt_params, const float* gt_boxes2d, int B, int N,
                      int M, const int32_t* pred_counts, const int32_t* gt_counts, float beta, int64_t* best_index, float* best_score,
                      float* targets, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GROOMED_NMS_HIP_H */
