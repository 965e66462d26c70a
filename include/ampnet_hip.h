/* ampnet_hip.h -- C ABI of libampnet_hip.so: the MI355X (gfx950) implementation of the AMP-Net
 * per-window hot path of marionacaros/3D-semantic-segmentation-AMP-Net.
 *
 * The reference has no FFI: its boundary for this path is Python (nn.Module.forward signatures,
 * function signatures, state_dict keys).  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root); the Python package
 * `3d-semantic-segmentation-amp-net_amd/` binds them with ctypes behind modules of the reference's
 * names (see INTEGRATION.md for the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; caller owns every buffer;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and never synchronised;
 *   - return value 0 = ok, negative = error (AMPNET_E_*); ampnet_last_error() gives the text
 *     (thread-local).  No exception crosses the ABI;
 *   - the library keeps no state between calls except the thread-local error string.
 *   - float tensors are fp32, row-major, point-major: activations are [rows, channels].
 *
 * Window batching: the reference calls its encoder W times per step, each time on the B windows that
 * occupy cluster slot w (train_pointnet-attention.py:396-410); BatchNorm statistics are therefore
 * per slot.  Here all Q = B*W windows go through one launch sequence; window q = b*W + w (sample-major,
 * the order of lo_feats in the reference, train_pointnet-attention.py:415-417) and `n_slots` = W tells
 * the kernels that windows with equal q % n_slots share batch statistics.  Eval mode ignores slots.
 */
#ifndef AMPNET_HIP_H
#define AMPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMPNET_ABI_VERSION 4

enum {
    AMPNET_OK = 0,
    AMPNET_E_ARG = -1,        /* bad shape / null pointer / unsupported size */
    AMPNET_E_LAUNCH = -2,     /* hipGetLastError() after a launch            */
    AMPNET_E_WORKSPACE = -3,  /* workspace too small                         */
    AMPNET_E_DEVICE = -4      /* not a gfx950 device / no device             */
};

int ampnet_abi_version(void);
const char *ampnet_last_error(void);

/* ---- a1: farthest-point sampling ---------------------------------------------------------------
 * replaces utils/utils.py:889-933 `fps(pc, n_samples)` (driver data_proc/sample_fps.py:23-31).
 * xyz: [n_clouds, n, ld] fp32, columns 0..2 = x,y,z (ld >= 3 lets the caller pass whole rows).
 * idx: [n_clouds, s] int32, selection order; idx[c][0] == 0 (utils.py:907-908).
 * Bit-exact with the reference: float32 ((dx*dx + dy*dy) + dz*dz), running minimum, first maximum.
 * 1 <= s <= n <= AMPNET_FPS_MAX_POINTS.  Clouds of up to AMPNET_FPS_RESIDENT_MAX points are register-resident and need no
 * workspace (workspace may be NULL); larger clouds (the raw 100 x 100 m tiles of data_proc/sample_fps.py:23-26) stream
 * their coordinates and running minima from a structure-of-arrays copy in `workspace` (ampnet_fps_workspace_bytes(n_clouds, n) bytes).  */
#define AMPNET_FPS_RESIDENT_MAX 16384
#define AMPNET_FPS_MAX_POINTS (1 << 24)
size_t ampnet_fps_workspace_bytes(int n_clouds, int n);
int ampnet_fps_f32(const float *xyz, int n_clouds, int n, int ld, int s, int32_t *idx, void *workspace,
                   size_t workspace_bytes, void *stream);

/* The same sampling for a RAGGED batch -- what one stage of data_proc/sample_fps.py:12-34 does to a directory of files of unequal size,
 * in ONE launch: cloud b = rows cloud_off[b] .. cloud_off[b + 1] of rows [total_rows, ld]; its out_off[b + 1] - out_off[b] samples (clamped
 * to the cloud's size) go to idx[out_off[b] ..) as indices RELATIVE to the cloud, selection order, first = 0.  cloud_off / out_off:
 * DEVICE int32 arrays of n_clouds + 1 ascending offsets; max_n = the largest cloud (picks the kernel: every cloud of the launch runs the
 * variant built for max_n, so callers bucket files by size class -- data_proc/sample_fps.py of the package does).  Same arithmetic, same
 * tie rule, bit-identical to ampnet_fps_f32 cloud by cloud.  max_n > AMPNET_FPS_RESIDENT_MAX needs ampnet_fps_ragged_workspace_bytes().
 * HARD PRECONDITION: max_n >= every cloud of the launch (registers and LDS are sized for it; the offsets live on the device, so the host
 * cannot check).  A cloud that breaks it is refused inside the kernel: the first index of its output range reads -1, nothing else is written. */
size_t ampnet_fps_ragged_workspace_bytes(int total_rows, int max_n);
int ampnet_fps_ragged_f32(const float *rows, int ld, const int32_t *cloud_off, const int32_t *out_off, int n_clouds, int total_rows,
                          int max_n, int32_t *idx, void *workspace, size_t workspace_bytes, void *stream);

/* diagnostic build of the 8192-point kernel (one cloud): stamps[4 r + {0,1,2,3}] = s_memtime of thread 0 after the update,
 * after the barrier, after the slot fold and after the winner's coordinates landed in round r (FPS round anatomy).
 * The stamps go to a buffer nothing else reads; idx is the ordinary result.                                              */
int ampnet_fps_round_stamps(const float *xyz, int n, int ld, int s, int32_t *idx, unsigned long long *stamps, void *stream);

/* rows gather: out[c][i][:] = src[c][idx[c][i]][:]  (the `pc[sample_inds]` of utils.py:933)         */
int ampnet_gather_rows_f32(const float *src, const int32_t *idx, int n_clouds, int n, int ld, int s,
                           float *out, void *stream);

/* ---- parameter tables -----------------------------------------------------------------------------
 * Model parameters cross the ABI as HOST arrays of DEVICE pointers, one entry per state_dict tensor of the
 * reference modules, in the fixed order below (the order of `params.ENC_PARAMS` / `ENC_BUFFERS` /
 * `HEAD_PARAMS` / `HEAD_BUFFERS` in the Python package; ampnet_*_name(i) returns the state_dict key):
 *   encoder params  (52): input_transform.{conv_1,conv_2,conv_3}.weight, bn_1..5.{weight,bias},
 *                         fc_1.weight, fc_2.weight, fc_3.{weight,bias}; feature_transform.<same 17>;
 *                         conv_1..6.weight; bn_1..6.{weight,bias}          (pointnetAtt.py:14-26, 66-78)
 *   encoder buffers (32): (running_mean, running_var) of input_transform.bn_1..5, feature_transform.bn_1..5,
 *                         bn_1..6
 *   head params     (18): fc1.{weight,bias}, fc2.{weight,bias}, attention.in_proj_{weight,bias},
 *                         attention.out_proj.{weight,bias}, conv_2.{weight,bias}, conv_3.{weight,bias},
 *                         conv_4.{weight,bias}, bn_2.{weight,bias}, bn_3.{weight,bias}   (pointnetAtt.py:160-174)
 *   head buffers     (4): (running_mean, running_var) of bn_2, bn_3
 * Only the AMP-Net configuration is built: point_dimension 3, 9 features, global 256, local 64,
 * 8 heads, <= 8 classes (train_pointnet-attention.py:110-118).                                      */
int ampnet_table_count(int table);                 /* table: 0 enc params, 1 enc buffers, 2 head params, 3 head buffers */
const char *ampnet_table_name(int table, int i);
long ampnet_table_numel(int table, int i);

/* ---- a2/a3: encoder forward ---------------------------------------------------------------------
 * replaces BasePointNet.forward (pointNet/model/pointnetAtt.py:80-112; TransformationNet.forward :28-47) for the
 * W encoder calls of one step (train_pointnet-attention.py:396-410) at once.
 *   x          [total_rows, 9]      the windows back to back; window q = rows win_off[q] .. win_off[q+1]
 *   win_off    [Q + 1] int32 (device); max_rows = the largest window
 *   n_slots    windows q with equal q % n_slots share BatchNorm batch statistics (train); Q = B * n_slots with
 *              q = b * n_slots + w.  Ignored in eval mode (running statistics).
 *   local      [total_rows, 64]     local_point_features (:97)
 *   global_feat[Q, 256]             max-pooled global feature (:104-106), row q
 *   feat_T     [Q, 64, 64]          feature_transform (:94); in_T [Q, 3, 3] input_transform (:84), may be NULL.
 *              Row order of feat_T / in_T: eval: q.  train: slot-major, row (q % n_slots) * (Q / n_slots) + q / n_slots,
 *              so the LAST slot's B matrices (what the reference's reg loss uses, train_pointnet-attention.py:463)
 *              are the last B rows.
 *   train != 0: batch statistics, running statistics updated in place (momentum 0.1, one update per slot in
 *              slot order, like W encoder calls); the workspace then holds what ampnet_encoder_bwd_f32 needs.
 * Eval logits built on these outputs match the reference CPU forward within 1e-3 (tests/test_forward_gpu.py). */
size_t ampnet_encoder_workspace_bytes(int Q, int n_slots, int total_rows, int max_rows, int train);
int ampnet_encoder_fwd_f32(const float *const *params_host, float *const *buffers_host, const float *x,
                           const int32_t *win_off, int Q, int n_slots, int total_rows, int max_rows, int train,
                           float *local, float *global_feat, float *feat_T, float *in_T, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ---- a2/a3 backward -----------------------------------------------------------------------------------
 * replaces autograd's backward through BasePointNet.forward (the reference: loss.backward(),
 * train_pointnet-attention.py:467) for all windows of a step.  Must follow a train-mode
 * ampnet_encoder_fwd_f32 on the same x / win_off / Q / n_slots with `fwd_workspace` untouched in between.
 *   grads_host [52] device pointers, same order as params_host; every gradient is WRITTEN (not accumulated)
 *   local, feat_T   the forward's outputs (read)
 *   d_local  [total_rows, 64] or NULL; d_global [Q, 256] (row q); d_feat_T [Q, 64, 64] (slot-major rows, like
 *            feat_T in train mode) or NULL: gradients of the loss wrt the three forward outputs.         */
size_t ampnet_encoder_bwd_workspace_bytes(int Q, int n_slots, int total_rows, int max_rows);
int ampnet_encoder_bwd_f32(const float *const *params_host, float *const *grads_host, const float *x,
                           const int32_t *win_off, int Q, int n_slots, int total_rows, int max_rows,
                           const float *local, const float *d_local, const float *d_global, const float *d_feat_T,
                           const float *feat_T, void *fwd_workspace, size_t fwd_workspace_bytes, void *bwd_workspace,
                           size_t bwd_workspace_bytes, void *stream);

/* ---- a4 (+a6 forward): attention head -------------------------------------------------------------
 * replaces SegmentationWithAttention.forward (pointNet/model/pointnetAtt.py:176-209) and, optionally, the
 * loss / prediction lines of train_loop (train_pointnet-attention.py:138,445-450).
 *   gl         [B * W, 256]   window tokens, row q = b * W + w (the reference's gl_feats[w, b, :])
 *   lo         [total_rows, 64] local features; sample b owns rows of windows b*W .. b*W + W - 1 back to back
 *                             (the reference's lo_feats[b]); every sample has the same np_cluster list, so
 *                             total_rows = B * P and window (b, w) has np_cluster[w] rows (win_off says so)
 *   centroids  [B, W, 2]
 *   key_pad_mask [B, W] bytes, non-zero = ignore that cluster token as a key; NULL = no mask (:189)
 *   logits     [B, n_classes, P]                                                             (:207-209)
 *   train != 0: batch statistics for bn_2 / bn_3 (over all B * P rows), running stats updated, dropout with
 *              probability drop_p on the attention weights and after both ReLUs (:204-206) from the
 *              counter-based generator keyed by `seed`; eval: running statistics, no dropout.
 *   targets    [B, P] int64 (-1 = ignore), class_w [n_classes], preds [B, P] int64, loss_out [2]
 *              (weighted-mean CE, sum of weights): all optional (NULL).                              */
size_t ampnet_head_workspace_bytes(int B, int W, int total_rows, int max_rows, int n_classes, int train);
/* The eval forward for SEVERAL FILES in one launch sequence (the reference's test loop, test_pointnet_att_segmen.py:127-181, runs one file
 * of <= W ragged clusters per step at batch 1): file f owns the window slots f * W .. f * W + W - 1, its real clusters first; unused slots
 * are windows of zero rows (win_off repeats its value) with key_pad_mask[f, w] = 1, so a file's attention sees exactly its own clusters.
 * Files differ in their point counts: logits [n_classes, total_rows] and preds [total_rows] run over the concatenated rows.
 * workspace: ampnet_head_workspace_bytes(n_files, W, total_rows, max_rows, n_classes, 0).  Results are identical, file by file, to
 * ampnet_head_fwd_f32 with B = 1 (tests/test_inference_gpu.py).                                                                   */
int ampnet_head_fwd_files_f32(const float *const *params_host, float *const *buffers_host, const float *gl, const float *lo,
                              const float *centroids, const int32_t *win_off, const uint8_t *key_pad_mask, int n_files, int W,
                              int total_rows, int max_rows, int n_classes, float *logits, long long *preds, void *workspace,
                              size_t workspace_bytes, void *stream);
int ampnet_head_fwd_f32(const float *const *params_host, float *const *buffers_host, const float *gl,
                        const float *lo, const float *centroids, const int32_t *win_off,
                        const uint8_t *key_pad_mask, int B, int W, int total_rows, int max_rows, int n_classes,
                        int train, float drop_p, uint32_t seed, float *logits, const long long *targets,
                        const float *class_w, long long *preds, float *loss_out, void *workspace,
                        size_t workspace_bytes, void *stream);

/* ---- a4 backward ---------------------------------------------------------------------------------------
 * autograd backward of SegmentationWithAttention.forward given dlogits [B, n_classes, P].  Must follow a
 * train-mode ampnet_head_fwd_f32 with the same arguments (drop_p, seed included) and an untouched fwd_workspace.
 *   grads_host [18] device pointers in the order of the head parameters; every gradient is WRITTEN
 *   d_lo [total_rows, 64] = dL/d(lo), d_gl [B * W, 256] = dL/d(gl) (row b * W + w); centroids get no gradient. */
size_t ampnet_head_bwd_workspace_bytes(int B, int W, int total_rows, int max_rows, int n_classes);
int ampnet_head_bwd_f32(const float *const *params_host, float *const *grads_host, const float *lo,
                        const float *centroids, const int32_t *win_off, int B, int W, int total_rows, int max_rows,
                        int n_classes, float drop_p, uint32_t seed, const float *dlogits, float *d_lo, float *d_gl,
                        void *fwd_workspace, size_t fwd_workspace_bytes, void *bwd_workspace,
                        size_t bwd_workspace_bytes, void *stream);

/* ---- f4: the GRU variant of the sequence model ---------------------------------------------------------------
 * replaces SegmentationWithGRU.forward (pointNet/model/pointnetAtt.py:212-258: nn.GRU(256 -> 64, batch_first, h0 = 0) over the W window
 * tokens of a sample, the hidden state of step w repeated over the points of window w, cat with the local features, conv_2 / bn_2 /
 * conv_3 / bn_3 / conv_4 with two dropouts) as pointNet/rnn/train_pointnetGRU.py:335-441 drives it; the repeat + cat is never built.
 *   params_host  [14] device pointers in state_dict order: gru_global.weight_ih_l0 [192,256], weight_hh_l0 [192,64], bias_ih_l0 [192],
 *                bias_hh_l0 [192], conv_2.weight [128,128], conv_2.bias, conv_3.weight [64,128], conv_3.bias, conv_4.weight [C,64],
 *                conv_4.bias, bn_2.weight, bn_2.bias, bn_3.weight, bn_3.bias
 *   buffers_host [4]  bn_2.running_mean, bn_2.running_var, bn_3.running_mean, bn_3.running_var (updated in train mode)
 *   gl [B * W, 256] window tokens (row b * W + w = global_seq[b, w, :]); lo, win_off, logits, targets, class_w, preds, loss_out,
 *   train, drop_p, seed: as ampnet_head_fwd_f32.  There is no key-padding mask: the reference runs the GRU over every window.   */
size_t ampnet_gru_head_workspace_bytes(int B, int W, int total_rows, int max_rows, int n_classes, int train);
int ampnet_gru_head_fwd_f32(const float *const *params_host, float *const *buffers_host, const float *gl, const float *lo,
                            const int32_t *win_off, int B, int W, int total_rows, int max_rows, int n_classes, int train, float drop_p,
                            uint32_t seed, float *logits, const long long *targets, const float *class_w, long long *preds,
                            float *loss_out, void *workspace, size_t workspace_bytes, void *stream);
/* autograd backward of the above given dlogits [B, n_classes, P] (must follow a train-mode forward with the same arguments and an
 * untouched fwd_workspace): grads_host [14] are WRITTEN; d_lo [total_rows, 64], d_gl [B * W, 256].                                */
size_t ampnet_gru_head_bwd_workspace_bytes(int B, int W, int total_rows, int max_rows, int n_classes);
int ampnet_gru_head_bwd_f32(const float *const *params_host, float *const *grads_host, const float *gl, const float *lo,
                            const int32_t *win_off, int B, int W, int total_rows, int max_rows, int n_classes, float drop_p, uint32_t seed,
                            const float *dlogits, float *d_lo, float *d_gl, void *fwd_workspace, size_t fwd_workspace_bytes,
                            void *bwd_workspace, size_t bwd_workspace_bytes, void *stream);

/* ---- f4: classification head on the window tokens ------------------------------------------------------------------------
 * replaces ClassificationWithAttention.forward (pointNet/model/pointnetAtt.py:115-151): MultiheadAttention over the W tokens of a sample,
 * conv_1 (Conv1d(num_w -> 1, 1)) over the attention output RE-VIEWED as [B, W, 256] (the reference views the sequence-first tensor, it
 * does not transpose it; restated literally), fc_2 -> bn_2 (over the B rows) -> ReLU -> fc_3.  No reference script reaches this module
 * (train_pointnet-attention.py:440-442 leaves the classification branch without a model call); built for the module's own contract.
 *   params_host  [12] attention.in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, conv_1.weight [1, W, 1], conv_1.bias [1],
 *                fc_2.weight [128, 256], fc_2.bias, fc_3.weight [C, 128], fc_3.bias, bn_2.weight, bn_2.bias
 *   buffers_host [2]  bn_2.running_mean, bn_2.running_var
 *   gl [B * W, 256] (row b * W + w = gl_feats[w, b, :]); key_pad_mask [B, W] or NULL; out [B, n_classes];
 *   attn_weights [B, W, W] or NULL: the attention probabilities averaged over the heads (after dropout in train mode), need_weights=True */
size_t ampnet_cls_head_workspace_bytes(int B, int W);
int ampnet_cls_head_fwd_f32(const float *const *params_host, float *const *buffers_host, const float *gl, const uint8_t *key_pad_mask, int B,
                            int W, int n_classes, int train, float drop_p, uint32_t seed, float *out, float *attn_weights, void *workspace,
                            size_t workspace_bytes, void *stream);
/* autograd backward given d_out [B, n_classes] (after a train-mode forward with the same arguments): grads_host [12] WRITTEN, d_gl [B * W, 256] */
size_t ampnet_cls_head_bwd_workspace_bytes(int B, int W);
int ampnet_cls_head_bwd_f32(const float *const *params_host, float *const *grads_host, const float *gl, int B, int W, int n_classes, float drop_p,
                            uint32_t seed, const float *d_out, float *d_gl, void *fwd_workspace, size_t fwd_workspace_bytes,
                            void *bwd_workspace, size_t bwd_workspace_bytes, void *stream);

/* ---- a6: loss recipe (train_pointnet-attention.py:138,445,463-467) ------------------------------------
 * reg = || I - F F^T ||_F over the whole stack feat_T [n, 64, 64] (torch.norm of a 3-D tensor = Frobenius over
 * all elements).  G [n, 64, 64] (optional) receives I - F F^T for the backward; part [n] is scratch.
 * ampnet_reg_loss_bwd_f32:  d_feat_T += coef * d(reg)/d(feat_T).
 * ampnet_ce_bwd_f32:        dlogits = grad_scale * d(ce)/d(logits) for the weighted-mean CE the head forward
 *                           returned in loss2 = {ce, sum of weights} (ignore_index -1).                     */
int ampnet_reg_loss_fwd_f32(const float *feat_T, int n, float *reg_out, float *G, float *part, void *stream);
int ampnet_reg_loss_bwd_f32(const float *feat_T, const float *G, const float *reg, float coef, int n, float *d_feat_T,
                            void *stream);
/* the same gradient WRITTEN into a stack d_feat_T_stack [n_total, 64, 64] whose last n matrices are the regularised ones (the reference
 * regularises the feature transform of the last cluster only, train_pointnet-attention.py:445): zeros in the first n_total - n, the
 * gradient (no accumulate) in the rest -- the gradient tensor of all feature transforms without a separate zero fill.                */
int ampnet_reg_loss_bwd_stack_f32(const float *feat_T, const float *G, const float *reg, float coef, int n, int n_total,
                                  float *d_feat_T_stack, void *stream);
int ampnet_ce_bwd_f32(const float *logits, const long long *targets, const float *class_w, const float *loss2,
                      float grad_scale, int B, int C, int P, float *dlogits, void *stream);

/* ---- a11 on the device: the counts behind get_accuracy / get_iou_obj (utils/get_metrics.py:6-31) ----------------------------------
 * counts[t * C + p] = number of points with target t and prediction p, counts[C * C] = number of ignored points (target -1 = padding,
 * what rm_padding removes, utils/utils.py:14-19); counts is WRITTEN ([C * C + 1] int64).  accuracy = trace / kept; IoU of label c =
 * counts[c][c] / (row sum c + column sum c - counts[c][c]).  Lets a training driver keep predictions on the device: no 2 x 9.4 MB
 * download and no synchronisation per step.                                                                                      */
/* the key-padding mask of train_loop (train_pointnet-attention.py:428-431): targets [B, P] int64 (cluster-major, -1 = padded point),
 * mask[b, w] = 1 iff targets[b, i * W + w] == -1 for every i -- the reference's literal `(targets_pc.view(B, -1, W) == -1).all(dim=1)`.
 * W <= 32, P % W == 0.  One launch instead of five torch launches per step.                                                         */
int ampnet_pad_mask_i64(const long long *targets, int B, int P, int W, uint8_t *mask, void *stream);
int ampnet_confusion_i64(const long long *preds, const long long *targets, long long n, int n_classes, long long *counts, void *stream);

/* ---- the token-level products of the backward (one row per window: the attention projections, the T-Net FC layers) ---------------------
 * C [M, N] (+)= op(A) op(B), op(A) = A [M, K] (trans_a = 0) or A^T with A [K, M]; op(B) = B [K, N] (trans_b = 0) or B^T with B [N, K];
 * row-major with leading dimensions lda / ldb / ldc; accumulate != 0 adds to C.  k_scale (optional, [K]): op(A)[m][k] is multiplied by k_scale[k]
 * as it is loaded, i.e. C = op(A) diag(k_scale) op(B) (the per-slot matrices W^T diag(P2) W of the pooled layers' backward); it also scales the
 * row sums.  row_sums (optional, [M]) receives sum_k op(A)[m][k]:
 * the bias gradient G^T 1 that rides in a weight-gradient product dW = G^T X (reference: autograd of nn.Linear / nn.MultiheadAttention
 * as train_pointnet-attention.py:463-467 calls it; this is not a reference interface, it is exported so that the kernel the backward
 * calls ~11 times per step can be tested on its own).  fp32 matrix cores, K split over the 16 waves of a workgroup in a fixed order:
 * bitwise reproducible.                                                                                                             */
int ampnet_small_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C,
                          int ldc, int accumulate, float *row_sums, const float *k_scale, void *stream);

/* ---- a8 on the device: the input pipeline of train_loop in one kernel --------------------------------------------
 * replaces the host augmentation of train_pointnet-attention.py:390-405 (shuffle_clusters utils/utils.py:620-632,
 * rotate_point_cloud_z :582-604, shuffle_data :607-617) and the [B, N, 9, W] -> [B, W, N, 9] re-layout.
 *   pc [B, N, 9, W] float32 and targets [B, N, W] int64 (may be NULL with t_out) as collate_seq_padd returns them, on the device
 *   cluster_perm [W], point_perm [W, N] (NULL = identity) int32 on the device: x_out[b, w, n] = pc[b, point_perm[w, n], :, cluster_perm[w]]
 *   rotate != 0: (x, y, z) <- (x c - y s, x s + y c, z) in float64, rounded to float32 (numpy's float64 dot of the reference)
 *   x_out [B, W, N, 9], t_out [B, W, N]                                                                         */
int ampnet_augment_f32(const float *pc, const long long *targets, const int32_t *cluster_perm, const int32_t *point_perm,
                       double cos_a, double sin_a, int rotate, int B, int N, int W, float *x_out, long long *t_out, void *stream);

/* ---- a9 + a8 on the device: collate_seq_padd's resampling / padding fused into the augmentation kernel --------------------
 * replaces, together with the package's collate_seq_ragged (pointNet/collate_fns.py), the host work of pointNet/collate_fns.py:33-45
 * (every sample gathered to exactly N = 2048 points, the cluster axis padded to W = 9 by replicating the last cluster, targets padded
 * with -1) -- in the reference this runs in the DataLoader workers on 42 MB per batch of 64 and is what a train epoch waits for
 * (bench.py: train_att_epoch).  The workers hand over the RAGGED samples and the resampling map instead:
 *   pts     float32: sample b = [n_b, 9, w_b] (as LidarKmeansDataset returns it) at element offset meta[b][2]
 *   labels  int8   : sample b = [n_b, w_b] segmentation labels 0 .. 4 at element offset meta[b][3]
 *   idx     int32 [B, N]: padded row p of sample b is its own row idx[b][p] (the draws of collate_seq_padd: torch.randint when
 *                         n_b < N, random.sample when n_b > N, the identity when n_b == N)
 *   meta    int32 [B, 4]: n_b, w_b, pts offset, labels offset
 * and the kernel writes what ampnet_augment_f32 would have written for the padded batch, bit for bit (tests/test_augment_gpu.py):
 *   x_out[b, w, n, f] = pts_b[idx[b][pp], f, min(cw, w_b - 1)],  t_out[b, w, n] = cw < w_b ? labels_b[idx[b][pp], cw] : -1,
 *   cw = cluster_perm[w], pp = point_perm[w, n] (NULL = n), then the z-rotation as above.  All arrays on the device.          */
int ampnet_collate_augment_f32(const float *pts, const signed char *labels, const int32_t *idx, const int32_t *meta,
                               const int32_t *cluster_perm, const int32_t *point_perm, double cos_a, double sin_a, int rotate,
                               int B, int N, int W, float *x_out, long long *t_out, void *stream);

/* ---- k-NN grouping of FPS centres (BUILD-DEFINED; BASELINE.json north_star / config 5) ------------------------------
 * The reference has no k-NN or ball query: nothing is replaced, parity against it is "unpinned"; the spec
 * below is pinned by oracle/fps_oracle.py:knn_indices.
 *   xyz      [n_clouds, n, ld] float32 (first 3 columns used), n * 12 bytes <= 144 KB (n <= 12288)
 *   centres  [n_clouds, s] int32 point indices (e.g. the output of ampnet_fps_f32)
 *   out      [n_clouds, s, k] int32: for each centre the k points with the smallest (distance, index), ascending;
 *            distance = float32 ((dx*dx + dy*dy) + dz*dz), no fused multiply-add; the centre itself comes first   */
int ampnet_knn_f32(const float *xyz, int n_clouds, int n, int ld, const int32_t *centres, int s, int k, int32_t *out,
                   void *stream);

/* ---- size-constrained k-means: the window grouping step in front of the path --------------------------
 * replaces the calls of the third-party k_means_constrained.KMeansConstrained at data_proc/3_kmeans.py:78-82 (size_min = size_max =
 * n_points, n_init 5, max_iter 10, tol 1e-2, features x, y, NDVI) and utils/utils.py:500-505 (size_min only).  That package is not part
 * of the reference repository: parity is UNPINNED; the algorithm below is this build's spec (csrc/kmeans.hip, oracle/kmeans_oracle.py):
 * farthest-point seeding, greedy capacity-constrained assignment in ascending (distance, point, cluster) order (every cluster first gets
 * size_min points, the rest go where capacity size_max allows), means, stop at centre shift <= tol * mean feature variance, best of n_init.
 *   feat     [n, 3] float32 device     labels  [n] int32 out     centres [k, 3] float32 out     inertia: device double out (may be NULL)
 *   1 <= k <= 32, k <= n <= 65536, size_min * k <= n <= size_max * k                                                                   */
size_t ampnet_kmeans_workspace_bytes(int n, int k);
int ampnet_kmeans_balanced_f32(const float *feat, int n, int k, int size_min, int size_max, int n_init, int max_iter, float tol,
                               uint32_t seed, int32_t *labels, float *centres, double *inertia, void *workspace, size_t workspace_bytes,
                               void *stream);

/* ---- global-batch BatchNorm under data parallelism (process-wide; SURVEY section 8(e) option A) ---------------------------------
 * The reference is single-device: its BatchNorm layers see the whole batch.  With a collective registered and world_size > 1,
 * every TRAIN-mode BatchNorm inside ampnet_encoder_fwd_f32 / _bwd_f32, ampnet_head_* and ampnet_gru_head_* uses the statistics of the
 * global batch: the forward all-gathers the per-slot (rows, mean, M2) of every rank and merges them (Chan), the backward all-reduces
 * the per-slot (sum dy, sum dy zhat, rows).  Running statistics are then identical on every rank.  The library calls `fn` on the host,
 * between two launches:
 *   op AMPNET_COLLECTIVE_ALLGATHER      recv[k * n_floats .. ] = rank k's send[0 .. n_floats)      (recv holds world_size * n_floats)
 *   op AMPNET_COLLECTIVE_ALLREDUCE_SUM  send == recv: element-wise float32 sum over the ranks, in place
 * send / recv point into `scratch` (device memory of ampnet_collective_scratch_bytes(world_size) bytes the caller owns and keeps
 * alive); the exchange must be ordered after the work already enqueued on `stream` and before what is enqueued after fn returns
 * (torch.distributed on the current stream does that).  fn returns 0 on success.  fn = NULL switches back to per-rank statistics.
 * 18 + 18 latency-bound collectives per AMP-Net step: off by default (per-rank BatchNorm is the documented deviation). */
#define AMPNET_COLLECTIVE_ALLGATHER 0
#define AMPNET_COLLECTIVE_ALLREDUCE_SUM 1
#define AMPNET_SYNC_MAX_SLOTS 32
#define AMPNET_SYNC_MAX_CHANNELS 256
typedef int (*ampnet_collective_fn)(void *ctx, int op, void *send, void *recv, size_t n_floats, void *stream);
size_t ampnet_collective_scratch_bytes(int world_size);
int ampnet_set_collective(ampnet_collective_fn fn, void *ctx, int rank, int world_size, void *scratch, size_t scratch_bytes);

/* ---- matrix-core operand precision (process-wide) -----------------------------------------------------------
 * AMPNET_PRECISION_F32 (default): v_mfma_f32_32x32x2_f32, exact fp32 products -- the mode every parity figure is quoted in.
 * AMPNET_PRECISION_BF16: the per-point layers of ampnet_encoder_fwd_f32 / ampnet_head_fwd_f32 round their MFMA operands
 * (activations after BatchNorm+ReLU, weights) to bf16 and accumulate in fp32 (v_mfma_f32_32x32x16_bf16); tensors in HBM,
 * BatchNorm statistics, loss and the whole backward stay fp32.  BASELINE.json config 3 ("bf16 MFMA MLP/attention").   */
#define AMPNET_PRECISION_F32 0
#define AMPNET_PRECISION_BF16 1
/* AMPNET_PRECISION_BF16_TRAIN: the forward of AMPNET_PRECISION_BF16 AND the fused backward of the shared per-point layers
 * (weight gradient dW = g^T a and data gradient dy = g W of one pass, csrc/pw_bwd_bf16.hip) with bf16 operands: g = dy P1 + z P2 + P3,
 * the recomputed activation a and the weights are formed in fp32 and rounded once; accumulation, BatchNorm-backward sums, the
 * K <= 12 input layers, the T-Net FC layers, the attention and every tensor in HBM stay fp32.                                   */
#define AMPNET_PRECISION_BF16_TRAIN 2
/* AMPNET_PRECISION_BF16_STORE: AMPNET_PRECISION_BF16_TRAIN, and the activations a train step keeps for its backward (the nine
 * pre-BatchNorm tensors of the encoder, z2 / z3 of the head) are STORED as bf16 (rounded once from the fp32 accumulator; the
 * BatchNorm statistics are taken before the rounding): the step moves about a third fewer HBM bytes.  Inputs, outputs (local,
 * global, feat_T, logits), gradients and parameters stay fp32.
 * ENFORCED: every train-mode ampnet_encoder_fwd_f32 / ampnet_head_fwd_f32 / ampnet_gru_head_fwd_f32 records (on the host) the mode its
 * workspace was written in; the matching *_bwd_f32 returns AMPNET_E_ARG when the storage format differs (mode 3 on one side only), or on a
 * workspace that holds no train-mode forward of this process, instead of misreading the saved activations (tests/test_bf16_gpu.py).
 * Modes 0 .. 2 share the fp32 tape: a backward in one of them may follow a forward in another.                                   */
#define AMPNET_PRECISION_BF16_STORE 3
/* AMPNET_PRECISION_F32_SPLIT ("f32x3"): fp32 results from the bf16 matrix pipe.  Every operand of the MFMA-bound per-point products
 * (the 128 -> 256 pooled layers and the 128 -> 128 layer of the forward; the Gram-form and the dense 128 x 128 fused backward) is split
 * into three bf16 terms a = a1 + a2 + a3 (three successive round-to-nearest roundings: the sum is the fp32 value exactly) and the
 * product is formed from six v_mfma_f32_32x32x16_bf16 instructions (a1 b1, a1 b2, a2 b1, a1 b3, a2 b2, a3 b1: each partial product
 * of two bf16 numbers is exact in fp32, the three dropped terms are below 2^-23 |a b|), accumulated in fp32: 6 / 16 of the fp32 MFMA
 * time at fp32 accuracy.  Tensors in HBM, BatchNorm statistics, prologues, epilogues and every other kernel are those of
 * AMPNET_PRECISION_F32 (same tape: a backward in one of the two modes may follow a forward in the other).  The parity tests of the
 * fp32 path run in this mode with the same bars (tests/conftest.py: AMPNET_TEST_PRECISION).                                       */
#define AMPNET_PRECISION_F32_SPLIT 4
int ampnet_set_matrix_precision(int mode);
int ampnet_get_matrix_precision(void);

/* ---- a12: baseline single-window PointNet segmentation, eval forward -----------------------------------------
 * replaces SegmentationPointNet.forward (module.eval()) of pointNet/model/pointnet.py:128-154 (variant 0: 1024-d,
 * convolutions with bias, T-Net on x[:, :, :3], :71) and of pointNet/model/light_pointnet_256.py:128-153 (variant 1:
 * 256-d, no conv / fc bias, T-Net on x[:, :, :2], :71).  BASELINE.json config 1 ([4, 512, 9]) is the reference's CPU
 * plumbing case: this entry exists for parity and is not tuned.
 *   layers_host  [AMPNET_POINTNET_LAYERS * 6] device pointers, per layer {weight, bias, bn.weight, bn.bias,
 *                bn.running_mean, bn.running_var}; bias and the four BatchNorm pointers may be NULL.  Layer order:
 *                0-5   base_pointnet.input_transform   conv_1 conv_2 conv_3 fc_1 fc_2 fc_3   (bn_1 .. bn_5, none)
 *                6-11  base_pointnet.feature_transform  (same)
 *                12-16 base_pointnet.conv_1 .. conv_5   (bn_1 .. bn_5)
 *                17-20 conv_1 .. conv_4                 (bn_1 .. bn_3, none)
 *   x [B, N, 9] -> logits [B, n_classes, N]; feat_T [B, 64, 64] (optional) = feature_transform            */
#define AMPNET_POINTNET_LAYERS 21
size_t ampnet_pointnet_seg_workspace_bytes(int variant, int B, int N, int n_classes);
int ampnet_pointnet_seg_fwd_f32(const float *const *layers_host, int variant, const float *x, int B, int N,
                                int n_classes, float *logits, float *feat_T, void *workspace, size_t workspace_bytes,
                                void *stream);

/* ---- a12: the same model in train mode (BASELINE.json config 1 end to end) --------------------------------------------------
 * replaces SegmentationPointNet.forward under module.train() + loss.backward() as pointNet/baseline/train_segmentation.py:274-328 drives
 * them: batch-statistics BatchNorm (running statistics updated in place through layers_host, momentum 0.1, unbiased variance), and
 * the gradients of every parameter from (dlogits [B, C, N], d_feat_T [B, 64, 64] or NULL).  The forward keeps its activations in
 * `workspace` (ampnet_pointnet_seg_train_workspace_bytes); the backward must get the same, untouched workspace.  B >= 2.
 *   grads_host   [AMPNET_POINTNET_LAYERS * 4] device pointers per layer {d weight, d bias, d bn.weight, d bn.bias}, NULL where the
 *                layer has no such parameter; every gradient is overwritten                                                    */
size_t ampnet_pointnet_seg_train_workspace_bytes(int variant, int B, int N, int n_classes);
int ampnet_pointnet_seg_train_fwd_f32(const float *const *layers_host, int variant, const float *x, int B, int N, int n_classes,
                                      float *logits, float *feat_T, void *workspace, size_t workspace_bytes, void *stream);
int ampnet_pointnet_seg_bwd_f32(const float *const *layers_host, float *const *grads_host, int variant, const float *x, int B, int N,
                                int n_classes, const float *dlogits, const float *d_feat_T, void *workspace, size_t workspace_bytes,
                                void *stream);

/* ---- f4: the baseline classification PointNet (train and eval) ---------------------------------------------------------------
 * replaces ClassificationPointNet.forward of pointNet/model/pointnet.py:100-125 (variant 0: fc 1024 -> 512 -> 256 -> n_classes with bias)
 * and of pointNet/model/light_pointnet_256.py:100-125 (variant 1: fc 256 -> 128 -> 64 without bias, -> n_classes with bias), and
 * loss.backward() through it: global feature of BasePointNet(return_local_features=False) -> relu(bn_1(fc_1)) -> relu(bn_2(fc_2)) ->
 * Dropout(p) -> log_softmax(fc_3).  Same tape as the segmentation model (csrc/baseline_train.hip); built for parity, not tuned.
 *   layers_host  [AMPNET_POINTNET_CLS_LAYERS * 6], per layer as above; layer order 0-16 = base_pointnet (as above), 17-19 = fc_1 (bn_1),
 *                fc_2 (bn_2), fc_3 (none)
 *   train != 0:  batch statistics (running statistics updated in place), dropout keep(i) = hash(seed, i) >= p * 2^32 scaled 1 / (1 - p)
 *                (the package's counter hash, restated by oracle/ampnet_oracle.py:keep_mask -- not torch's Philox stream); B >= 2
 *   train == 0:  running statistics, no dropout
 *   x [B, N, 9] -> log_probs [B, n_classes]; feat_T [B, 64, 64] = feature_transform.
 * The backward takes (d_log_probs [B, n_classes], d_feat_T [B, 64, 64] or NULL), the drop_p / seed of the forward and its untouched
 * workspace; grads_host [AMPNET_POINTNET_CLS_LAYERS * 4] as above, every gradient overwritten.                                      */
#define AMPNET_POINTNET_CLS_LAYERS 20
size_t ampnet_pointnet_cls_workspace_bytes(int variant, int B, int N, int n_classes);
int ampnet_pointnet_cls_fwd_f32(const float *const *layers_host, int variant, const float *x, int B, int N, int n_classes, int train,
                                float drop_p, uint32_t seed, float *log_probs, float *feat_T, void *workspace, size_t workspace_bytes,
                                void *stream);
int ampnet_pointnet_cls_bwd_f32(const float *const *layers_host, float *const *grads_host, int variant, const float *x, int B, int N,
                                int n_classes, float drop_p, uint32_t seed, const float *d_log_probs, const float *d_feat_T,
                                void *workspace, size_t workspace_bytes, void *stream);

/* ---- a7: optimiser -----------------------------------------------------------------------------------------
 * replaces torch.optim.Adam.step as the reference configures it (train_pointnet-attention.py:140-141,469-470):
 * betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad; `step` counts from 1.  One launch updates a list of
 * tensors: four HOST arrays of DEVICE pointers (parameter, gradient, exp_avg, exp_avg_sq) and a host array of sizes.
 * grad_scale multiplies the gradient on the fly (1 / world_size after a SUM all-reduce).                     */
int ampnet_adam_step_f32(float *const *params_host, const float *const *grads_host, float *const *m_host,
                         float *const *v_host, const long *numel_host, int n_tensors, float lr, float beta1,
                         float beta2, float eps, int step, float grad_scale, void *stream);

/* ---- measurement hooks (bench.py roofline leg) --------------------------------------------------------
 * ampnet_profile_enable(1) clears the table and brackets every instrumented kernel launch with two HIP events on
 * the launch stream; ampnet_profile_read() synchronises the device and sums elapsed ms, launches, algorithmic
 * flops and bytes per kernel name (names: max_rows x 64 chars).  Off by default: no events, no overhead.      */
int ampnet_profile_enable(int on);
int ampnet_profile_read(int max_rows, char *names, double *ms, long long *calls, double *flops, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* AMPNET_HIP_H */
