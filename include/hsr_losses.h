/*
 * hsr_losses.h — C ABI of the fused loss heads on the rendered maps (libhsr_rast.so), SURVEY.md §8(f) rank 2.
 *
 * Replaces the torch eager chains of get_loss_semantic_mlp (scripts/hierslam.py:921-1016) that turn the rasterizer's
 * outputs into scalar losses and — through autograd — into the upstream gradients the rasterizer backward consumes:
 *   masked depth / colour L1          torch.abs(gt - x)[mask].sum() | .mean()        scripts/hierslam.py:921-937
 *   l1_loss_v1                        torch.abs(x - y).mean()                        utils/slam_helpers.py:5-6
 *   calc_ssim                         11x11 Gaussian window, zero padding, mean      utils/slam_external.py:54-97
 *   multi-level cross-entropy         CrossEntropyLoss per tree level over channel   scripts/hierslam.py:963-974, :993-1003,
 *                                     ranges of the K logit planes                   transfer_tree_rendered_labelmap :91-111
 * Every entry point computes the loss value AND d loss / d input in the same call (one read of the maps): the value
 * goes to a device scalar, the gradient to a caller-owned plane set.  The 1x1-conv leaf MLP (scripts/hierslam.py:1756)
 * stays a torch Conv2d; its logits go through hsr_loss_tree_ce with one level.
 *
 * All pointers are DEVICE pointers (except `level_sizes`, host), planar CHW fp32 as the rasterizer writes them.
 * Reductions are two-stage with a fixed order: results are reproducible bit for bit.
 * Errors: return <0 and hsr_last_error() (hsr_rasterizer.h).  No allocation inside the library.
 */
#ifndef HSR_LOSSES_H_INCLUDED
#define HSR_LOSSES_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSR_LOSS_SUM 0  /* torch.abs(..)[mask].sum()   — tracking (scripts/hierslam.py:925, :935, :937) */
#define HSR_LOSS_MEAN 1 /* torch.abs(..)[mask].mean()  — mapping  (scripts/hierslam.py:927), l1_loss_v1 */
#define HSR_LOSS_MAX_LEVELS 16

/* bytes of device scratch any entry point below needs for maps of `channels` x H x W */
size_t hsr_loss_scratch_bytes(int channels, int H, int W);

/* L1 between `pred` and `gt` ([C,H,W]) over the pixels selected by `mask` (uint8 [H,W], shared by the C planes as
 * torch.tile(mask, (C,1,1)) does; NULL = all pixels).  out_loss: float[1].  out_grad ([C,H,W], may be NULL) receives
 * d loss / d pred = sign(pred - gt) * (1 or 1/count) on selected pixels, 0 elsewhere.  An empty selection gives
 * 0 (sum) or NaN (mean), like torch. */
int hsr_loss_l1(int C, int H, int W, const float* pred, const float* gt, const uint8_t* mask, int reduction, float* out_loss,
                float* out_grad, char* scratch, size_t scratch_bytes, void* stream);

/* The tracking loss of get_loss / get_loss_semantic / get_loss_semantic_mlp with the reference's shipped tracking settings (use_l1,
 * ignore_outlier_depth_loss = False; scripts/hierslam.py:903-937, configs/replica/hierslam_semantic_run.py:75-84):
 *     mask  = (gt_depth > 0) & ~isnan(depth) & (silhouette > sil_thres)          (the last factor only if use_sil)
 *     depth = sum |gt_depth - depth|[mask]        im = sum |gt_im - im|[mask tiled over the C channels]
 * Value pass: out4 (DEVICE float[4]) = { depth, im, w_depth * depth + w_im * im, 1 / selected pixels }.  Gradient pass (when autograd
 * asks): d_im ([C,H,W]) / d_depth ([H,W]) = upstream[0] * w * sign(pred - gt) on the selected pixels, 0 elsewhere (`upstream`: DEVICE
 * float, NULL = 1; either output may be NULL).  The silhouette enters the mask only (the reference detaches it there).
 * reduction HSR_LOSS_MEAN: the mapping branch's depth term, torch.abs(gt_depth - depth)[mask].mean() with mask = (gt_depth > 0) &
 * ~isnan(depth) (scripts/hierslam.py:927; use_sil = 0 there, C = 0 skips the colour term: im / gt_im / d_im NULL) — the gradient pass
 * then takes `inv_count` = &out4[3] of the value pass (NULL for sums).
 * im / gt_im: [C,H,W]; depth / gt_depth / silhouette: [H,W].  Scratch: hsr_loss_tracking_scratch_bytes(H, W). */
size_t hsr_loss_tracking_scratch_bytes(int H, int W);
int hsr_loss_tracking_value(int C, int H, int W, const float* im, const float* gt_im, const float* depth, const float* gt_depth,
                            const float* silhouette, float sil_thres, int use_sil, int reduction, float w_depth, float w_im, float* out4,
                            char* scratch, size_t scratch_bytes, void* stream);
int hsr_loss_tracking_grad(int C, int H, int W, const float* im, const float* gt_im, const float* depth, const float* gt_depth,
                           const float* silhouette, float sil_thres, int use_sil, float w_depth, float w_im, const float* upstream,
                           const float* inv_count, float* d_im, float* d_depth, void* stream);

/* calc_ssim(img1, img2, window_size = 11, size_average = True) (utils/slam_external.py:66-97).  out_ssim: float[1];
 * out_grad ([C,H,W], may be NULL) receives d ssim / d img1. */
int hsr_loss_ssim(int C, int H, int W, const float* img1, const float* img2, float* out_ssim, float* out_grad, char* scratch,
                  size_t scratch_bytes, void* stream);

/* The same two heads in two passes, for an autograd node (hsr_utils/losses.py): value now, gradient when autograd asks — multiplied by
 * the node's incoming gradient read from DEVICE memory (`upstream`, NULL = 1), so that no stashed gradient is rescaled afterwards.
 * hsr_loss_l1_grad: sums (masked or not) and the unmasked mean (the masked mean's gradient needs the selection count: hsr_loss_l1).
 * hsr_loss_ssim_value writes the three partial-derivative maps into `maps` (3 * C * H * W floats owned by the caller, alive until
 * hsr_loss_ssim_grad; NULL = value only); scratch: hsr_loss_scratch_bytes(C, H, W) is ample. */
int hsr_loss_l1_grad(int C, int H, int W, const float* pred, const float* gt, const uint8_t* mask, int reduction, const float* upstream,
                     float* out_grad, void* stream);
int hsr_loss_ssim_value(int C, int H, int W, const float* img1, const float* img2, float* out_ssim, float* maps, char* scratch,
                        size_t scratch_bytes, void* stream);
int hsr_loss_ssim_grad(int C, int H, int W, const float* img1, const float* img2, const float* maps, const float* upstream, float* out_grad,
                       void* stream);

/* For level l = 0..num_levels-1 with channel range [sum(level_sizes[:l]), +level_sizes[l]) of `logits` ([K,H,W]):
 *   out_level_loss[l] = CrossEntropyLoss()(logits[range] viewed as [H*W, n_l], labels[l])      (mean over pixels whose
 * label != ignore_index; torch's default ignore_index is -100).  `labels` is int64 [>= num_levels, H, W] (the reference
 * calls .long()).  out_grad ([K,H,W], may be NULL) receives d (sum_l level_weight[l] * loss_l) / d logits; channels
 * outside every level get 0.  `level_sizes` and `level_weight` are HOST arrays of num_levels entries; level_weight == NULL
 * means all ones.  Labels must lie in [0, level_sizes[l]) or equal ignore_index (torch asserts this; here an out-of-range
 * label is treated as matching no class). */
int hsr_loss_tree_ce(int K, int H, int W, int num_levels, const int* level_sizes, const float* level_weight, const float* logits,
                     const int64_t* labels, int ignore_index, float* out_level_loss, float* out_grad, char* scratch,
                     size_t scratch_bytes, void* stream);

/* The same loss in two passes, for an autograd node (hsr_utils/losses.py): the value pass counts the valid labels itself and writes no
 * gradient — out_level_loss[l] as above, out_inv_count[l] = 1 / (pixels of level l whose label != ignore_index), both DEVICE float
 * [num_levels]; the gradient pass writes out_grad = upstream[0] * d (sum_l level_weight[l] * loss_l) / d logits, where `upstream` is a
 * DEVICE float (the node's incoming gradient; NULL = 1) and `inv_count` is what the value pass returned.  `add_grad` ([K,H,W] or NULL):
 * another head's gradient with respect to the same map (the leaf head's stashed d loss / d sem) joins in the same pass,
 * out_grad += add_grad * add_scale[0] * add_host_scale (`add_scale`: DEVICE float or NULL = 1) — instead of a `stash * g` pass of its own and
 * autograd's add of two K x H x W maps.  Scratch (value pass):
 * hsr_loss_tree_ce_scratch_bytes(H, W) — block partials only, a few hundred KB. */
size_t hsr_loss_tree_ce_scratch_bytes(int H, int W);
int hsr_loss_tree_ce_value(int K, int H, int W, int num_levels, const int* level_sizes, const float* logits, const int64_t* labels,
                           int ignore_index, float* out_level_loss, float* out_inv_count, char* scratch, size_t scratch_bytes,
                           void* stream);
int hsr_loss_tree_ce_grad(int K, int H, int W, int num_levels, const int* level_sizes, const float* level_weight, const float* logits,
                          const int64_t* labels, int ignore_index, const float* inv_count, const float* upstream, const float* add_grad,
                          const float* add_scale, float add_host_scale, float* out_grad, void* stream);

/* Leaf head, fused: logits = Conv2d(K, C, kernel_size=1)(sem) (weight [C,K] = the conv's [C,K,1,1], bias [C];
 * scripts/hierslam.py:1756), loss = CrossEntropyLoss()(logits as [H*W, C], labels) (scripts/hierslam.py:976-983), and the
 * gradients d loss / d sem ([K,H,W]), d weight ([C,K]), d bias ([C]) — each may be NULL.  The [C,H,W] logits are never
 * materialised.  Supports K <= 31 and C <= 128 (returns HSR_ERR_INVALID_ARGUMENT beyond: use the conv + hsr_loss_tree_ce
 * composition).  labels: int64 [H,W].  Scratch: hsr_loss_scratch_bytes(K, H, W). */
int hsr_loss_leaf_mlp_ce(int K, int C, int H, int W, const float* sem, const float* weight, const float* bias, const int64_t* labels,
                         int ignore_index, float* out_loss, float* d_sem, float* d_weight, float* d_bias, char* scratch,
                         size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HSR_LOSSES_H_INCLUDED */
