/*
 * hsr_densify.h — C ABI of the silhouette densification step on device (libhsr_rast.so), SURVEY.md §8(f) rank 3.
 *
 * Replaces, for one mapped frame, the numeric part of add_new_gaussians_semantic (scripts/hierslam.py:1264-1305; same in
 * add_new_gaussians :1169-1215): the non-presence mask
 *     depth_error = |gt_depth - render_depth| * (gt_depth > 0)
 *     mask = (silhouette < sil_thres) | ((render_depth > gt_depth) & (depth_error > 50 * depth_error.median()))      :1271-1278
 *     mask &= gt_depth > 0                                                                                            :1289-1290
 * the back-projection of the selected pixels (get_pointcloud, scripts/hierslam.py:144-194: pixel grid -> camera points at the
 * ground-truth depth -> world through c2w; colours from the frame; mean3_sq_dist = (z / ((fx+fy)/2))^2, "projective"), and
 * log_scales = log(sqrt(mean3_sq_dist)) (initialize_new_params_semantic, :1157) — as ONE order-preserving stream compaction
 * (row-major pixel order, exactly what `point_cld[mask]` yields) instead of a global sort for the median, full-frame point
 * clouds, boolean-mask gathers and their temporaries.
 * torch.median's definition is kept: the LOWER median, element (N-1)/2 of the sorted N values (zeros of invalid pixels
 * included), found by a 4-pass radix select on the float bit patterns (non-negative finite values order like unsigned ints).
 * The Parameter / optimizer-state concatenation (:1297-1304, utils/slam_external.py:121-137) stays in torch: it is
 * bookkeeping on torch objects.
 *
 * All pointers are DEVICE pointers.  Errors: <0 and hsr_last_error().  No allocation inside the library.
 */
#ifndef HSR_DENSIFY_H_INCLUDED
#define HSR_DENSIFY_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

size_t hsr_densify_scratch_bytes(int H, int W);

/* silhouette, render_depth, gt_depth: [H,W]; color: [3,H,W]; c2w: 16 floats row-major (inverse of the frame's w2c).
 * Outputs: out_count (int[1], the number of selected pixels M — may exceed `capacity`, in which case only the first
 * `capacity` points are written), out_means3D [capacity,3], out_rgb [capacity,3], out_log_scales [capacity],
 * out_mean_sq_dist [capacity] (may be NULL), out_mask uint8 [H,W] (may be NULL), out_median float[1] (may be NULL). */
int hsr_densify_frame(int H, int W, const float* silhouette, const float* render_depth, const float* gt_depth, const float* color,
                      float fx, float fy, float cx, float cy, const float* c2w, float sil_thres, float depth_factor, int capacity,
                      int* out_count, float* out_means3D, float* out_rgb, float* out_log_scales, float* out_mean_sq_dist,
                      uint8_t* out_mask, float* out_median, char* scratch, size_t scratch_bytes, void* stream);

/* ---- prune + concat of the map: utils/slam_external.py:121-188 --------------------------------------------------------------
 * prune_gaussians (:167-188) removes the Gaussians with sigmoid(logit_opacity) < removal threshold (and, past
 * `remove_big_after`, those with max_c exp(log_scale_c) > 0.1 * scene_radius) from EVERY per-Gaussian tensor: the six
 * parameters, their two Adam moments each, and the bookkeeping vectors (remove_points, :139-165) — ~22 boolean-mask
 * gathers, each with its own nonzero() and host sync.  cat_params_to_optimizer (:121-137) appends the densified Gaussians to
 * the same tensors, zeros for the Adam moments.  Here both are ONE order-preserving row compaction over a table of tensors:
 *   hsr_prune_mask           the keep mask of :175-180 + its scan (the compaction's offsets) + the kept count;
 *   hsr_compact_append_rows  for every table: dst[rank(i)] = src[i] for kept rows i (source order, as tensor[to_keep]), then
 *                            the n_append new rows (table.append, or zeros when NULL) behind them.
 * The caller allocates dst with room for P + n_append rows and narrows it to *out_rows afterwards (one read-back). */
#define HSR_MAX_ROW_TABLES 40
typedef struct hsr_row_table {
    const float* src;      /* [P, cols] */
    const float* append;   /* [n_append, cols] or NULL = zeros */
    float* dst;            /* [>= kept + n_append, cols] */
    int cols;
} hsr_row_table;

size_t hsr_compact_scratch_bytes(int P);

/* out_keep: uint8[P] (1 = keep); out_kept: int[1].  log_scales is [P,S] (S = 1 isotropic, 3 anisotropic);
 * big_world_threshold <= 0 switches the size test off (iter < remove_big_after).  `scratch` afterwards holds the scanned block
 * counts hsr_compact_append_rows needs: pass the same scratch and keep_is_scanned = 1. */
int hsr_prune_mask(int P, int S, const float* logit_opacities, const float* log_scales, float removal_opacity_threshold,
                   float big_world_threshold, uint8_t* out_keep, int* out_kept, char* scratch, size_t scratch_bytes, void* stream);

/* keep == NULL: every row is kept (pure concat).  keep_is_scanned: `scratch` and *out_rows come from hsr_prune_mask on the same
 * mask.  out_rows (int[1], device): kept + n_append.  At most HSR_MAX_ROW_TABLES tables; `tables` is a HOST array. */
int hsr_compact_append_rows(int P, const uint8_t* keep, int keep_is_scanned, int n_tables, const hsr_row_table* tables, int n_append,
                            int* out_rows, char* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HSR_DENSIFY_H_INCLUDED */
