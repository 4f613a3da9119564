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

#ifdef __cplusplus
}
#endif
#endif /* HSR_DENSIFY_H_INCLUDED */
