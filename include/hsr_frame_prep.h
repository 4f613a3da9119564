/*
 * hsr_frame_prep.h — C ABI of the fused rasterizer-input preparation (libhsr_rast.so), SURVEY.md §8(f) rank 1.
 *
 * Replaces, for one frame, the chain of small torch kernels the reference runs immediately before every
 * rasterizer call:
 *   transform_to_frame                        utils/slam_helpers.py:278-330   (+ build_rotation, utils/slam_external.py:25-42,
 *                                                                               quat_mult, utils/slam_helpers.py:21-28)
 *   transformed_params2rendervar              utils/slam_helpers.py:124-139   (and the silhouette twin, :176-193)
 *   transformed_params2rendervar_semantic     utils/slam_helpers.py:195-219
 *   transformed_params2depthplussilhouette    utils/slam_helpers.py:260-275   (+ get_depth_and_silhouette, :222-239)
 * and their autograd backward, including the reduction that forms the camera-pose gradients.  One launch
 * forward, two launches backward (per-Gaussian pass + a one-block finish of the pose reduction).
 *
 * Semantics restated (all fp32, all pointers DEVICE pointers, contiguous):
 *   q  = F.normalize(cam_unnorm_rots[0, :, time_idx])      x / max(|x|, 1e-12)
 *   R  = build_rotation(q)                                  normalises q once more, then the usual 3x3
 *   out_means3D[p]        = R * means3D[p] + cam_trans[0, :, time_idx]
 *   out_unnorm_rot[p]     = transform_rots ? quat_mult(q, F.normalize(unnorm_rotations[p])) : unnorm_rotations[p]
 *   out_rotations[p]      = F.normalize(rot_source == HSR_PREP_ROT_PARAMS ? unnorm_rotations[p] : out_unnorm_rot[p])
 *   out_opacities[p]      = sigmoid(logit_opacities[p])
 *   out_scales[p, 0..2]   = exp(log_scales[p, S == 1 ? 0 : 0..2])
 *   out_depth_sil[p]      = { z, 1, z*z },  z = w2c[2, 0..2] . out_means3D[p] + w2c[2, 3]      (only if requested)
 * `cam_unnorm_rots` is [1, 4, num_frames] and `cam_trans` [1, 3, num_frames] exactly as the reference stores
 * them (scripts/hierslam.py:394-395): element (c, t) sits at c*num_frames + t.
 *
 * Errors: return <0 and hsr_last_error() (declared in hsr_rasterizer.h).  P == 0 is legal (pose gradients
 * come out zero).  No torch types, no allocation inside the library.
 */
#ifndef HSR_FRAME_PREP_H_INCLUDED
#define HSR_FRAME_PREP_H_INCLUDED

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSR_PREP_ROT_PARAMS 0      /* rotations from params['unnorm_rotations']   (slam_helpers.py:212, semantic variant) */
#define HSR_PREP_ROT_TRANSFORMED 1 /* rotations from the transformed quaternions  (slam_helpers.py:134, :188, :270) */

/* bytes of device scratch hsr_frame_prep_backward needs (per-block partial sums of the pose reduction) */
size_t hsr_frame_prep_scratch_bytes(int P);

/* transform_to_frame + transformed_params2rendervar[_semantic | depthplussilhouette], forward.
 * S = log_scales.shape[1] (1 isotropic, 3 anisotropic); transform_rots as slam_helpers.py:302-306 (the
 * reference derives it from S; it is explicit here).  `w2c` (16 floats, row-major, DEVICE) and
 * `out_depth_sil` are both NULL unless the depth+silhouette colours are wanted; `out_unnorm_rot` may be
 * NULL when the caller does not need transformed_gaussians['unnorm_rotations'] itself. */
int hsr_frame_prep_forward(int P, int S, int transform_rots, int rot_source, const float* means3D, const float* unnorm_rotations,
                           const float* logit_opacities, const float* log_scales, const float* cam_unnorm_rots,
                           const float* cam_trans, int num_frames, int time_idx, const float* w2c, float* out_means3D,
                           float* out_unnorm_rot, float* out_rotations, float* out_opacities, float* out_scales,
                           float* out_depth_sil, void* stream);

/* Backward of the above (what torch.autograd derives for the reference's op chain).  Upstream gradients
 * that are NULL count as zero.  Every output gradient is fully written; outputs that are NULL are skipped.
 * dL_dcam_unnorm_rot[4] / dL_dcam_tran[3] are the gradients of the `time_idx` column only (the other
 * columns of the reference's parameter receive zero).  The pose reduction is two-stage and deterministic
 * (fixed block partition, fixed summation order): same inputs give the same bits. */
int hsr_frame_prep_backward(int P, int S, int transform_rots, int rot_source, const float* means3D, const float* unnorm_rotations,
                            const float* logit_opacities, const float* log_scales, const float* cam_unnorm_rots,
                            const float* cam_trans, int num_frames, int time_idx, const float* w2c,
                            const float* dL_dout_means3D, const float* dL_dout_unnorm_rot, const float* dL_dout_rotations,
                            const float* dL_dout_opacities, const float* dL_dout_scales, const float* dL_dout_depth_sil,
                            float* dL_dmeans3D, float* dL_dunnorm_rotations, float* dL_dlogit_opacities, float* dL_dlog_scales,
                            float* dL_dcam_unnorm_rot, float* dL_dcam_tran, char* scratch, size_t scratch_bytes, void* stream);

/* The same, with the pose gradients shaped like the parameters the reference optimises (scripts/hierslam.py:394-395):
 * dL_dcam_unnorm_rots [4 * num_frames] and dL_dcam_trans [3 * num_frames], element (c, t) at c * num_frames + t — column
 * `time_idx` receives the gradient, every other column is zeroed by the same launch (what autograd's slicing backward
 * produces with a zero-fill and a strided copy per parameter). */
int hsr_frame_prep_backward_params(int P, int S, int transform_rots, int rot_source, const float* means3D, const float* unnorm_rotations,
                                   const float* logit_opacities, const float* log_scales, const float* cam_unnorm_rots,
                                   const float* cam_trans, int num_frames, int time_idx, const float* w2c,
                                   const float* dL_dout_means3D, const float* dL_dout_unnorm_rot, const float* dL_dout_rotations,
                                   const float* dL_dout_opacities, const float* dL_dout_scales, const float* dL_dout_depth_sil,
                                   float* dL_dmeans3D, float* dL_dunnorm_rotations, float* dL_dlogit_opacities, float* dL_dlog_scales,
                                   float* dL_dcam_unnorm_rots, float* dL_dcam_trans, char* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HSR_FRAME_PREP_H_INCLUDED */
