/*
 * hsr_rasterizer.h — C ABI of the MI355X-native differentiable Gaussian rasterizer (libhsr_rast.so).
 *
 * This is the drop-in boundary for the ONE hot path of LeeBY68/Hier-SLAM: it replaces the static
 * C++ interface `CudaRasterizer::Rasterizer` of the reference
 * (hierslam-diff-gaussian-rasterization-w-depth/cuda_rasterizer/rasterizer.h:24-162), which the
 * reference's torch glue (rasterize_points.cu:36-432) calls.  Each entry point below names the
 * reference member it replaces.  Differences from the reference interface, all forced by making it
 * a plain C ABI and by the MI355X design:
 *   - `std::function<char*(size_t)>` allocator callbacks (rasterizer.h:37-39, rasterize_points.cu:27-33)
 *     become `hsr_buffer` descriptors: a caller-owned device allocation plus an optional C `grow`
 *     callback.  Steady state needs no callback at all (the caller sizes buffers with the
 *     hsr_required_*_bytes() queries), which keeps the host off the critical path.
 *   - NUM_SEMANTIC is a compile-time macro in the reference (config.h:18, edit + reinstall per
 *     dataset); here K is a run-time argument.
 *   - every launch goes to the `stream` argument (a hipStream_t); the reference launches on the legacy
 *     default stream (e.g. forward.cu:652).
 *   - errors are return codes (<0) + hsr_last_error(); the reference throws std::runtime_error
 *     (rasterizer_impl.cu:509-512, auxiliary.h:166-173).
 *   - gradient outputs are fully written by the library (no caller zero-fill needed; the reference
 *     requires torch::zeros, rasterize_points.cu:378-388).
 * All pointers except `hsr_buffer*` and the host callback are DEVICE pointers to contiguous
 * fp32 / int32 data laid out exactly as the reference's tensors (AoS [P,3], [P,4], [P,K]; planar
 * CHW images).  Absent optional inputs are NULL (the reference's `data_ptr()==nullptr` switches,
 * forward.cu:205, :241).  No torch types appear in this interface.
 */
#ifndef HSR_RASTERIZER_H_INCLUDED
#define HSR_RASTERIZER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSR_OK 0
#define HSR_ERR_INVALID_ARGUMENT (-1) /* bad shape / NULL where data is required / unsupported combination */
#define HSR_ERR_BUFFER_TOO_SMALL (-2) /* an hsr_buffer is too small and has no grow callback */
#define HSR_ERR_HIP (-3)              /* a HIP runtime call failed; text in hsr_last_error() */
#define HSR_ERR_NO_DEVICE (-4)        /* no gfx950 device / kernels not loadable */

/* grows a device allocation to at least `bytes` and returns its base (NULL on failure).
 * Replaces std::function<char*(size_t)> (rasterizer.h:37-39); `user` is opaque. */
typedef char* (*hsr_grow_fn)(size_t bytes, void* user);

/* A caller-owned device scratch/state buffer (the reference's geomBuffer / binningBuffer / imgBuffer,
 * rasterize_points.cu:287-292).  The library uses `ptr` if `capacity` suffices, else calls `grow`
 * (if non-NULL) and updates ptr/capacity in place.  Contents are opaque to the caller and must be
 * handed back unchanged to the matching backward call. */
typedef struct hsr_buffer {
    char* ptr;
    size_t capacity;
    hsr_grow_fn grow;
    void* user;
} hsr_buffer;

/* Sizes of the three state buffers (the reference's required<GeometryState>(P) etc.,
 * rasterizer_impl.h:66-73).  binning is a function of num_rendered. */
size_t hsr_required_geometry_bytes(int P);
size_t hsr_required_image_bytes(int width, int height);
size_t hsr_required_binning_bytes(int num_rendered);
/* Size of the backward's scratch buffer for the accumulation mode in force: the packed per-Gaussian gradient rows of mode 0
 * (P rows of a 64-byte-aligned stride that holds the 10 + K sums of one Gaussian: 48 floats at K = 26), which the tile kernel
 * adds into and the per-Gaussian kernel unpacks into the reference's arrays; 0 in mode 2, which needs none.  The caller owns it
 * (the reference allocates its backward scratch itself with cudaMalloc/cudaFree per call, rasterizer_impl.cu:673-701).  A call
 * without scratch, or with too little, falls back to mode 2 for that call. */
size_t hsr_backward_scratch_bytes(int P, int K, int num_rendered);
/* How the backward accumulates per-Gaussian sums (process-wide; default 0, or HSR_BWD_IMPL=legacy):
 *   0 packed : fp32 atomics into one 64-byte-aligned scratch row per Gaussian, unpacked by the per-Gaussian
 *              kernel — about half the atomic requests of the reference's six separate arrays (default);
 *   1        : refused by this library (HSR_ERR_INVALID_ARGUMENT): the per-instance-rows experiment lives in the
 *              diagnostic build `make ablate` only;
 *   2 legacy : atomics straight into the six output arrays, as the reference does; needs no scratch. */
int hsr_set_backward_mode(int mode);
int hsr_get_backward_mode(void);   /* 0 packed, 2 legacy */

/* Gradient of the semantic loss with respect to alpha (process-wide; default 0, or HSR_SEMANTIC_ALPHA=exact):
 *   0 reference : none.  The reference stages the features for this term into a shared array nothing writes
 *                 (RAST/cuda_rasterizer/backward.cu:778-779 commented out, :834-845 read it), so its semantic loss
 *                 moves the features only, never opacity / covariance / position.  The drop-in default.
 *   1 exact     : the term those lines intend — (feature - accum_rec) . dL_dsemantic joins dL_dalpha like the colour
 *                 channels' — as ceil(K / 16) extra passes of the tile kernel over the packed rows.  Needs the packed
 *                 accumulation mode (a scratch buffer) and semantics_precomp; any K.  Not what the reference trains with. */
int hsr_set_semantic_alpha_mode(int mode);
int hsr_get_semantic_alpha_mode(void);

/* Thread-local text of the last error returned by any hsr_* call on this thread. */
const char* hsr_last_error(void);
/* Library / build identification, e.g. "hsr_rast 0.1 gfx950". */
const char* hsr_version(void);

/* Replaces Rasterizer::markVisible (rasterizer.h:27-32, rasterizer_impl.cu:141-153).
 * present: device uint8[P] (1 = view-space z > 0.2). */
int hsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* Replaces Rasterizer::forward (rasterizer.h:34-64, rasterizer_impl.cu:198-345).
 * Outputs: out_color[3,H,W], out_depth/out_median_depth/out_opacity/out_mask[1,H,W], radii int32[P]
 * (radii may be NULL).  Returns num_rendered (>= 0) or a negative HSR_ERR_*.
 * D = active SH degree, M = SH coefficients per Gaussian (0 when colors_precomp is given). */
int hsr_forward(hsr_buffer* geometry, hsr_buffer* binning, hsr_buffer* image,
                int P, int D, int M, const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered,
                float* out_color, float* out_depth, float* out_median_depth, float* out_opacity, float* out_mask,
                int* radii, int debug, void* stream);

/* Replaces Rasterizer::forward_semantic (rasterizer.h:98-128, rasterizer_impl.cu:460-610).
 * K = number of semantic channels (semantics_precomp is [P,K], out_semantic is [K,H,W]). */
int hsr_forward_semantic(hsr_buffer* geometry, hsr_buffer* binning, hsr_buffer* image,
                         int P, int D, int M, int K, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* colors_precomp,
                         const float* semantics_precomp, const float* opacities,
                         const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                         float tan_fovx, float tan_fovy, int prefiltered,
                         float* out_color, float* out_semantic, float* out_depth, float* out_median_depth,
                         float* out_opacity, int* radii, int debug, void* stream);

/* ---- non-blocking forward (no counterpart in the reference, whose forward stalls the stream and the host on a 4-byte
 * cudaMemcpy of num_rendered in every frame, rasterizer_impl.cu:285 / :548) ----
 * hsr_forward* normally returns num_rendered, i.e. waits until the device has counted the instances.  A caller that has sized the
 * binning buffer generously (num_rendered changes slowly from frame to frame) can take the host off that wait:
 *     hsr_ticket tk;  hsr_forward_arm_async(&tk);
 *     rc = hsr_forward_semantic(...);           // enqueues EVERYTHING (count, emit, sort, render) and returns HSR_PENDING at once
 *     ... enqueue more work, e.g. the loss ...
 *     R = hsr_forward_end(&tk, 1, stream);      // num_rendered — long since written when the backward needs it
 * If the armed call cannot run ahead (no pre-sized binning buffer, more than 8192 tiles, debug) it runs as usual and returns
 * num_rendered; the ticket then has seq == 0.  If num_rendered turns out NOT to fit the binning buffer the device kernels that
 * depend on it do nothing except fill the output images with NaN, hsr_forward_end returns HSR_ERR_BUFFER_TOO_SMALL (ticket->rendered
 * holds the count) and the caller must run the forward again with a larger buffer: everything computed from those outputs in
 * between is invalid — which is why this is opt-in and the Python layer raises (diff_gaussian_rasterization.set_async_forward).
 * Counts live in a ring of 256 host-mapped slots per device: a ticket that is still unresolved when 256 later non-blocking forwards
 * have started on its device has lost its count, and hsr_forward_end says so at once (HSR_ERR_INVALID_ARGUMENT, "overwritten"). */
#define HSR_PENDING (-100)
typedef struct hsr_ticket {
    uint32_t seq;                 /* 0: the armed call did not run ahead (it returned num_rendered itself) */
    int32_t device;
    volatile uint32_t* slot;      /* host-mapped { num_rendered, seq, prefilter flag } of this call */
    char* binning_base;           /* the binning buffer the device kernels resolved their arrays in */
    size_t binning_capacity;
    int32_t prefiltered;
    int32_t rendered;             /* out (hsr_forward_end): num_rendered, also when it did not fit */
} hsr_ticket;
/* arms the calling thread's NEXT hsr_forward / hsr_forward_semantic call */
int hsr_forward_arm_async(hsr_ticket* ticket);
/* block != 0: waits for the count (10 s watchdog, then a synchronisation of `stream`); block == 0: HSR_PENDING if it has not arrived.
 * Returns num_rendered (>= 0), HSR_ERR_BUFFER_TOO_SMALL (outputs are NaN-filled, see above), or another negative HSR_ERR_*. */
int hsr_forward_end(hsr_ticket* ticket, int block, void* stream);

/* Replaces Rasterizer::backward (rasterizer.h:66-96, rasterizer_impl.cu:349-454).
 * R = num_rendered returned by the matching hsr_forward; the three buffers are the ones it filled.
 * dL_dmean2D is [P,3] (z unused), dL_dconic [P,4] (.z unused), dL_dopacity [P], dL_dcolor [P,3],
 * dL_ddepth [P], dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3], dL_dscale [P,3], dL_drot [P,4].
 * dL_dcov3D, dL_dscale, dL_drot may be NULL (not wanted); dL_dconic and dL_ddepth — intermediates the reference keeps to
 * itself (rasterize_points.cu:380-383) — may be NULL when `scratch` carries the accumulation (the default).
 * Geometry-only call: dL_dcolor, dL_dopacity and dL_dsemantics ALL NULL (with colors_precomp and the default accumulation
 * mode) = the caller optimises the camera pose only (a tracking iteration, scripts/hierslam.py:1683-1860): the tile kernel
 * then forms just the mean2D / conic / depth sums — one 64-byte row per Gaussian instead of three, no semantic upstream
 * gradients read; dL_dmean2D and dL_dmean3D are the same as in a full call.
 * All are fully overwritten.
 * scratch: device buffer of hsr_backward_scratch_bytes(P, K, R) bytes, or NULL.  With it (default accumulation mode,
 * hsr_set_backward_mode) the per-splat sums go by fp32 atomics into ONE packed, 64-byte-aligned row per Gaussian
 * inside the scratch, which the per-Gaussian kernel unpacks; without it they go straight into the six output arrays
 * like the reference (more atomic requests).  Either way last bits are order-dependent. */
int hsr_backward(int P, int D, int M, int R, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* campos,
                 float tan_fovx, float tan_fovy, const int* radii,
                 const char* geom_buffer, const char* binning_buffer, const char* img_buffer,
                 const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_median_depth,
                 const float* dL_dpix_final_opacity,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 char* scratch, size_t scratch_bytes, int debug, void* stream);

/* Replaces Rasterizer::backward_semantic (rasterizer.h:130-162, rasterizer_impl.cu:614-731).
 * dL_dpix_semantic is [K,H,W], dL_dsemantics [P,K].  The semantic loss reaches only dL_dsemantics:
 * the reference's semantic->alpha term reads a scratch buffer it never writes (backward.cu:834,
 * rasterizer_impl.cu:673-674), i.e. contributes 0; this library reproduces that observed behaviour. */
int hsr_backward_semantic(int P, int D, int M, int K, int R, const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp,
                          const float* semantics_precomp,
                          const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, const float* campos,
                          float tan_fovx, float tan_fovy, const int* radii,
                          const char* geom_buffer, const char* binning_buffer, const char* img_buffer,
                          const float* dL_dpix, const float* dL_dpix_semantic, const float* dL_dpix_depth,
                          const float* dL_dpix_median_depth, const float* dL_dpix_final_opacity,
                          float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                          float* dL_dsemantics, float* dL_ddepth,
                          float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                          char* scratch, size_t scratch_bytes, int debug, void* stream);

/* Read-only view into the opaque state buffers, for parity tests and debugging only (the reference
 * keeps the same fields in GeometryState / BinningState / ImageState, rasterizer_impl.h:29-64).
 * Offsets are in bytes from the start of the respective buffer. */
typedef struct hsr_state_layout {
    size_t geom_depths, geom_means2D, geom_conic_opacity, geom_cov3D, geom_rgb, geom_clamped,
        geom_tiles_touched, geom_point_offsets, geom_radii;
    size_t bin_keys_unsorted, bin_keys, bin_vals_unsorted, bin_vals;
    size_t img_ranges, img_final_T, img_n_contrib;
    size_t img_median_pos;   /* u32[N]: 1 + list position of the splat at which the pixel's T crossed 0.5 (0: never) — no
                              * counterpart in the reference, whose backward re-derives it from a reconstructed T */
} hsr_state_layout;
int hsr_get_state_layout(int P, int width, int height, int num_rendered, hsr_state_layout* out);

/* ---- measurement hooks (no counterpart in the reference, which has no profiling: SURVEY.md §5) ----
 * With profiling enabled every stage launch is bracketed by hipEventRecord on the caller's stream;
 * hsr_profile_read() waits for the recorded events and returns accumulated device time per stage.
 * bench.py uses this to obtain the dominant kernel's average duration "live" for the roofline line. */
enum {
    HSR_STAGE_FWD_PREPROCESS = 0, HSR_STAGE_FWD_SCAN, HSR_STAGE_FWD_DUPLICATE, HSR_STAGE_FWD_SORT,
    HSR_STAGE_FWD_RANGES, HSR_STAGE_FWD_RENDER, HSR_STAGE_BWD_ZERO, HSR_STAGE_BWD_RENDER, HSR_STAGE_BWD_PREPROCESS,
    HSR_STAGE_COUNT
};
typedef struct hsr_profile {
    double ms[HSR_STAGE_COUNT];      /* summed device milliseconds per stage since the last reset */
    uint64_t calls[HSR_STAGE_COUNT]; /* number of timed launches (a stage may be several kernels) */
} hsr_profile;
int hsr_profile_enable(int on);
/* restrict the stage timers to the stages whose bit is set (bit i = stage i; default: all).  Every timed stage costs two
 * event records per call, which perturb a tight launch sequence; a measurement of one kernel should time only that one. */
int hsr_profile_select(unsigned stage_mask);
int hsr_profile_read(hsr_profile* out, int reset);
const char* hsr_stage_name(int stage);
/* milliseconds the calling thread's hsr_forward* calls have spent blocked on the num_rendered read-back since the last reset.
 * Near zero while the device keeps working = the HOST is the bottleneck of the call sequence (the device idles between
 * calls); about one device step per call = device-bound. */
double hsr_profile_host_wait_ms(int reset);

#ifdef __cplusplus
}
#endif
#endif /* HSR_RASTERIZER_H_INCLUDED */
