// hsr_common.h — shared declarations of the gfx950 rasterizer library (internal, not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define HSR_TILE_X 16
#define HSR_TILE_Y 16
#define HSR_TILE_PIX 256
#define HSR_NUM_CHANNELS 3

// ---- opaque state, carved out of the caller's three byte buffers (all fields 256-B aligned) ----
// The reference keeps the same information in GeometryState / ImageState / BinningState
// (cuda_rasterizer/rasterizer_impl.h:29-64); the layout here is ours: ranges are per TILE (the
// reference reserves one per pixel, rasterizer_impl.cu:177), there is no cub temp storage, and the
// scan works on per-block partial sums.
struct GeomState {
    float* depths;            // [P]   view-space z
    float2* means2D;          // [P]   pixel centre
    float4* conic_opacity;    // [P]   conic.xyz, opacity
    float* cov3D;             // [P,6]
    float* rgb;               // [P,3] SH colours (only when shs given)
    uint8_t* clamped;         // [P,3]
    uint32_t* tiles_touched;  // [P]
    uint32_t* point_offsets;  // [P]   inclusive scan of tiles_touched
    int* radii;               // [P]   internal radii when the caller passes NULL
    uint32_t* block_sums;     // [ceil(P/256)+1] per-preprocess-block partial sums, then exclusive offsets
    uint32_t* counters;       // [8]   [0] = num_rendered
    float4* rec;              // [P,4] packed record the tile kernels gather: {x, y, depth, -} {conic.xyz, opacity} {r, g, b, -} {-}
};
struct ImgState {
    uint2* ranges;        // [T]
    float* final_T;       // [N]
    uint32_t* n_contrib;  // [N]
    uint32_t* median_pos; // [N] 1 + list position of the splat at which the pixel's T crossed 0.5 in the forward (0: it never did).
                          //     The reference's backward re-finds that splat from the T it reconstructs by division
                          //     (backward.cu:623-626, :854-857), which is exact only up to rounding: on a pixel whose T passes
                          //     within an ulp of 0.5 it picks a neighbour, none or two.  Recording the forward's own decision
                          //     makes the median-depth gradient land on exactly the splat whose depth the forward output.
};
struct BinState {
    uint64_t* keys_unsorted;  // [R]
    uint64_t* keys;           // [R]
    uint32_t* vals_unsorted;  // [R]
    uint32_t* vals;           // [R]
    uint32_t* hist;           // radix-sort per-block digit histograms
};

// Binning state resolved ON THE DEVICE from num_rendered (speculative forward, hsr_api.hip): the host enqueues the emit,
// per-tile sort and render kernels before it has read num_rendered back, so their array bases — which depend on it
// (hsr_carve_bin) — are derived by the kernels themselves from the device-side counter.  base == NULL: not used.
struct BinDevRef {
    char* base;               // the caller's binning buffer
    const uint32_t* R_dev;    // num_rendered, written by the scan
    size_t capacity;          // bytes usable from base
};
constexpr int HSR_SORT_TILE = 4096;   // pairs per radix-sort block (hsr_sort.hip)
__host__ __device__ inline uint32_t hsr_sort_hist_entries_inline(uint32_t R)
{
    const uint32_t nblocks = (R + HSR_SORT_TILE - 1) / HSR_SORT_TILE;
    return 256u * (nblocks > 0 ? nblocks : 1u) + 256u;
}
// same arithmetic as hsr_carve_bin (each array 256-byte aligned); false when the buffer is too small for R instances
__host__ __device__ inline bool hsr_bin_resolve(const BinDevRef& ref, uint32_t R, BinState* out)
{
    const size_t Rn = R > 0 ? R : 1;
    uintptr_t p = reinterpret_cast<uintptr_t>(ref.base);
    p = (p + 255) & ~(uintptr_t)255; out->keys_unsorted = reinterpret_cast<uint64_t*>(p); p += 8 * Rn;
    p = (p + 255) & ~(uintptr_t)255; out->keys = reinterpret_cast<uint64_t*>(p); p += 8 * Rn;
    p = (p + 255) & ~(uintptr_t)255; out->vals_unsorted = reinterpret_cast<uint32_t*>(p); p += 4 * Rn;
    p = (p + 255) & ~(uintptr_t)255; out->vals = reinterpret_cast<uint32_t*>(p); p += 4 * Rn;
    p = (p + 255) & ~(uintptr_t)255; out->hist = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)hsr_sort_hist_entries_inline(R);
    return p - reinterpret_cast<uintptr_t>(ref.base) <= ref.capacity && R <= 0x7fffffffu;
}

size_t hsr_carve_geom(char* base, int P, GeomState* out);
size_t hsr_carve_img(char* base, int W, int H, ImgState* out);
size_t hsr_carve_bin(char* base, int R, BinState* out);
uint32_t hsr_sort_hist_entries(int R);

void hsr_set_error(const char* fmt, ...);

// Ablation switches (HSR_DEBUG_FLAGS, HSR_FWD_DEBUG, HSR_NO_SPECULATION) and the measured-slower experimental kernel
// families (csrc/experiments/: moments backward, per-instance rows backward, pair-pipelined forward) exist only in the
// diagnostic build (`make ablate` -> libhsr_rast_ablate.so, -DHSR_ABLATE).  The product library reads none of them: a
// stray environment variable cannot change its results.  What the product still reads are the parity-tested kernel-family
// selectors HSR_FWD_IMPL=valu, HSR_BWD_IMPL=valu|mfma|legacy and HSR_SORT_IMPL=radix|block (every choice gives the same
// results; tests/test_gpu_golden_and_scale.py runs the parity cases under each).
#ifdef HSR_ABLATE
#include <stdlib.h>
static inline const char* hsr_ablate_env(const char* name) { return getenv(name); }
#else
static inline const char* hsr_ablate_env(const char*) { return nullptr; }
#endif

#define HSR_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            hsr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return HSR_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

// launch check: always catches launch-configuration errors; in debug mode also synchronises the
// stream so that a kernel fault is reported at its call site (reference CHECK_CUDA, auxiliary.h:166-173)
#define HSR_LAUNCH_CHECK(debug, stream)                            \
    do {                                                           \
        HSR_HIP_CHECK(hipGetLastError());                          \
        if (debug) HSR_HIP_CHECK(hipStreamSynchronize(stream));    \
    } while (0)

// ---- kernel launchers (one translation unit per stage) ----
struct PreprocessArgs {
    int P, D, M, W, H;
    const float* means3D;
    const float* scales;
    float scale_modifier;
    const float* rotations;
    const float* opacities;
    const float* shs;
    const float* cov3D_precomp;
    const float* colors_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* cam_pos;
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int* radii;
    int prefiltered;
    int tiles_x, tiles_y;
};

int hsr_launch_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present,
                            hipStream_t stream);
int hsr_launch_preprocess(const PreprocessArgs& a, GeomState& g, hipStream_t stream);
int hsr_launch_scan_block_sums(int P, GeomState& g, hipStream_t stream);
int hsr_launch_duplicate(int P, const int* radii, int tiles_x, int tiles_y, GeomState& g, BinState& b, uint2* ranges,
                         hipStream_t stream);  // also zeroes ranges[0, tiles)
int hsr_launch_sort_pairs(BinState& b, int R, int end_bit, int T, uint2* ranges, hipStream_t stream);  // also fills ranges
struct HsrBinPlan { int nblk, per_block; };   // direct tile binning: workgroups and Gaussians per workgroup
bool hsr_bin_plan(int P, int T, size_t scratch_words, HsrBinPlan* plan);   // false = not applicable (radix path)
int hsr_launch_bin_count(const HsrBinPlan& plan, int P, const int* radii, int tiles_x, int tiles_y, GeomState& g, uint32_t* scratch,
                         uint2* ranges, hipStream_t stream, uint32_t* host_counter = nullptr, uint32_t host_seq = 0);
                         // per-tile counts; produces num_rendered (counters[0]; also {value, seq} into host-mapped memory)
int hsr_launch_bin_emit(const HsrBinPlan& plan, int P, const int* radii, int tiles_x, int tiles_y, GeomState& g, const uint32_t* scratch,
                        uint2* ranges, uint64_t* comp, hipStream_t stream, const BinDevRef* ref = nullptr);
                        // tile ranges; (depth, index) composites into the tile segments
int hsr_launch_tile_sort(BinState& b, int T, int P, const uint2* ranges, hipStream_t stream, const BinDevRef* ref = nullptr,
                         int avg_per_tile_hint = 0);  // per-tile (depth, index) sort; hint: entries per tile of the previous frame
int hsr_sort_tile_passes(int end_bit);
bool hsr_sort_emit_into_sorted_buffers(int end_bit);
int hsr_launch_tile_ranges(int R, int T, const uint64_t* keys, uint2* ranges, hipStream_t stream);
int hsr_launch_tile_ranges_only(int R, const uint64_t* keys, uint2* ranges, hipStream_t stream);

struct RenderFwdArgs {
    int W, H, K, semantic;
    const uint2* ranges;
    const uint32_t* point_list;
    uint32_t* masks;         // [R], parallel to point_list: the 16-bit sub-block mask of every list entry the forward stages (hsr_tile_common.h:
                             // subblock_mask), kept for the backward in BinState::vals_unsorted — free once the per-tile sort has run
    const float2* means2D;
    const float4* conic_opacity;
    const float* depths;
    const float* colors;     // [P,3]
    const float4* rec;       // packed per-Gaussian record (GeomState::rec): ONE 64-byte line instead of four arrays, or NULL
    const float* semantics;  // [P,K] or NULL
    float* final_T;
    uint32_t* n_contrib;
    uint32_t* median_pos;
    float* out_color;
    float* out_semantic;
    float* out_depth;
    float* out_median_depth;
    float* out_opacity;
    float* out_mask;  // non-semantic variant only
    int debug_flags;  // ablation switches for tools/ablate.sh (wide kernel: 1 no MFMA, 2 no row gather, 4 no blend loop); 0 in production
    BinDevRef bin;    // base != NULL: point_list is resolved on the device (speculative forward)
};
int hsr_launch_render_forward(const RenderFwdArgs& a, hipStream_t stream);
bool hsr_launch_render_forward_mma(const RenderFwdArgs& a, hipStream_t stream);   // experiments/ (ablate build): semantic, K <= 140, channel sums on the fp32 matrix cores
bool hsr_launch_render_forward_wide(const RenderFwdArgs& a, hipStream_t stream);  // experiments/ (ablate build): semantic, 29 <= K <= 124; false otherwise

struct RenderBwdArgs {
    int W, H, K, semantic, P;
    int debug_flags;  // HSR_DEBUG_FLAGS env (timing experiments only): bit0 = drop the gradient atomics
    const float* bg;  // device [3]
    const uint2* ranges;
    const uint32_t* point_list;
    const uint32_t* masks;   // [R] sub-block masks written by the forward's staging (RenderFwdArgs::masks)
    const float2* means2D;
    const float4* conic_opacity;
    const float* depths;
    const float* colors;
    const float4* rec;    // packed per-Gaussian record (GeomState::rec), or NULL
    const float* final_T;
    const uint32_t* n_contrib;
    const uint32_t* median_pos;
    const float* dL_dpix;
    const float* dL_dpix_sem;
    const float* dL_dpix_depth;
    const float* dL_dpix_median;
    const float* dL_dpix_opacity;
    float* dL_dmean2D;    // [P,3]
    float* dL_dconic;     // [P,4]
    float* dL_dopacity;   // [P]
    float* dL_dcolor;     // [P,3]
    float* dL_dsemantics; // [P,K]
    float* dL_ddepth;     // [P]
    float* rows;          // rows mode: [R][ROW] per-instance sums (scratch)
    float* grow;          // packed mode: [P][grow_stride] one gradient row per Gaussian (scratch), else NULL
    int grow_stride;
    int grow_layout;      // 0: classic packed row; 1: compact (hsr_tile_common.h, hsr_grow_col) — hsr_render_bwd_q.hip only
    const float* semantics = nullptr;   // [P,K] features: read by the exact semantic -> alpha passes only (hsr_launch_render_backward_qsema)
    int sem_c0 = 0;                      // ... first channel of the pass
};
int hsr_launch_render_backward(const RenderBwdArgs& a, hipStream_t stream);
int hsr_launch_render_backward_mfma(const RenderBwdArgs& a, hipStream_t stream);
int hsr_launch_render_backward_mom(const RenderBwdArgs& a, hipStream_t stream);   // K <= 27, packed mode: all sums on MFMA
int hsr_launch_render_backward_sub(const RenderBwdArgs& a, hipStream_t stream);   // K <= 27, packed mode: 4x4 sub-block lists, rows merged per tile in LDS
int hsr_launch_render_backward_geo(const RenderBwdArgs& a, hipStream_t stream);   // packed mode, geometry gradients only (grow_stride 16)
int hsr_launch_render_backward_q(const RenderBwdArgs& a, hipStream_t stream);     // K <= 27, packed mode: round 4, both per-pixel factors in LDS panels, moments per chunk
int hsr_launch_render_backward_qgeo(const RenderBwdArgs& a, hipStream_t stream);  // geometry gradients only, same scheme
int hsr_launch_render_backward_qsema(const RenderBwdArgs& a, hipStream_t stream); // packed rows: + the exact semantic -> alpha term into columns 0..5 (opt-in)
int hsr_backward_row_layout(int K_semantic, bool packed, int P);   // layout hsr_launch_render_backward will expect for this K (0 unless it takes the Q-panel kernel and the compact row saves a line)
int hsr_launch_render_backward_subw(const RenderBwdArgs& a, hipStream_t stream);  // K > 27, packed mode: sub-block masks, channel passes
int hsr_launch_render_backward_wide(const RenderBwdArgs& a, hipStream_t stream);  // semantic, K > 27: matrix-core channel passes
int hsr_launch_render_backward_rows(const RenderBwdArgs& a, hipStream_t stream);  // returns the kernel's KC
int hsr_rows_row_floats(int K);
bool hsr_rows_supported(int K);
int hsr_launch_inverse_map(int R, int tiles_x, int tiles_y, const uint64_t* keys, const uint32_t* vals, const float2* means2D,
                           const int* radii, const uint32_t* offsets, uint32_t* inv, hipStream_t stream);

struct PreBwdArgs {
    int P, D, M;
    const float* means3D;
    const int* radii;
    const float* shs;
    const uint8_t* clamped;
    const float* scales;
    const float* rotations;
    float scale_modifier;
    const float* cov3Ds;
    const float* viewmatrix;
    const float* projmatrix;
    float focal_x, focal_y, tan_fovx, tan_fovy;
    const float* campos;
    const float* dL_dmean2D;
    const float* dL_dconic;
    float* dL_dmean3D;
    float* dL_dcolor;
    const float* dL_ddepth;
    float* dL_dcov3D;
    float* dL_dsh;
    float* dL_dscale;
    float* dL_drot;
    // rows mode (deterministic backward): per-instance rows are summed per Gaussian here, and the sums are
    // written to the out_* arrays (then used in place of the dL_dmean2D / dL_dconic / dL_ddepth inputs)
    int rows_kc;             // 0 = legacy (atomic) mode; else the KC the rows kernel was instantiated for
    int K;
    const float* rows;       // [R][8 + 16*NG]
    const uint32_t* inv;     // emission index -> sorted position
    const uint32_t* point_offsets;
    float *out_mean2D, *out_conic, *out_opacity, *out_color, *out_semantics, *out_depth;
    // packed mode: one atomically accumulated row per Gaussian (see hsr_grow_* in hsr_tile_common.h)
    const float* grow;
    int grow_stride;
    int grow_layout;         // layout of the packed rows (RenderBwdArgs::grow_layout)
    int geo;                 // geometry-only rows (16 floats: columns 0..6, depth complete in column 6); out_color / out_opacity / out_semantics NULL
};
int hsr_launch_preprocess_backward(const PreBwdArgs& a, hipStream_t stream);
int hsr_launch_zero_visible_rows(int P, const int* radii, float* grow, int stride, hipStream_t stream);   // packed rows of visible Gaussians := 0

#ifndef HSR_OK
#define HSR_OK 0
#define HSR_ERR_INVALID_ARGUMENT (-1)
#define HSR_ERR_BUFFER_TOO_SMALL (-2)
#define HSR_ERR_HIP (-3)
#define HSR_ERR_NO_DEVICE (-4)
#endif
