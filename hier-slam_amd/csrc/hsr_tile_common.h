// hsr_tile_common.h — pieces shared by the forward and backward 16x16-tile kernels (gfx950, wave64).
//
// Tile decomposition: one 256-thread workgroup per 16x16 tile (the reference's BLOCK_X x BLOCK_Y,
// config.h:16-17, which also fixes the binning keys), four waves, each wave owning one 8x8 QUADRANT
// of the tile (lane l -> pixel (l & 7, l >> 3) of the quadrant).  A compact wave footprint matters
// because every per-splat decision ("does any of my 64 pixels see this splat?") is taken per wave.
//
// Staged splat record (LDS, one per list entry of the current batch), pre-scaled so that the blend loop
// evaluates alpha with as few VALU instructions as possible — these kernels are VALU-issue bound:
//     geo  = { x, y, A, B }      A = -0.5*log2(e)*conic.x     B = -log2(e)*conic.y
//     co   = { C, opacity }      C = -0.5*log2(e)*conic.z
//   => log2(G) = A*dx*dx + B*dx*dy + C*dy*dy,  G = exp2(.) (one v_exp_f32),  alpha = min(0.99, opacity*G)
// which is the reference's power = -0.5*(cx*dx^2 + cz*dy^2) - cy*dx*dy, alpha = min(0.99, o*exp(power))
// (forward.cu:484-492) up to fp32 rounding (tolerance-tested; thresholds are applied to the same alpha).
//
// Per-wave culling: at staging time the lane that owns a splat also computes a conservative bounding
// box of the region where alpha can reach 1/255 (an ellipse: A*dx^2 + B*dx*dy + C*dy^2 >= -log2(255*o))
// and, for each quadrant, whether the box touches it.  A wave ballot turns that into a compacted,
// order-preserving list of batch slots per quadrant, so the blend loop of a wave only ever visits splats
// that can touch its 64 pixels.  The exact per-pixel tests still run on the survivors, so results do not
// depend on the box (it only has to be conservative).
#pragma once
#include "hsr_common.h"

#define HSR_LOG2E 1.4426950408889634f

// XCD-aware workgroup -> tile mapping.  The dispatcher deals workgroups round-robin to the 8 XCDs (workgroup b runs on
// XCD b % 8) and each XCD has its own L2, so with tile = b every one of the ~1.8 tiles a splat touches pulls the splat's
// record (and its 4K-byte semantic row) into a different L2.  Here XCD x walks the contiguous tile range
// [x * ceil(T/8), (x+1) * ceil(T/8)) in order: horizontally and vertically adjacent tiles run on the same XCD at about the
// same time and share those lines.  Launch hsr_tile_grid(T) workgroups; workgroups mapped past T exit at once.
__host__ __device__ inline int hsr_tile_grid(int T) { return 8 * ((T + 7) / 8); }
__device__ __forceinline__ int hsr_block_tile(int b, int T) { return (b & 7) * ((T + 7) >> 3) + (b >> 3); }
// Experiment ("the untried 2-D XCD -> region tile mapping"; -DHSR_TILE_MAP_2D, tools/r04_map2d_ab.sh): each XCD gets a
// RECTANGLE of the tile grid (4 x 2 regions) and walks it row by row, so that the ~128 workgroups an XCD has in flight cover a compact
// patch (about 7 rows of 19 tiles at 1200 x 680) instead of 1.7 rows of the whole image width: splats that span two tile rows find their
// record and feature row in this XCD's L2 more often.  Workgroups mapped outside the grid exit at once.
__host__ __device__ inline int hsr_tile_grid_2d(int tx, int ty) { return 8 * (((tx + 3) / 4) * ((ty + 1) / 2)); }
__device__ __forceinline__ int hsr_block_tile_2d(int b, int tx, int ty)
{
    const int rw = (tx + 3) / 4, rh = (ty + 1) / 2;     // region size in tiles
    const int x = b & 7, i = b >> 3;
    const int cx = (x & 3) * rw + i % rw, cy = (x >> 2) * rh + i / rw;
    return (cx < tx && cy < ty) ? cy * tx + cx : tx * ty;   // tx * ty: "past the grid"
}
#ifdef HSR_TILE_MAP_2D
#define HSR_TILE_OF_BLOCK(b, tx, ty) hsr_block_tile_2d((b), (tx), (ty))
#define HSR_GRID_OF_TILES(tx, ty) hsr_tile_grid_2d((tx), (ty))
#else
#define HSR_TILE_OF_BLOCK(b, tx, ty) hsr_block_tile((b), (tx) * (ty))
#define HSR_GRID_OF_TILES(tx, ty) hsr_tile_grid((tx) * (ty))
#endif

// The speculative forward found its binning buffer too small for num_rendered (hsr_bin_resolve failed): nothing can be rendered.  A
// blocking hsr_forward* call runs the kernels again with a grown buffer; a NON-blocking one (hsr_forward_arm_async) cannot, so the tile
// kernel leaves NaN in every output it owns — whatever is computed from them before hsr_forward_end reports the overflow is visibly
// invalid instead of silently wrong.  Thread t of the tile's workgroup: pixel (t & 15, t >> 4); channels [c0, c0 + nsem) of the
// semantic map; `base`: also colour / depth / median depth / opacity (/ mask).
__device__ __forceinline__ void hsr_poison_tile(const RenderFwdArgs& a, int tile, int t, bool base, int c0, int nsem)
{
    const int tiles_x = (a.W + HSR_TILE_X - 1) / HSR_TILE_X;
    const int px = (tile % tiles_x) * HSR_TILE_X + (t & 15), py = (tile / tiles_x) * HSR_TILE_Y + (t >> 4);
    if (px >= a.W || py >= a.H) return;
    const size_t N = (size_t)a.W * a.H, pix = (size_t)a.W * py + px;
    const float nan = __uint_as_float(0x7fc00000u);
    if (base) {
        a.out_color[pix] = nan; a.out_color[N + pix] = nan; a.out_color[2 * N + pix] = nan;
        a.out_depth[pix] = nan; a.out_median_depth[pix] = nan; a.out_opacity[pix] = nan;
        if (a.out_mask) a.out_mask[pix] = nan;
    }
    if (a.out_semantic)
        for (int c = c0; c < c0 + nsem && c < a.K; c++) a.out_semantic[(size_t)c * N + pix] = nan;
}

struct TileGeom {
    int tx, ty;        // tile coordinates
    int px, py;        // this lane's pixel
    bool inside;
    float pfx, pfy;
    float qx0, qy0;    // first pixel of this wave's quadrant (float)
};

__device__ __forceinline__ TileGeom tile_geom(int tile, int W, int H, int t)
{
    TileGeom g;
    const int tiles_x = (W + HSR_TILE_X - 1) / HSR_TILE_X;
    g.tx = tile % tiles_x;
    g.ty = tile / tiles_x;
    const int wv = t >> 6, l = t & 63;
    const int qx = (wv & 1) * 8, qy = (wv >> 1) * 8;
    g.px = g.tx * HSR_TILE_X + qx + (l & 7);
    g.py = g.ty * HSR_TILE_Y + qy + (l >> 3);
    g.inside = g.px < W && g.py < H;
    g.pfx = (float)g.px;
    g.pfy = (float)g.py;
    g.qx0 = (float)(g.tx * HSR_TILE_X + qx);
    g.qy0 = (float)(g.ty * HSR_TILE_Y + qy);
    return g;
}

// 4-bit mask of the tile's quadrants (bit q = wave q) that the splat's alpha >= 1/255 region can touch.
// xy: centre, conic (cx, cy, cz), opacity.  Conservative: boxes are inflated by a relative 1e-3 + 0.05 px.
__device__ __forceinline__ uint32_t quadrant_mask(float x, float y, float cx, float cy, float cz, float opacity, float tile_x0,
                                                  float tile_y0)
{
    // alpha >= 1/255  <=>  power >= -ln(255*o) =: -tau;  no pixel qualifies when 255*o < 1
    const float t255 = 255.0f * opacity;
    if (!(t255 >= 1.0f)) return 0u;
    const float tau2 = 2.0f * __logf(t255) * 1.001f + 1e-4f;  // 2*tau, inflated
    const float det = cx * cz - cy * cy;
    // degenerate / non-positive-definite conic: do not cull
    if (!(det > 0.0f) || !(cx > 0.0f) || !(cz > 0.0f)) return 0xFu;
    const float inv_det = 1.0f / det;
    const float hx = sqrtf(tau2 * cz * inv_det) * 1.001f + 0.05f;
    const float hy = sqrtf(tau2 * cx * inv_det) * 1.001f + 0.05f;
    const float x0 = x - hx - tile_x0, x1 = x + hx - tile_x0;  // tile-relative extent
    const float y0 = y - hy - tile_y0, y1 = y + hy - tile_y0;
    // quadrant pixel centres: [0,7] and [8,15] on each axis
    const bool xl = x0 <= 7.0f && x1 >= 0.0f, xr = x0 <= 15.0f && x1 >= 8.0f;
    const bool yt = y0 <= 7.0f && y1 >= 0.0f, yb = y0 <= 15.0f && y1 >= 8.0f;
    return (uint32_t)(xl && yt) | ((uint32_t)(xr && yt) << 1) | ((uint32_t)(xl && yb) << 2) | ((uint32_t)(xr && yb) << 3);
}

// Builds the per-quadrant compacted slot lists for one staged batch.  Lane t staged slot t (or nothing
// when t >= cnt).  s_list[q][sw*64 + k] = k-th slot staged by wave `sw` that touches quadrant q (order
// preserved); s_lcnt[q][sw] = how many.  Call between the record stores and the barrier before blending.
__device__ __forceinline__ void publish_quadrant_lists(uint32_t qmask, int t, uint8_t (*s_list)[256], uint8_t (*s_lcnt)[4])
{
    const int lane = t & 63, sw = t >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const bool on = (qmask >> q) & 1u;
        const uint64_t m = __ballot(on);
        if (on) s_list[q][sw * 64 + __popcll(m & lt)] = (uint8_t)t;
        if (lane == 0) s_lcnt[q][sw] = (uint8_t)__popcll(m);
    }
}

// Consumer side: the wave that owns quadrant `wv` concatenates its four segments into one flat list
// (wave-local LDS traffic, no barrier: LDS operations of one wave execute in order).  A flat list lets
// the blend loop prefetch the next splat's slot and record one iteration ahead, which matters because
// with 3-5 waves per SIMD an exposed LDS round trip per splat is not hidden by other waves.
__device__ __forceinline__ int build_flat_list(int wv, int lane, const uint8_t (*s_list)[256], const uint8_t (*s_lcnt)[4],
                                               uint8_t (*s_flat)[256])
{
    int total = 0;
#pragma unroll
    for (int seg = 0; seg < 4; seg++) {
        const int c = s_lcnt[wv][seg];
        if (lane < c) s_flat[wv][total + lane] = s_list[wv][seg * 64 + lane];
        total += c;
    }
    __builtin_amdgcn_wave_barrier();
    return total;
}

// ---- 4x4 sub-block decomposition (hsr_render_fwd_sub.hip) ----
// Same tile and quadrant ownership as above, but lane l of a wave owns pixel (4*(gq&1) + (l&3), 4*(gq>>1) + ((l>>2)&3)) of
// the quadrant, gq = l >> 4: every 16-lane group of the wave is one 4x4 SUB-BLOCK, and walks its own compacted list.  A
// SLAM-sized splat (alpha >= 1/255 inside a radius of 3-4 px) touches 3 quadrant visits x 64 lanes today but only ~4 of
// the 16 sub-blocks x 16 lanes: the blend loop of a wave runs max-over-its-four-groups iterations, 0.6x of the quadrant
// list at the headline workload (tools/sim in DESIGN §4).  Sub-block b = 4*wave + gq.
__device__ __forceinline__ TileGeom tile_geom_sub(int tile, int W, int H, int t)
{
    TileGeom g;
    const int tiles_x = (W + HSR_TILE_X - 1) / HSR_TILE_X;
    g.tx = tile % tiles_x;
    g.ty = tile / tiles_x;
    const int wv = t >> 6, l = t & 63, gq = l >> 4;
    const int qx = (wv & 1) * 8 + (gq & 1) * 4, qy = (wv >> 1) * 8 + (gq >> 1) * 4;
    g.px = g.tx * HSR_TILE_X + qx + (l & 3);
    g.py = g.ty * HSR_TILE_Y + qy + ((l >> 2) & 3);
    g.inside = g.px < W && g.py < H;
    g.pfx = (float)g.px;
    g.pfy = (float)g.py;
    g.qx0 = (float)(g.tx * HSR_TILE_X + qx);
    g.qy0 = (float)(g.ty * HSR_TILE_Y + qy);
    return g;
}

// 16-bit mask of the tile's sub-blocks (bit 4*wave + gq) in which some pixel CENTRE can reach alpha >= 1/255.
//
// alpha >= 1/255  <=>  Q(d) = A dx^2 + 2 B dx dy + C dy^2 <= 2 ln(255 o) =: tau  (d = splat centre - pixel centre, conic (A, B, C):
// power = -Q/2, reference forward.cu:481-496): the ellipse E.  A bounding box of E, which is what round 1 tested, keeps every
// sub-block the box overlaps; for the ~4 px splats of a SLAM map an eighth of those are corners E never reaches, for elongated
// splats a quarter (tests/sim_sublists.py, this function re-evaluated in numpy for every instance of the headline scenes: 3.67 M ->
// 3.22 M sub-block entries; anisotropic scene 8.38 M -> 6.45 M).  Exact test, one ROW of sub-blocks at a time: the slab
// dy in [lo, hi] (the row's pixel centres) cuts E in a convex set whose projection on x is an interval [xmin, xmax]; a
// sub-block of the row is touched iff its dx interval meets it.  For a fixed dy the ellipse spans
//     dx in (-B dy -+ sqrt(A tau - det dy^2)) / A,
// the upper end is concave in dy with its maximum (the ellipse's rightmost point, dx = hx) at dy+ = -(B/C) hx, the lower end
// convex with its minimum at dy- = -dy+; over the slab each is therefore attained at the slab's point nearest dy+ / dy-:
// two clamps, two square roots per row.  Conservative: tau is inflated by 0.2 % + 0.02 and the interval by 0.02 px (the
// per-pixel test in the blend loop stays exact, so a block kept in vain costs time, never accuracy; a block dropped wrongly
// would cost accuracy — the parity and fuzz suites compare every pixel with the oracle).  ~25 live registers: it is inlined
// into the staging phase of kernels that sit at the 128-register step.
__device__ __forceinline__ uint32_t subblock_mask(float x, float y, float cx, float cy, float cz, float opacity, float tile_x0,
                                                  float tile_y0, bool cull = true)
{
    const float t255 = 255.0f * opacity;
    if (!(t255 >= 1.0f)) return 0u;
    if (!cull) return 0xFFFFu;   // ablate build, HSR_DEBUG_FLAGS & 16: every sub-block visits every splat (tools/check_culling.py)
    const float tau = 2.0f * __logf(t255) * 1.002f + 0.02f;
    const float det = cx * cz - cy * cy;
    if (!(det > 0.0f) || !(cx > 0.0f) || !(cz > 0.0f)) return 0xFFFFu;
    const float inv_det = 1.0f / det, inv_a = 1.0f / cx;
    const float hx = sqrtf(tau * cz * inv_det), hy = sqrtf(tau * cx * inv_det) * 1.001f + 0.02f;   // half extents of E
    const float dyp = -(cy / cz) * hx;                  // dy of E's rightmost point; its leftmost point sits at -dyp
    const float atau = cx * tau, nb = -cy;
    const float rx = x - tile_x0, ry = y - tile_y0;     // splat centre, tile-relative
    uint32_t m = 0u;
    float ry_hi = ry;                                   // ry - 4 r
#pragma nounroll                                        // rolled on purpose: unrolled, the four rows' temporaries spill in the backward
    for (int r = 0; r < 4; r++, ry_hi -= 4.0f) {
        // the row's slab of dy = centre - pixel centre, cut to E's own extent
        const float lo = fmaxf(ry_hi - 3.0f, -hy), hi = fminf(ry_hi, hy);
        const float dyu = __builtin_amdgcn_fmed3f(dyp, lo, hi), dyl = __builtin_amdgcn_fmed3f(-dyp, lo, hi);
        const float xmax = fmaf(nb, dyu, sqrtf(fmaxf(fmaf(-det * dyu, dyu, atau), 0.0f))) * inv_a + 0.02f;
        const float xmin = fmaf(nb, dyl, -sqrtf(fmaxf(fmaf(-det * dyl, dyl, atau), 0.0f))) * inv_a - 0.02f;
        // bit of sub-block (column c, row r): wave = (r>>1)*2 + (c>>1), gq = (r&1)*2 + (c&1): the row's columns sit at bits 0, 1, 4, 5
        uint32_t rb = 0u;
#pragma unroll
        for (int c = 0; c < 4; c++)   // the sub-block's dx interval is [rx - (4c + 3), rx - 4c]
            rb |= (uint32_t)((rx - (4.0f * c + 3.0f)) <= xmax && (rx - 4.0f * c) >= xmin) << (4 * (c >> 1) + (c & 1));
        if (lo <= hi) m |= rb << (8 * (r >> 1) + 2 * (r & 1));
    }
    return m;
}

// quadrant mask (bit q = wave q) from the exact sub-block test: a quadrant is visited iff one of its four sub-blocks is
__device__ __forceinline__ uint32_t quadrant_bits(uint32_t mask)
{
    return (uint32_t)((mask & 0xFu) != 0u) | ((uint32_t)((mask & 0xF0u) != 0u) << 1) | ((uint32_t)((mask & 0xF00u) != 0u) << 2) |
           ((uint32_t)((mask & 0xF000u) != 0u) << 3);
}
__device__ __forceinline__ uint32_t quadrant_mask_exact(float x, float y, float cx, float cy, float cz, float opacity, float tile_x0,
                                                        float tile_y0)
{
    return quadrant_bits(subblock_mask(x, y, cx, cy, cz, opacity, tile_x0, tile_y0));
}

constexpr int HSR_SUB_LSTRIDE = 260;   // bytes per sub-block list: 256 slots + 4 so that the four groups of a wave hit different banks

// Staging side: s_list[b * LSTRIDE + sw*64 + k] = k-th slot staged by wave sw that touches sub-block b; s_lcnt[sw][b] = how many.
__device__ __forceinline__ void publish_subblock_lists(uint32_t mask, int t, uint8_t* s_list, uint8_t (*s_lcnt)[16])
{
    const int lane = t & 63, sw = t >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int b = 0; b < 16; b++) {
        const bool on = (mask >> b) & 1u;
        const uint64_t m = __ballot(on);
        if (on) s_list[b * HSR_SUB_LSTRIDE + sw * 64 + __popcll(m & lt)] = (uint8_t)t;
        if (lane == 0) s_lcnt[sw][b] = (uint8_t)__popcll(m);
    }
}

// Consumer side: each 16-lane group closes the gaps between the four segments of its sub-block's list, in place (every copy
// moves an entry to a lower or equal index, segment by segment in ascending order; only this group reads or writes the list
// between the two workgroup barriers).  Returns this group's list length.
__device__ __forceinline__ int flatten_sublist(int sb, int lane, uint8_t* s_list, const uint8_t (*s_lcnt)[16])
{
    const int l16 = lane & 15;
    uint8_t* list = s_list + sb * HSR_SUB_LSTRIDE;
    int total = s_lcnt[0][sb];
#pragma unroll
    for (int seg = 1; seg < 4; seg++) {
        const int c = s_lcnt[seg][sb];
        if (total != seg * 64) {
#pragma unroll
            for (int i = 0; i < 64; i += 16) {
                const bool mv = i + l16 < c;
                const uint8_t v = list[seg * 64 + i + l16];
                if (mv) list[total + i + l16] = v;
            }
        }
        total += c;
    }
    __builtin_amdgcn_wave_barrier();
    return total;
}

// ---- packed per-Gaussian gradient row (backward, default accumulation mode) ----
// The reference keeps the per-Gaussian sums in six separate arrays (dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors,
// dL_ddepths, dL_dsemantics: rasterize_points.cu:378-388), so one (tile, Gaussian) update touches six cache lines.
// Float atomics on MI355X execute at the memory side in 64-byte requests at a fixed chip-wide rate, so the
// number of LINES touched is what an update costs.  Packed layout, one row per Gaussian, 64-byte aligned:
//   line 0 : col 0,1 mean2D.xy | 2,3,4 conic.xyw | 5 opacity (alpha path, or total) | 6 depth (median part, or total)
//   line 1+: col 16 + c semantic channel c (c < K), then the "direct" sums at col 16 + K + {0,1,2} rgb,
//            + 3 depth (direct part), + 4 opacity (direct part) — directly behind the semantics so that the
//            matrix-core kernel's second 16-channel group (sem 16..25, r, g, b, depth, opacity at K = 26) is ONE line.
// preprocess_backward_kernel unpacks the row into the reference's arrays.
#define HSR_GROW_SEM0 16
__host__ __device__ inline int hsr_grow_stride(int K) { return 16 + 16 * ((K + 5 + 15) / 16); }
__host__ __device__ inline int hsr_grow_direct0(int K) { return HSR_GROW_SEM0 + K; }
// Layout 1 ("compact", round 4; hsr_render_bwd_q.hip only): what an update costs is the number of 64-byte LINES it touches, and line 0 has nine
// free columns.  The K + 5 channel columns [sem 0 .. K-1, r, g, b, depth(direct), opacity(direct)] are split: the LAST min(9, K + 5) of them
// ride in columns 7..15 of line 0, the first R = K + 5 - that many follow from column 16.  K = 0: ONE line per row instead of two;
// 12 <= K <= 20 (the NYU40 tree's K = 16): two instead of three.  hsr_grow_col gives the column of channel column ch under either layout.
__host__ __device__ inline int hsr_grow_nl0(int K) { return K + 5 < 9 ? K + 5 : 9; }
__host__ __device__ inline int hsr_grow_stride_l(int layout, int K)
{
    if (!layout) return hsr_grow_stride(K);
    const int R = K + 5 - hsr_grow_nl0(K);
    return 16 + 16 * ((R + 15) / 16);
}
__host__ __device__ inline int hsr_grow_col(int layout, int K, int ch)
{
    if (!layout) return HSR_GROW_SEM0 + ch;
    const int R = K + 5 - hsr_grow_nl0(K);
    return ch < R ? HSR_GROW_SEM0 + ch : 7 + (ch - R);
}
__host__ __device__ inline bool hsr_grow_compact_pays(int K) { return hsr_grow_stride_l(1, K) < hsr_grow_stride(K); }
