// hsr_render_bwd_sub.hip — backward tile kernel whose 16-lane groups walk 4x4 SUB-BLOCK lists (packed mode, K <= 27).
//
// Same per-pixel semantics as hsr_render_bwd.hip (reference backward.cu:472-899, see that file's header) and the same
// matrix-core contraction for the K+5 "direct" sums as experiments/hsr_render_bwd_mfma.hip (round 1).  What changes is WHO visits a splat.
// In the quadrant kernels a wave walks one list and all 64 lanes evaluate every entry, but a SLAM-sized splat reaches
// alpha >= 1/255 on ~40 pixels: 15-20 % of the lanes of its ~3 quadrant visits do useful work.  Here every 16-lane group
// of a wave owns one 4x4 sub-block (hsr_tile_common.h, tile_geom_sub) and only visits the splats whose alpha bounding box
// touches THAT sub-block.
//
// The wave still walks its QUADRANT list, 16 entries (a "chunk") at a time: one ballot gives every group the 16-bit mask
// of the chunk's entries that touch its sub-block, each group then steps through its own set bits, and the wave iterates
// max-over-groups times — 0.69x the entries at the headline workload (0.61x with unbounded chunks), with the same
// instructions per iteration (the four groups read four different staged records; LDS serves a b128 read 16 lanes per
// pass anyway).  Chunks keep the merge of a splat's sums over the quadrant where it costs nothing:
//   * panel row r belongs to chunk entry r; a group that visits the entry writes its 16 weights into its 16 columns of
//     that row (the rest of the row is zero), and the matrix cores contract over all 64 pixels as before:
//     D[entry][channel] is already the quadrant's sum, the 4 groups never meet in an atomic;
//   * the 7 values that are not of the form sum_pixels w*g are reduced over the 16 lanes of the group (four in-row
//     butterfly stages) into a per-(entry, group) slot and summed over the groups once per chunk;
//   * the chunk's rows leave with one atomic per accumulator register (4 rows x 64 bytes each) plus two for the butterfly
//     columns (8 rows x 7 values each): the same 3 requests per (splat, quadrant) as experiments/hsr_render_bwd_mfma.hip (round 1).
// Merging further, over the TILE, would cut the atomic requests by a quarter — they execute at the memory side at a
// fixed rate (MI355X guide, "Global float atomics"; twice the requests = +0.15 ms here) — but the only place the four
// waves can meet is an LDS table, and ds_add_f32 costs ~170 cycles per wave-instruction on gfx950 (measured with exactly
// that design: 1.03 ms, of which 0.68 ms LDS atomics).
#include "hsr_tile_common.h"
#include "hsr_wave_reduce.h"
#include <stdlib.h>
#include <string.h>

#ifdef HSR_TRACE
// Diagnostic build only (make -C hier-slam_amd/csrc trace -> libhsr_rast_trace.so, tools/trace_bwd.py): per-wave cycle counts of
// the phases of render_bwd_sub_kernel, clock64 deltas accumulated in registers and dumped at the end.  Never in the product.
#define HSR_TRACE_SLOTS 8
__device__ unsigned long long g_hsr_trace_sub[16384 * HSR_TRACE_SLOTS];
extern "C" int hsr_debug_read_trace_sub(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hsr_trace_sub), sizeof(unsigned long long) * (size_t)n);
}
#define TR_NOW() clock64()
#define TR_ADD(acc, t0) (acc) += (unsigned long long)(clock64() - (t0))
#else
#define TR_NOW() 0ll
#define TR_ADD(acc, t0) ((void)0)
#endif

namespace {

// Makes the staging registers of the next batch "used" BEFORE the first atomics of this batch are issued: hipcc then waits for their
// loads here — they were issued a chunk of blending ago and have landed — instead of at the next batch's staging, where the same
// s_waitcnt would also have to sit out every atomic issued in between (loads, stores and atomics retire through one in-order counter).
#define HSR_SETTLE_STAGING()                                                                                                   \
    asm volatile("" ::"v"(id_next), "v"(p_xy.x), "v"(p_xy.y), "v"(p_co.x), "v"(p_co.y), "v"(p_co.z), "v"(p_co.w), "v"(p_r), "v"(p_g), \
                 "v"(p_b), "v"(p_d), "v"(p_mask))

// The forward's staging phase leaves the 16-bit sub-block mask of every list entry it stages in RenderBwdArgs::masks (the backward
// stages a subset of those entries: it stops at the tile's largest n_contrib): one 4-byte load per entry instead of ~460 instructions of
// subblock_mask per staged splat, a tenth of this kernel's VALU work.  The diagnostic build keeps deriving it (its experimental forward
// kernels do not write masks, and HSR_DEBUG_FLAGS=16 switches the culling off per call).
#ifdef HSR_ABLATE
constexpr bool SAVED_MASKS = false;
#else
constexpr bool SAVED_MASKS = true;
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// orders the LDS accesses of ONE wave (stores before it are visible to the wave's loads after it); no workgroup barrier
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int SB_SLOTS = 16;        // quadrant-list entries per chunk (M dimension of the matrix-core flush)
constexpr int SB_STRIDE = 66;       // floats per panel row
constexpr int SB_PANEL = 64 * 17;   // floats per wave: max(16 * 66, 64 * 17 for the G transpose)
constexpr int SB_NV = 7;

// Sums each of the 7 per-lane values over the 16 lanes of the lane's ROW (16-lane group); a lane with bit 1 clear returns
// the total of v[((l >> 2) & 3) | ((l & 1) << 2)], the others return don't-care.  First four stages of
// wave_reduce_transpose (hsr_wave_reduce.h).
__device__ __forceinline__ float row_reduce_transpose7(const float (&v)[7], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float a[4], b[2];
    pair2_half_mirror(v[0], v[1], v[2], v[3], a[0], a[1]);
    pair2_half_mirror(v[4], v[5], v[6], 0.f, a[2], a[3]);
    pair2_ror8(a[0], a[1], a[2], a[3], b[0], b[1]);
    (void)b2;
    const float c = pair_dpp<DPP_QUAD_XOR1>(b[0], b[1], b0);
    return pair_dpp<DPP_QUAD_XOR2>(c, 0.f, b1);
}

// ---- fp32 products on the bf16 matrix cores: the exact three-way split ----
// v_mfma_f32_16x16x4_f32 is the slow matrix instruction of gfx950 (32 cycles for 1024 multiply-adds, measured: tools/micro/valu_rate.hip);
// v_mfma_f32_16x16x32_bf16 does 8192 in 16.  An fp32 number is EXACTLY hi + mid + lo with three bf16 numbers of 8 significant bits
// each (truncate, subtract, truncate, subtract: the 24-bit significand is cut in three), so a * b = sum of nine bf16 products, each exact
// in the fp32 accumulator; the six kept here (all but mid*lo, lo*mid, lo*lo) leave a relative error <= 2^-23 per product — the size of
// one fp32 rounding.  Six 16-cycle instructions over 32 pixels replace eight 32-cycle ones over 4 pixels each.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3_bf16(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo)
{
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const uint32_t a0 = __float_as_uint(x[e]), a1 = __float_as_uint(x[e + 1]);
        const float r0 = x[e] - __uint_as_float(a0 & 0xFFFF0000u), r1 = x[e + 1] - __uint_as_float(a1 & 0xFFFF0000u);
        const uint32_t b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
        const float t0 = r0 - __uint_as_float(b0 & 0xFFFF0000u), t1 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
        // the upper halves of two registers, packed (element e in the low half)
        hi[e / 2] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
        mid[e / 2] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        lo[e / 2] = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
    }
}

__device__ __forceinline__ f32x4 mma_bf16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int KC, int BATCH>
__global__ void __launch_bounds__(256, 4) render_bwd_sub_kernel(RenderBwdArgs a)
{
    constexpr int NCH = KC + 5;               // sem[KC], r, g, b, depth, opacity(direct)
    constexpr int NG = (NCH + 15) / 16;       // 16-channel groups
    static_assert(NG <= 2, "at most 32 direct channels per launch");
    static_assert(BATCH <= 256, "batch slots are bytes");
    // one 48-byte record per staged splat { x, y, A', B' | r, g, b, depth | C', opacity, -, - }: ONE address computation per visit
    __shared__ float4 s_ent[3 * BATCH];
    __shared__ int s_id[BATCH];
    __shared__ uint16_t s_mask[BATCH];                  // sub-block mask of each staged splat
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ uint8_t s_flat[4][256];
    __shared__ int s_wmax[4];
    __shared__ float s_panel[4][SB_PANEL];
    __shared__ uint8_t s_cj[4][SB_SLOTS];               // batch slot of each chunk row
    __shared__ __attribute__((aligned(16))) uint32_t s_cid[4][SB_SLOTS];   // packed-row offset (Gaussian id x row stride) of each chunk row
    __shared__ __attribute__((aligned(16))) float s_u7[4][SB_SLOTS * 8 * 4];   // [row][value 0..7][group]: butterfly sums; the four groups of a value are ONE 16-byte read at emission

    const int tile = hsr_block_tile(blockIdx.x, ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y));
    if (tile >= ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, gq = lane >> 4, l16 = lane & 15;
    const TileGeom tg = tile_geom_sub(tile, a.W, a.H, t);
    const bool inside = tg.inside;
    const size_t N = (size_t)a.W * a.H;
    const size_t pix_id = (size_t)a.W * tg.py + tg.px;
    const float pfx = tg.pfx, pfy = tg.pfy;
    const float tile_x0 = (float)(tg.tx * HSR_TILE_X), tile_y0 = (float)(tg.ty * HSR_TILE_Y);
    const uint2 range = a.ranges[tile];
    float* panel = s_panel[wv];
    float* u7 = s_u7[wv];
    unsigned long long tr_stage = 0, tr_loop = 0, tr_flush = 0, tr_iters = 0, tr_chunks = 0, tr_accepted = 0;
    const long long tr_t0 = TR_NOW();
    (void)tr_stage; (void)tr_loop; (void)tr_flush; (void)tr_iters; (void)tr_chunks; (void)tr_accepted; (void)tr_t0;

    // every prologue load unconditional and issued before anything consumes one (see experiments/hsr_render_bwd_mfma.hip)
    const size_t pix_ld = inside ? pix_id : 0;
    const float inm = inside ? 1.f : 0.f;
    const float T_final_ld = a.final_T[pix_ld];
    const int last_contributor_ld = (int)a.n_contrib[pix_ld];
    const int median_at_ld = (int)a.median_pos[pix_ld];
    float dpx0 = a.dL_dpix[pix_ld], dpx1 = a.dL_dpix[N + pix_ld], dpx2 = a.dL_dpix[2 * N + pix_ld];
    float dpd = a.dL_dpix_depth[pix_ld], dpm = a.dL_dpix_median[pix_ld], dpo = a.dL_dpix_opacity[pix_ld];
    float semv[KC > 0 ? KC : 1];
#pragma unroll
    for (int c = 0; c < KC; c++) semv[c] = a.dL_dpix_sem[(size_t)min(c, a.K - 1) * N + pix_ld];
    dpx0 *= inm; dpx1 *= inm; dpx2 *= inm; dpd *= inm; dpm *= inm; dpo *= inm;
    const float T_final = T_final_ld * inm;
    float T = T_final;
    const int last_contributor = inside ? last_contributor_ld : 0;
    const int median_at = (inside ? median_at_ld : 0) - 1;   // list position of the forward's T = 0.5 crossing (-1: none): gets dL_dmedian_depth

    int wmax = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    if (lane == 0) s_wmax[wv] = wmax;

    // ---- the MFMA B operand: G transposed through LDS (lane l holds G[pixel lane 4m + (l>>4)][channel 16g + (l&15)]) ----
    float Breg[NG][16];
    {
        float gv[NG][16];
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int ch = 16 * g + c;
                float v = 0.f;
                if (ch < KC) {
                    v = (ch < a.K) ? semv[ch < KC ? ch : 0] * inm : 0.f;
                } else if (ch == KC) v = dpx0;
                else if (ch == KC + 1) v = dpx1;
                else if (ch == KC + 2) v = dpx2;
                else if (ch == KC + 3) v = dpd;
                else if (ch == KC + 4) v = dpo;
                gv[g][c] = v;
            }
        // the panel is private to the wave: a wave-level fence orders its LDS stores and loads, the four waves do not have to meet
        // (they would wait for the slowest wave's 30-odd global loads twice per channel group)
#pragma unroll
        for (int g = 0; g < NG; g++) {
#pragma unroll
            for (int c = 0; c < 16; c++) panel[lane * 17 + c] = gv[g][c];
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < 16; m++) Breg[g][m] = panel[(4 * m + (lane >> 4)) * 17 + (lane & 15)];
            wave_lds_fence();
        }
    }
    __syncthreads();   // s_wmax
    const int hi_all = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    const long long tr_t1 = TR_NOW();   // end of the prologue
    (void)tr_t1;

    const float bg_dot = a.bg[0] * dpx0 + a.bg[1] * dpx1 + a.bg[2] * dpx2;
    const float kx = (0.5f * a.W) / HSR_LOG2E, ky = (0.5f * a.H) / HSR_LOG2E;
    float Racc = 0.f;   // the reference's accum_rec AFTER the last accepted splat: last_alpha * last_h + (1 - last_alpha) * accum_rec (backward.cu:630-640, h = colour . dL_dpixel)

    // butterfly value this lane holds after row_reduce_transpose7, or -1
    const int myv = (lane & 2) ? -1 : (((lane >> 2) & 3) | ((lane & 1) << 2));
    const bool myv_on = myv >= 0 && myv < SB_NV;
    // packed-row columns of the accumulator columns this lane holds (col = lane & 15 of channel group g), or -1
    int colg[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int ch = 16 * g + l16;
        colg[g] = ch < KC ? (ch < a.K ? HSR_GROW_SEM0 + ch : -1) : (ch < KC + 5 ? hsr_grow_direct0(a.K) + (ch - KC) : -1);
    }

    // zeroes the panel rows and the butterfly slots of the next chunk (wave-private LDS: no barrier)
    auto clear_chunk = [&]() {
        float2* p2 = reinterpret_cast<float2*>(panel);
#pragma unroll
        for (int i = 0; i < (SB_SLOTS * SB_STRIDE) / 2; i += 64)
            if (i + lane < (SB_SLOTS * SB_STRIDE) / 2) p2[i + lane] = make_float2(0.f, 0.f);
        float4* u4 = reinterpret_cast<float4*>(u7);
#pragma unroll
        for (int i = 0; i < (SB_SLOTS * 4 * 8) / 4; i += 64) u4[i + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // the chunk's 16 panel rows -> D[16 entries][16*NG channels] -> packed rows; butterfly slots -> columns 0..6
    auto flush = [&](int nrows) {
        f32x4 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* arow = panel + l16 * SB_STRIDE + (lane >> 4);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const float av = arow[4 * m];
#pragma unroll
            for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Breg[g][m], acc[g], 0, 0, 0);
        }
        // D[row = 4*(lane>>4) + r][col = lane&15]: one atomic wave-instruction per register = 4 rows x 64 bytes
        {
            const uint4 b4 = *reinterpret_cast<const uint4*>(&s_cid[wv][4 * (lane >> 4)]);   // the four row offsets in one LDS read
            const uint32_t bb[4] = {b4.x, b4.y, b4.z, b4.w};
            const int nr = nrows - 4 * (lane >> 4);   // how many of this lane's four rows exist
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int g = 0; g < NG; g++)
                    if (r < nr && colg[g] >= 0 && acc[g][r] != 0.f && !(a.debug_flags & 1))
                        atomicAdd(a.grow + (bb[r] + (uint32_t)colg[g]), acc[g][r]);
        }
        // butterfly columns 0..6: two wave-instructions of 8 rows x 7 values, so that each row's line is ONE request
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int row = (lane >> 3) + 8 * pass, vi = lane & 7;
            // sums over the four groups' slots -> the reference's sums (backward.cu:648-660, :887-893):
            //   dL_dmean2D.{x,y} = 2 {kx, ky} x the per-pixel combined sums;  dL_dconic.{x,y,w} = -0.5 {Sxx, Sxy, Syy} (raw moments of q);
            //   columns 5, 6 (opacity, median depth) pass through
            const float4 sa = *reinterpret_cast<const float4*>(u7 + row * 32 + vi * 4);
            const float ta = (sa.x + sa.y) + (sa.z + sa.w);
            const float wa = vi == 0 ? 2.0f * kx : (vi == 1 ? 2.0f * ky : (vi <= 4 ? -0.5f : 1.0f));
            const float val = wa * ta;
            const uint32_t base = s_cid[wv][row] + (uint32_t)vi;
            if (vi < SB_NV && row < nrows && val != 0.f && !(a.debug_flags & 1)) atomicAdd(a.grow + base, val);
        }
    };

    // ---- software-pipelined staging ----
    int id_next = 0, id_cur = 0;
    float2 p_xy = {0, 0};
    float4 p_co = {0, 0, 0, 0};
    float p_r = 0, p_g = 0, p_b = 0, p_d = 0;
    uint32_t p_mask = 0u;
    // unconditional, clamped staging loads, the id of the batch after next requested before the next batch's records: see
    // render_fwd_kernel (a load inside a divergent `if`, or into a register the loads before it took their addresses from, is waited
    // for where it is issued — and the (rec == NULL) fallback kept three of these values in a scratch slot).  a.rec is never NULL.
    const int n_list = (int)(range.y - range.x);
    auto fetch_id = [&](int hi) -> int { return (int)a.point_list[range.x + min(max(hi - 1 - t, 0), max(n_list - 1, 0))]; };
    auto fetch_mask = [&](int hi) -> uint32_t { return SAVED_MASKS ? a.masks[range.x + min(max(hi - 1 - t, 0), max(n_list - 1, 0))] : 0u; };
    auto load_record = [&](int id_of) {
        const size_t id = (size_t)id_of;
        id_cur = id_of;
        const float4* rec = a.rec + 4 * id;
        const float4 r0 = rec[0], r2 = rec[2];
        p_co = rec[1];
        p_xy = make_float2(r0.x, r0.y);
        p_d = r0.z;
        p_r = r2.x; p_g = r2.y; p_b = r2.z;
    };
    if (n_list > 0) {
        const int id0 = fetch_id(hi_all);
        id_next = fetch_id(hi_all - BATCH);
        load_record(id0);
        p_mask = fetch_mask(hi_all);
    }

    for (int hi = hi_all; hi > 0; hi -= BATCH) {
        const int cnt = min(BATCH, hi);
        // list position of batch slot j is hi - 1 - j: "behind the last contributor" and "the median splat" as slot tests, per batch
        const int j_first = hi - last_contributor, j_median = hi - 1 - median_at;
        const long long ts = TR_NOW();
        (void)ts;
        __syncthreads();
        uint32_t qmask = 0u;
        if (t < cnt) {
            const uint32_t mask = SAVED_MASKS ? p_mask
                                              : subblock_mask(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0, !(a.debug_flags & 16));
            qmask = (uint32_t)((mask & 0xFu) != 0u) | ((uint32_t)((mask & 0xF0u) != 0u) << 1) | ((uint32_t)((mask & 0xF00u) != 0u) << 2) |
                    ((uint32_t)((mask & 0xF000u) != 0u) << 3);
            s_mask[t] = (uint16_t)mask;
            s_id[t] = id_cur;
            s_ent[3 * t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_ent[3 * t + 1] = make_float4(p_r, p_g, p_b, p_d);
            s_ent[3 * t + 2] = make_float4((-0.5f * HSR_LOG2E) * p_co.z, p_co.w, (-0.5f * HSR_LOG2E) * p_co.y, 0.f);   // C', opacity, B' / 2
        }
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        __syncthreads();
        {
            const int id_use = id_next;            // ids of the next batch, requested a whole batch ago
            id_next = fetch_id(hi - 2 * BATCH);
            load_record(id_use);
            p_mask = fetch_mask(hi - BATCH);
        }
        TR_ADD(tr_stage, ts);
        if (hi - cnt >= wmax) {   // this wave's pixels all stopped in front of this batch
            HSR_SETTLE_STAGING();
            continue;
        }
        const long long tl = TR_NOW();
        (void)tl;

        const int total = build_flat_list(wv, lane, s_list, s_lcnt, s_flat);
        if (total == 0) HSR_SETTLE_STAGING();
        for (int c0 = 0; c0 < total; c0 += SB_SLOTS) {
            const int nrows = min(SB_SLOTS, total - c0);
            // lane (group gq, row l16): does chunk entry l16 touch sub-block (wv, gq)?
            const int jr = l16 < nrows ? (int)s_flat[wv][c0 + l16] : 0;
            const bool touch = l16 < nrows && ((s_mask[jr] >> (4 * wv + gq)) & 1u);
            const uint64_t ball = __ballot(touch);
            if (gq == 0) {
                s_cj[wv][l16] = (uint8_t)jr;
                s_cid[wv][l16] = (uint32_t)s_id[jr] * (uint32_t)a.grow_stride;   // once per chunk row, not once per emitted register
            }
            clear_chunk();
            // wave-uniform (readfirstlane: the loop counter then lives in a scalar register, not in a VALU down-counter)
            const int iters = __builtin_amdgcn_readfirstlane(max(max(__popc((uint32_t)ball & 0xFFFFu), __popc((uint32_t)(ball >> 16) & 0xFFFFu)),
                                                                 max(__popc((uint32_t)(ball >> 32) & 0xFFFFu), __popc((uint32_t)(ball >> 48)))));
            uint32_t todo = (uint32_t)(ball >> (16 * gq)) & 0xFFFFu;   // this group's entries, visited in list order
            int r_next = todo ? __builtin_ctz(todo) : 0;
            int j_next = s_cj[wv][r_next];
            for (int it = 0; it < iters; it++) {
                const bool valid = todo != 0u;
                const int r = r_next, j = j_next;
                todo &= todo - 1u;
                r_next = todo ? __builtin_ctz(todo) : 0;
                j_next = s_cj[wv][r_next];
                const float4* ent = &s_ent[3 * j];
                const float4 g = ent[0];
                const float4 cd = ent[1];
                const float4 co4 = ent[2];
                const float2 co = make_float2(co4.x, co4.y);
                const float hB = co4.z;   // B' / 2
                asm volatile("" ::"v"(cd.x), "v"(cd.y), "v"(cd.z), "v"(cd.w));
                const float dx = g.x - pfx, dy = g.y - pfy;
                const float dxx = dx * dx, dxy = dx * dy, dyy = dy * dy;
                const float power2 = fmaf(co.x, dyy, fmaf(g.w, dxy, g.z * dxx));
                const float G = __builtin_amdgcn_exp2f(power2);
                const float alpha = fminf(0.99f, co.y * G);
                const bool active = valid && j >= j_first && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
#ifdef HSR_TRACE
                {   // (group, entry) visits, and those in which at least one of the group's 16 pixels accepts the splat
                    const uint64_t bv = __ballot(valid), ba = __ballot(active);
#pragma unroll
                    for (int gg = 0; gg < 4; gg++) {
                        tr_iters += ((bv >> (16 * gg)) & 0xFFFFull) != 0ull;
                        tr_accepted += ((ba >> (16 * gg)) & 0xFFFFull) != 0ull;
                    }
                }
#endif
                if (__ballot(active) == 0ull) continue;

                const float inv_one_m_a = __builtin_amdgcn_rcpf(1.0f - alpha);
                const float test_T = T * inv_one_m_a;
                const float w = active ? alpha * test_T : 0.f;
                if (valid) panel[r * SB_STRIDE + lane] = w;

                const float h = fmaf(cd.x, dpx0, fmaf(cd.y, dpx1, fmaf(cd.z, dpx2, fmaf(cd.w, dpd, dpo))));
                const float Rn = Racc;
                float dL_dalpha = (h - Rn) * test_T;
                dL_dalpha += (-T_final * inv_one_m_a) * bg_dot;
                const float Gs = active ? G : 0.f;
                const float gda = Gs * dL_dalpha;
                const float q = co.y * gda;
                // The three conic sums are the RAW second moments of q over the pixels (sum q dx^2, q dx dy, q dy^2) times -0.5, and the
                // mean2D sums carry the constant factors 2 kx / 2 ky: those coefficients are applied once per (chunk row, value) when
                // the row is emitted (flush), not per pixel.
                float v[SB_NV];
                // dL_dmean2D: the two terms combined PER PIXEL, as the reference does (backward.cu:887-888).  Round 2 summed the raw moments
                // q dx and q dy and combined them at emission; for elongated splats A' dx and B' dy / 2 largely cancel, and cancelling AFTER
                // the fp32 sums over the pixels cost up to 15x the error in these two sums (found with the truth build of the oracle:
                // tests/test_gpu_truth.py) — which the per-Gaussian chain then amplifies into dL_dscales / dL_dmeans3D.
                v[0] = q * fmaf(g.z, dx, hB * dy);    // (A' dx + B' dy / 2): x kx * 2 at emission
                v[1] = q * fmaf(co.x, dy, hB * dx);   // (C' dy + B' dx / 2)
                v[2] = q * dxx;
                v[3] = q * dxy;
                v[4] = q * dyy;
                v[5] = gda;
                v[6] = (active && j == j_median) ? dpm : 0.f;
                if (active) {
                    Racc = fmaf(alpha, h - Rn, Rn);   // evaluated now instead of at the next visit: one select instead of three
                    T = test_T;
                }
                const float total7 = row_reduce_transpose7(v, lane);
                if (myv_on && valid) u7[r * 32 + myv * 4 + gq] = total7;
            }
            {
                const long long tf = TR_NOW();
                (void)tf;
                if (c0 == 0) HSR_SETTLE_STAGING();
            flush(nrows);
                TR_ADD(tr_flush, tf);
#ifdef HSR_TRACE
                tr_chunks++;
#endif
            }
        }
        TR_ADD(tr_loop, tl);
    }
#ifdef HSR_TRACE
    if (lane == 0) {
        const int wid = tile * 4 + wv;
        if (wid < 16384) {
            unsigned long long* o = g_hsr_trace_sub + (size_t)wid * HSR_TRACE_SLOTS;
            o[0] = (unsigned long long)(clock64() - tr_t0);
            o[1] = (unsigned long long)(tr_t1 - tr_t0);
            o[2] = tr_stage; o[3] = tr_loop; o[4] = tr_flush; o[5] = tr_chunks; o[6] = tr_iters; o[7] = tr_accepted;
        }
    }
#endif
}

// Geometry-only variant: the caller wants no gradient for colours, opacities or semantics (a TRACKING iteration of Hier-SLAM:
// only the camera pose is optimised, scripts/hierslam.py:1683-1860, so autograd asks for dL_dmeans3D / dL_dmeans2D alone).
// Then no sum of the form sum_pixels w*g is needed except the depth one, which joins the median-depth term in column 6:
// no upstream semantic gradients are read, no panel, no matrix cores, and a row is ONE 64-byte line (columns 0..6 of a
// 16-float row) instead of three — a third of the atomic requests.
template <int BATCH>
__global__ void __launch_bounds__(256, 4) render_bwd_geo_kernel(RenderBwdArgs a)
{
    static_assert(BATCH <= 256, "batch slots are bytes");
    // one 48-byte record per staged splat { x, y, A', B' | r, g, b, depth | C', opacity, -, - }: ONE address computation per visit
    __shared__ float4 s_ent[3 * BATCH];
    __shared__ int s_id[BATCH];
    __shared__ uint16_t s_mask[BATCH];                  // sub-block mask of each staged splat
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ uint8_t s_flat[4][256];
    __shared__ int s_wmax[4];
    __shared__ uint8_t s_cj[4][SB_SLOTS];               // batch slot of each chunk row
    __shared__ __attribute__((aligned(16))) uint32_t s_cid[4][SB_SLOTS];   // packed-row offset (Gaussian id x row stride) of each chunk row
    __shared__ __attribute__((aligned(16))) float s_u7[4][SB_SLOTS * 8 * 4];   // [row][value 0..7][group]: butterfly sums; the four groups of a value are ONE 16-byte read at emission

    const int tile = hsr_block_tile(blockIdx.x, ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y));
    if (tile >= ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, gq = lane >> 4, l16 = lane & 15;
    const TileGeom tg = tile_geom_sub(tile, a.W, a.H, t);
    const bool inside = tg.inside;
    const size_t N = (size_t)a.W * a.H;
    const size_t pix_id = (size_t)a.W * tg.py + tg.px;
    const float pfx = tg.pfx, pfy = tg.pfy;
    const float tile_x0 = (float)(tg.tx * HSR_TILE_X), tile_y0 = (float)(tg.ty * HSR_TILE_Y);
    const uint2 range = a.ranges[tile];
    float* u7 = s_u7[wv];

    // every prologue load unconditional and issued before anything consumes one (see experiments/hsr_render_bwd_mfma.hip)
    const size_t pix_ld = inside ? pix_id : 0;
    const float inm = inside ? 1.f : 0.f;
    const float T_final_ld = a.final_T[pix_ld];
    const int last_contributor_ld = (int)a.n_contrib[pix_ld];
    const int median_at_ld = (int)a.median_pos[pix_ld];
    float dpx0 = a.dL_dpix[pix_ld], dpx1 = a.dL_dpix[N + pix_ld], dpx2 = a.dL_dpix[2 * N + pix_ld];
    float dpd = a.dL_dpix_depth[pix_ld], dpm = a.dL_dpix_median[pix_ld], dpo = a.dL_dpix_opacity[pix_ld];
    dpx0 *= inm; dpx1 *= inm; dpx2 *= inm; dpd *= inm; dpm *= inm; dpo *= inm;
    const float T_final = T_final_ld * inm;
    float T = T_final;
    const int last_contributor = inside ? last_contributor_ld : 0;
    const int median_at = (inside ? median_at_ld : 0) - 1;   // list position of the forward's T = 0.5 crossing (-1: none): gets dL_dmedian_depth

    int wmax = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    if (lane == 0) s_wmax[wv] = wmax;

    __syncthreads();
    const int hi_all = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    const float bg_dot = a.bg[0] * dpx0 + a.bg[1] * dpx1 + a.bg[2] * dpx2;
    const float kx = (0.5f * a.W) / HSR_LOG2E, ky = (0.5f * a.H) / HSR_LOG2E;
    float Racc = 0.f;   // the reference's accum_rec AFTER the last accepted splat: last_alpha * last_h + (1 - last_alpha) * accum_rec (backward.cu:630-640, h = colour . dL_dpixel)

    // butterfly value this lane holds after row_reduce_transpose7, or -1
    const int myv = (lane & 2) ? -1 : (((lane >> 2) & 3) | ((lane & 1) << 2));
    const bool myv_on = myv >= 0 && myv < SB_NV;
    // zeroes the panel rows and the butterfly slots of the next chunk (wave-private LDS: no barrier)
    auto clear_chunk = [&]() {
        float4* u4 = reinterpret_cast<float4*>(u7);
#pragma unroll
        for (int i = 0; i < (SB_SLOTS * 4 * 8) / 4; i += 64) u4[i + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // the chunk's 16 panel rows -> D[16 entries][16*NG channels] -> packed rows; butterfly slots -> columns 0..6
    auto flush = [&](int nrows) {
        // butterfly columns 0..6: two wave-instructions of 8 rows x 7 values, so that each row's line is ONE request
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int row = (lane >> 3) + 8 * pass, vi = lane & 7;
            const float4 sa = *reinterpret_cast<const float4*>(u7 + row * 32 + vi * 4);
            const float ta = (sa.x + sa.y) + (sa.z + sa.w);
            const float wa = vi == 0 ? 2.0f * kx : (vi == 1 ? 2.0f * ky : (vi <= 4 ? -0.5f : 1.0f));
            const float val = wa * ta;
            const uint32_t base = s_cid[wv][row] + (uint32_t)vi;
            if (vi < SB_NV && row < nrows && val != 0.f && !(a.debug_flags & 1)) atomicAdd(a.grow + base, val);
        }
    };

    // ---- software-pipelined staging ----
    int id_next = 0, id_cur = 0;
    float2 p_xy = {0, 0};
    float4 p_co = {0, 0, 0, 0};
    float p_r = 0, p_g = 0, p_b = 0, p_d = 0;
    uint32_t p_mask = 0u;
    // unconditional, clamped staging loads, the id of the batch after next requested before the next batch's records: see
    // render_fwd_kernel (a load inside a divergent `if`, or into a register the loads before it took their addresses from, is waited
    // for where it is issued — and the (rec == NULL) fallback kept three of these values in a scratch slot).  a.rec is never NULL.
    const int n_list = (int)(range.y - range.x);
    auto fetch_id = [&](int hi) -> int { return (int)a.point_list[range.x + min(max(hi - 1 - t, 0), max(n_list - 1, 0))]; };
    auto fetch_mask = [&](int hi) -> uint32_t { return SAVED_MASKS ? a.masks[range.x + min(max(hi - 1 - t, 0), max(n_list - 1, 0))] : 0u; };
    auto load_record = [&](int id_of) {
        const size_t id = (size_t)id_of;
        id_cur = id_of;
        const float4* rec = a.rec + 4 * id;
        const float4 r0 = rec[0], r2 = rec[2];
        p_co = rec[1];
        p_xy = make_float2(r0.x, r0.y);
        p_d = r0.z;
        p_r = r2.x; p_g = r2.y; p_b = r2.z;
    };
    if (n_list > 0) {
        const int id0 = fetch_id(hi_all);
        id_next = fetch_id(hi_all - BATCH);
        load_record(id0);
        p_mask = fetch_mask(hi_all);
    }

    for (int hi = hi_all; hi > 0; hi -= BATCH) {
        const int cnt = min(BATCH, hi);
        // list position of batch slot j is hi - 1 - j: "behind the last contributor" and "the median splat" as slot tests, per batch
        const int j_first = hi - last_contributor, j_median = hi - 1 - median_at;
        __syncthreads();
        uint32_t qmask = 0u;
        if (t < cnt) {
            const uint32_t mask = SAVED_MASKS ? p_mask
                                              : subblock_mask(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0, !(a.debug_flags & 16));
            qmask = (uint32_t)((mask & 0xFu) != 0u) | ((uint32_t)((mask & 0xF0u) != 0u) << 1) | ((uint32_t)((mask & 0xF00u) != 0u) << 2) |
                    ((uint32_t)((mask & 0xF000u) != 0u) << 3);
            s_mask[t] = (uint16_t)mask;
            s_id[t] = id_cur;
            s_ent[3 * t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_ent[3 * t + 1] = make_float4(p_r, p_g, p_b, p_d);
            s_ent[3 * t + 2] = make_float4((-0.5f * HSR_LOG2E) * p_co.z, p_co.w, (-0.5f * HSR_LOG2E) * p_co.y, 0.f);   // C', opacity, B' / 2
        }
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        __syncthreads();
        {
            const int id_use = id_next;            // ids of the next batch, requested a whole batch ago
            id_next = fetch_id(hi - 2 * BATCH);
            load_record(id_use);
            p_mask = fetch_mask(hi - BATCH);
        }
        if (hi - cnt >= wmax) {   // this wave's pixels all stopped in front of this batch
            HSR_SETTLE_STAGING();
            continue;
        }

        const int total = build_flat_list(wv, lane, s_list, s_lcnt, s_flat);
        if (total == 0) HSR_SETTLE_STAGING();
        for (int c0 = 0; c0 < total; c0 += SB_SLOTS) {
            const int nrows = min(SB_SLOTS, total - c0);
            // lane (group gq, row l16): does chunk entry l16 touch sub-block (wv, gq)?
            const int jr = l16 < nrows ? (int)s_flat[wv][c0 + l16] : 0;
            const bool touch = l16 < nrows && ((s_mask[jr] >> (4 * wv + gq)) & 1u);
            const uint64_t ball = __ballot(touch);
            if (gq == 0) {
                s_cj[wv][l16] = (uint8_t)jr;
                s_cid[wv][l16] = (uint32_t)s_id[jr] * (uint32_t)a.grow_stride;   // once per chunk row, not once per emitted register
            }
            clear_chunk();
            // wave-uniform (readfirstlane: the loop counter then lives in a scalar register, not in a VALU down-counter)
            const int iters = __builtin_amdgcn_readfirstlane(max(max(__popc((uint32_t)ball & 0xFFFFu), __popc((uint32_t)(ball >> 16) & 0xFFFFu)),
                                                                 max(__popc((uint32_t)(ball >> 32) & 0xFFFFu), __popc((uint32_t)(ball >> 48)))));
            uint32_t todo = (uint32_t)(ball >> (16 * gq)) & 0xFFFFu;   // this group's entries, visited in list order
            int r_next = todo ? __builtin_ctz(todo) : 0;
            int j_next = s_cj[wv][r_next];
            for (int it = 0; it < iters; it++) {
                const bool valid = todo != 0u;
                const int r = r_next, j = j_next;
                todo &= todo - 1u;
                r_next = todo ? __builtin_ctz(todo) : 0;
                j_next = s_cj[wv][r_next];
                const float4* ent = &s_ent[3 * j];
                const float4 g = ent[0];
                const float4 cd = ent[1];
                const float4 co4 = ent[2];
                const float2 co = make_float2(co4.x, co4.y);
                const float hB = co4.z;   // B' / 2
                asm volatile("" ::"v"(cd.x), "v"(cd.y), "v"(cd.z), "v"(cd.w));
                const float dx = g.x - pfx, dy = g.y - pfy;
                const float dxx = dx * dx, dxy = dx * dy, dyy = dy * dy;
                const float power2 = fmaf(co.x, dyy, fmaf(g.w, dxy, g.z * dxx));
                const float G = __builtin_amdgcn_exp2f(power2);
                const float alpha = fminf(0.99f, co.y * G);
                const bool active = valid && j >= j_first && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                if (__ballot(active) == 0ull) continue;

                const float inv_one_m_a = __builtin_amdgcn_rcpf(1.0f - alpha);
                const float test_T = T * inv_one_m_a;
                const float w = active ? alpha * test_T : 0.f;

                const float h = fmaf(cd.x, dpx0, fmaf(cd.y, dpx1, fmaf(cd.z, dpx2, fmaf(cd.w, dpd, dpo))));
                const float Rn = Racc;
                float dL_dalpha = (h - Rn) * test_T;
                dL_dalpha += (-T_final * inv_one_m_a) * bg_dot;
                const float Gs = active ? G : 0.f;
                const float gda = Gs * dL_dalpha;
                const float q = co.y * gda;
                float v[SB_NV];   // raw moments of q; the splat's coefficients are applied at emission (see render_bwd_sub_kernel)
                // dL_dmean2D: the two terms combined PER PIXEL, as the reference does (backward.cu:887-888).  Round 2 summed the raw moments
                // q dx and q dy and combined them at emission; for elongated splats A' dx and B' dy / 2 largely cancel, and cancelling AFTER
                // the fp32 sums over the pixels cost up to 15x the error in these two sums (found with the truth build of the oracle:
                // tests/test_gpu_truth.py) — which the per-Gaussian chain then amplifies into dL_dscales / dL_dmeans3D.
                v[0] = q * fmaf(g.z, dx, hB * dy);    // (A' dx + B' dy / 2): x kx * 2 at emission
                v[1] = q * fmaf(co.x, dy, hB * dx);   // (C' dy + B' dx / 2)
                v[2] = q * dxx;
                v[3] = q * dxy;
                v[4] = q * dyy;
                v[5] = gda;
                v[6] = fmaf(w, dpd, (active && j == j_median) ? dpm : 0.f);   // depth: direct sum + median term
                if (active) {
                    Racc = fmaf(alpha, h - Rn, Rn);   // evaluated now instead of at the next visit: one select instead of three
                    T = test_T;
                }
                const float total7 = row_reduce_transpose7(v, lane);
                if (myv_on && valid) u7[r * 32 + myv * 4 + gq] = total7;
            }
            if (c0 == 0) HSR_SETTLE_STAGING();
            flush(nrows);
        }
    }
}

// Wide trees (K > 27), in channel passes like experiments/hsr_render_bwd_wide.hip: semantic channels [c0, c0 + ns) of the image; the BASE
// pass adds the five direct sums and the seven butterfly values, a SEM pass only re-derives alpha and T and feeds the panel.
// 16 * NG >= ns + (BASE ? 5 : 0).
// BF: the panel contraction runs on the bf16 matrix cores (split3_bf16): 24 registers of B operand per 16 columns instead of 16.
template <int NG, bool BASE, int BATCH, bool BF>
__global__ void __launch_bounds__(256, BF ? 2 : (NG <= 2 ? 4 : (NG <= 4 ? 3 : 2))) render_bwd_subw_kernel(RenderBwdArgs a, int c0, int ns)
{
    static_assert(NG <= 7, "at most 112 channels per pass (the B operand lives in 16 * NG registers)");
    static_assert(!BF || NG <= 5, "the split B operand of more than 80 columns does not fit the register file at two waves per SIMD");
    constexpr int STRIDE = BF ? 68 : SB_STRIDE;   // floats per panel row; 68: rows 16-byte aligned, b128 reads of 16 rows hit 64 banks
    static_assert(SB_SLOTS * STRIDE <= SB_PANEL, "panel");
    static_assert(BATCH <= 256, "batch slots are bytes");
    // one 48-byte record per staged splat { x, y, A', B' | r, g, b, depth | C', opacity, -, - }: ONE address computation per visit
    __shared__ float4 s_ent[3 * BATCH];
    __shared__ int s_id[BATCH];
    __shared__ uint16_t s_mask[BATCH];                  // sub-block mask of each staged splat
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ uint8_t s_flat[4][256];
    __shared__ int s_wmax[4];
    __shared__ float s_panel[4][SB_PANEL];
    __shared__ uint8_t s_cj[4][SB_SLOTS];               // batch slot of each chunk row
    __shared__ __attribute__((aligned(16))) uint32_t s_cid[4][SB_SLOTS];   // packed-row offset (Gaussian id x row stride) of each chunk row
    __shared__ __attribute__((aligned(16))) float s_u7[4][SB_SLOTS * 8 * 4];   // [row][value 0..7][group]: butterfly sums; the four groups of a value are ONE 16-byte read at emission

    const int tile = hsr_block_tile(blockIdx.x, ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y));
    if (tile >= ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, gq = lane >> 4, l16 = lane & 15;
    const TileGeom tg = tile_geom_sub(tile, a.W, a.H, t);
    const bool inside = tg.inside;
    const size_t N = (size_t)a.W * a.H;
    const size_t pix_id = (size_t)a.W * tg.py + tg.px;
    const float pfx = tg.pfx, pfy = tg.pfy;
    const float tile_x0 = (float)(tg.tx * HSR_TILE_X), tile_y0 = (float)(tg.ty * HSR_TILE_Y);
    const uint2 range = a.ranges[tile];
    float* panel = s_panel[wv];
    float* u7 = s_u7[wv];

    // every prologue load unconditional and issued before anything consumes one (see experiments/hsr_render_bwd_mfma.hip)
    const size_t pix_ld = inside ? pix_id : 0;
    const float inm = inside ? 1.f : 0.f;
    const float T_final_ld = a.final_T[pix_ld];
    const int last_contributor_ld = (int)a.n_contrib[pix_ld];
    const int median_at_ld = (int)a.median_pos[pix_ld];
    float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f, dpd = 0.f, dpm = 0.f, dpo = 0.f;
    if (BASE) {
        dpx0 = a.dL_dpix[pix_ld] * inm; dpx1 = a.dL_dpix[N + pix_ld] * inm; dpx2 = a.dL_dpix[2 * N + pix_ld] * inm;
        dpd = a.dL_dpix_depth[pix_ld] * inm; dpm = a.dL_dpix_median[pix_ld] * inm; dpo = a.dL_dpix_opacity[pix_ld] * inm;
    }
    const float T_final = T_final_ld * inm;
    float T = T_final;
    const int last_contributor = inside ? last_contributor_ld : 0;
    const int median_at = (inside ? median_at_ld : 0) - 1;   // list position of the forward's T = 0.5 crossing (-1: none): gets dL_dmedian_depth

    int wmax = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    if (lane == 0) s_wmax[wv] = wmax;

    // ---- the MFMA B operand: G transposed through LDS (lane l holds G[pixel lane 4m + (l>>4)][channel 16g + (l&15)]) ----
    float Breg[BF ? 1 : NG][16];
    u32x4 Bh[BF ? NG : 1][2], Bm[BF ? NG : 1][2], Bl[BF ? NG : 1][2];   // BF: pixels 32 s + 8 (l>>4) + e, e = 0..7, of channel 16 g + (l&15)
#pragma unroll
    for (int g = 0; g < NG; g++) {
        float gv[16];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int ch = 16 * g + c;
            const float sv = a.dL_dpix_sem[(size_t)min(c0 + ch, a.K - 1) * N + pix_ld] * inm;
            float v = ch < ns ? sv : 0.f;
            if (BASE) {
                v = ch == ns ? dpx0 : v;
                v = ch == ns + 1 ? dpx1 : v;
                v = ch == ns + 2 ? dpx2 : v;
                v = ch == ns + 3 ? dpd : v;
                v = ch == ns + 4 ? dpo : v;
            }
            gv[c] = v;
        }
#pragma unroll
        for (int c = 0; c < 16; c++) panel[lane * 17 + c] = gv[c];
        wave_lds_fence();   // wave-private panel: see render_bwd_sub_kernel
        if (BF) {
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; e++) x[e] = panel[(32 * s2 + 8 * (lane >> 4) + e) * 17 + (lane & 15)];
                split3_bf16(x, Bh[g][s2], Bm[g][s2], Bl[g][s2]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++) Breg[g][m] = panel[(4 * m + (lane >> 4)) * 17 + (lane & 15)];
        }
        wave_lds_fence();
    }
    __syncthreads();   // s_wmax
    const int hi_all = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    const float bg_dot = a.bg[0] * dpx0 + a.bg[1] * dpx1 + a.bg[2] * dpx2;
    const float kx = (0.5f * a.W) / HSR_LOG2E, ky = (0.5f * a.H) / HSR_LOG2E;
    float Racc = 0.f;   // the reference's accum_rec AFTER the last accepted splat: last_alpha * last_h + (1 - last_alpha) * accum_rec (backward.cu:630-640, h = colour . dL_dpixel)

    // butterfly value this lane holds after row_reduce_transpose7, or -1
    const int myv = (lane & 2) ? -1 : (((lane >> 2) & 3) | ((lane & 1) << 2));
    const bool myv_on = myv >= 0 && myv < SB_NV;
    // packed-row columns of the accumulator columns this lane holds (col = lane & 15 of channel group g), or -1
    int colg[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int ch = 16 * g + l16;
        colg[g] = ch < ns ? HSR_GROW_SEM0 + c0 + ch : ((BASE && ch < ns + 5) ? hsr_grow_direct0(a.K) + (ch - ns) : -1);
    }

    // zeroes the panel rows and the butterfly slots of the next chunk (wave-private LDS: no barrier)
    auto clear_chunk = [&]() {
        float2* p2 = reinterpret_cast<float2*>(panel);
#pragma unroll
        for (int i = 0; i < (SB_SLOTS * STRIDE) / 2; i += 64)
            if (i + lane < (SB_SLOTS * STRIDE) / 2) p2[i + lane] = make_float2(0.f, 0.f);
        if (BASE) {
            float4* u4 = reinterpret_cast<float4*>(u7);
#pragma unroll
            for (int i = 0; i < (SB_SLOTS * 4 * 8) / 4; i += 64) u4[i + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // the chunk's 16 panel rows -> D[16 entries][16*NG channels] -> packed rows; butterfly slots -> columns 0..6
    auto flush = [&](int nrows) {
        f32x4 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (BF) {
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                // row l16, pixels 32 s + 8 (l>>4) .. + 7: two aligned b128 reads
                const float4* ap = reinterpret_cast<const float4*>(panel + l16 * STRIDE + 32 * s2 + 8 * (lane >> 4));
                const float4 x0 = ap[0], x1 = ap[1];
                const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                u32x4 Ah, Am, Al;
                split3_bf16(x, Ah, Am, Al);
                // small products first
#pragma unroll
                for (int g = 0; g < NG; g++) acc[g] = mma_bf16(Al, Bh[g][s2], acc[g]);
#pragma unroll
                for (int g = 0; g < NG; g++) acc[g] = mma_bf16(Ah, Bl[g][s2], acc[g]);
#pragma unroll
                for (int g = 0; g < NG; g++) acc[g] = mma_bf16(Am, Bm[g][s2], acc[g]);
#pragma unroll
                for (int g = 0; g < NG; g++) acc[g] = mma_bf16(Am, Bh[g][s2], acc[g]);
#pragma unroll
                for (int g = 0; g < NG; g++) acc[g] = mma_bf16(Ah, Bm[g][s2], acc[g]);
#pragma unroll
                for (int g = 0; g < NG; g++) acc[g] = mma_bf16(Ah, Bh[g][s2], acc[g]);
            }
        } else {
            const float* arow = panel + l16 * STRIDE + (lane >> 4);
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const float av = arow[4 * m];
#pragma unroll
                for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Breg[g][m], acc[g], 0, 0, 0);
            }
        }
        // D[row = 4*(lane>>4) + r][col = lane&15]: one atomic wave-instruction per register = 4 rows x 64 bytes
        {
            const uint4 b4 = *reinterpret_cast<const uint4*>(&s_cid[wv][4 * (lane >> 4)]);   // the four row offsets in one LDS read
            const uint32_t bb[4] = {b4.x, b4.y, b4.z, b4.w};
            const int nr = nrows - 4 * (lane >> 4);   // how many of this lane's four rows exist
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int g = 0; g < NG; g++)
                    if (r < nr && colg[g] >= 0 && acc[g][r] != 0.f && !(a.debug_flags & 1))
                        atomicAdd(a.grow + (bb[r] + (uint32_t)colg[g]), acc[g][r]);
        }
        // butterfly columns 0..6: two wave-instructions of 8 rows x 7 values, so that each row's line is ONE request
#pragma unroll
        for (int pass = 0; pass < (BASE ? 2 : 0); pass++) {
            const int row = (lane >> 3) + 8 * pass, vi = lane & 7;
            const float4 sa = *reinterpret_cast<const float4*>(u7 + row * 32 + vi * 4);
            const float ta = (sa.x + sa.y) + (sa.z + sa.w);
            const float wa = vi == 0 ? 2.0f * kx : (vi == 1 ? 2.0f * ky : (vi <= 4 ? -0.5f : 1.0f));
            const float val = wa * ta;
            const uint32_t base = s_cid[wv][row] + (uint32_t)vi;
            if (vi < SB_NV && row < nrows && val != 0.f && !(a.debug_flags & 1)) atomicAdd(a.grow + base, val);
        }
    };

    // ---- software-pipelined staging ----
    int id_next = 0, id_cur = 0;
    float2 p_xy = {0, 0};
    float4 p_co = {0, 0, 0, 0};
    float p_r = 0, p_g = 0, p_b = 0, p_d = 0;
    uint32_t p_mask = 0u;
    // unconditional, clamped staging loads, the id of the batch after next requested before the next batch's records: see
    // render_fwd_kernel (a load inside a divergent `if`, or into a register the loads before it took their addresses from, is waited
    // for where it is issued — and the (rec == NULL) fallback kept three of these values in a scratch slot).  a.rec is never NULL.
    const int n_list = (int)(range.y - range.x);
    auto fetch_id = [&](int hi) -> int { return (int)a.point_list[range.x + min(max(hi - 1 - t, 0), max(n_list - 1, 0))]; };
    auto fetch_mask = [&](int hi) -> uint32_t { return SAVED_MASKS ? a.masks[range.x + min(max(hi - 1 - t, 0), max(n_list - 1, 0))] : 0u; };
    auto load_record = [&](int id_of) {
        const size_t id = (size_t)id_of;
        id_cur = id_of;
        const float4* rec = a.rec + 4 * id;
        const float4 r0 = rec[0], r2 = rec[2];
        p_co = rec[1];
        p_xy = make_float2(r0.x, r0.y);
        p_d = r0.z;
        p_r = r2.x; p_g = r2.y; p_b = r2.z;
    };
    if (n_list > 0) {
        const int id0 = fetch_id(hi_all);
        id_next = fetch_id(hi_all - BATCH);
        load_record(id0);
        p_mask = fetch_mask(hi_all);
    }

    for (int hi = hi_all; hi > 0; hi -= BATCH) {
        const int cnt = min(BATCH, hi);
        // list position of batch slot j is hi - 1 - j: "behind the last contributor" and "the median splat" as slot tests, per batch
        const int j_first = hi - last_contributor, j_median = hi - 1 - median_at;
        __syncthreads();
        uint32_t qmask = 0u;
        if (t < cnt) {
            const uint32_t mask = SAVED_MASKS ? p_mask
                                              : subblock_mask(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0, !(a.debug_flags & 16));
            qmask = (uint32_t)((mask & 0xFu) != 0u) | ((uint32_t)((mask & 0xF0u) != 0u) << 1) | ((uint32_t)((mask & 0xF00u) != 0u) << 2) |
                    ((uint32_t)((mask & 0xF000u) != 0u) << 3);
            s_mask[t] = (uint16_t)mask;
            s_id[t] = id_cur;
            s_ent[3 * t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_ent[3 * t + 1] = make_float4(p_r, p_g, p_b, p_d);
            s_ent[3 * t + 2] = make_float4((-0.5f * HSR_LOG2E) * p_co.z, p_co.w, (-0.5f * HSR_LOG2E) * p_co.y, 0.f);   // C', opacity, B' / 2
        }
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        __syncthreads();
        {
            const int id_use = id_next;            // ids of the next batch, requested a whole batch ago
            id_next = fetch_id(hi - 2 * BATCH);
            load_record(id_use);
            p_mask = fetch_mask(hi - BATCH);
        }
        if (hi - cnt >= wmax) {   // this wave's pixels all stopped in front of this batch
            HSR_SETTLE_STAGING();
            continue;
        }

        const int total = build_flat_list(wv, lane, s_list, s_lcnt, s_flat);
        if (total == 0) HSR_SETTLE_STAGING();
        for (int c0 = 0; c0 < total; c0 += SB_SLOTS) {
            const int nrows = min(SB_SLOTS, total - c0);
            // lane (group gq, row l16): does chunk entry l16 touch sub-block (wv, gq)?
            const int jr = l16 < nrows ? (int)s_flat[wv][c0 + l16] : 0;
            const bool touch = l16 < nrows && ((s_mask[jr] >> (4 * wv + gq)) & 1u);
            const uint64_t ball = __ballot(touch);
            if (gq == 0) {
                s_cj[wv][l16] = (uint8_t)jr;
                s_cid[wv][l16] = (uint32_t)s_id[jr] * (uint32_t)a.grow_stride;   // once per chunk row, not once per emitted register
            }
            clear_chunk();
            // wave-uniform (readfirstlane: the loop counter then lives in a scalar register, not in a VALU down-counter)
            const int iters = __builtin_amdgcn_readfirstlane(max(max(__popc((uint32_t)ball & 0xFFFFu), __popc((uint32_t)(ball >> 16) & 0xFFFFu)),
                                                                 max(__popc((uint32_t)(ball >> 32) & 0xFFFFu), __popc((uint32_t)(ball >> 48)))));
            uint32_t todo = (uint32_t)(ball >> (16 * gq)) & 0xFFFFu;   // this group's entries, visited in list order
            int r_next = todo ? __builtin_ctz(todo) : 0;
            int j_next = s_cj[wv][r_next];
            for (int it = 0; it < iters; it++) {
                const bool valid = todo != 0u;
                const int r = r_next, j = j_next;
                todo &= todo - 1u;
                r_next = todo ? __builtin_ctz(todo) : 0;
                j_next = s_cj[wv][r_next];
                const float4* ent = &s_ent[3 * j];
                const float4 g = ent[0];
                const float4 cd = ent[1];
                const float4 co4 = ent[2];
                const float2 co = make_float2(co4.x, co4.y);
                const float hB = co4.z;   // B' / 2
                asm volatile("" ::"v"(cd.x), "v"(cd.y), "v"(cd.z), "v"(cd.w));
                const float dx = g.x - pfx, dy = g.y - pfy;
                const float dxx = dx * dx, dxy = dx * dy, dyy = dy * dy;
                const float power2 = fmaf(co.x, dyy, fmaf(g.w, dxy, g.z * dxx));
                const float G = __builtin_amdgcn_exp2f(power2);
                const float alpha = fminf(0.99f, co.y * G);
                const bool active = valid && j >= j_first && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                if (__ballot(active) == 0ull) continue;

                const float inv_one_m_a = __builtin_amdgcn_rcpf(1.0f - alpha);
                const float test_T = T * inv_one_m_a;
                const float w = active ? alpha * test_T : 0.f;
                if (valid) panel[r * STRIDE + lane] = w;

                if (!BASE) {
                    if (active) T = test_T;
                    continue;
                }
                const float h = fmaf(cd.x, dpx0, fmaf(cd.y, dpx1, fmaf(cd.z, dpx2, fmaf(cd.w, dpd, dpo))));
                const float Rn = Racc;
                float dL_dalpha = (h - Rn) * test_T;
                dL_dalpha += (-T_final * inv_one_m_a) * bg_dot;
                const float Gs = active ? G : 0.f;
                const float gda = Gs * dL_dalpha;
                const float q = co.y * gda;
                float v[SB_NV];   // raw moments of q; the splat's coefficients are applied at emission (see render_bwd_sub_kernel)
                // dL_dmean2D: the two terms combined PER PIXEL, as the reference does (backward.cu:887-888).  Round 2 summed the raw moments
                // q dx and q dy and combined them at emission; for elongated splats A' dx and B' dy / 2 largely cancel, and cancelling AFTER
                // the fp32 sums over the pixels cost up to 15x the error in these two sums (found with the truth build of the oracle:
                // tests/test_gpu_truth.py) — which the per-Gaussian chain then amplifies into dL_dscales / dL_dmeans3D.
                v[0] = q * fmaf(g.z, dx, hB * dy);    // (A' dx + B' dy / 2): x kx * 2 at emission
                v[1] = q * fmaf(co.x, dy, hB * dx);   // (C' dy + B' dx / 2)
                v[2] = q * dxx;
                v[3] = q * dxy;
                v[4] = q * dyy;
                v[5] = gda;
                v[6] = (active && j == j_median) ? dpm : 0.f;
                if (active) {
                    Racc = fmaf(alpha, h - Rn, Rn);   // evaluated now instead of at the next visit: one select instead of three
                    T = test_T;
                }
                const float total7 = row_reduce_transpose7(v, lane);
                if (myv_on && valid) u7[r * 32 + myv * 4 + gq] = total7;
            }
            if (c0 == 0) HSR_SETTLE_STAGING();
            flush(nrows);
        }
    }
}

}  // namespace

// packed mode, K <= 27, P * grow_stride < 2^30 (32-bit row addressing): the caller checks
int hsr_launch_render_backward_sub(const RenderBwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles)), block(256);
    const int K = a.semantic ? a.K : 0;
    if (K == 0) render_bwd_sub_kernel<0, 224><<<grid, block, 0, stream>>>(a);
    else if (K <= 11) render_bwd_sub_kernel<11, 224><<<grid, block, 0, stream>>>(a);
    else if (K == 16) render_bwd_sub_kernel<16, 224><<<grid, block, 0, stream>>>(a);
    else if (K == 26) render_bwd_sub_kernel<26, 224><<<grid, block, 0, stream>>>(a);
    else render_bwd_sub_kernel<27, 224><<<grid, block, 0, stream>>>(a);
    return HSR_OK;
}

namespace {
template <bool BASE>
void launch_subw_pass(const RenderBwdArgs& a, int c0, int ns, dim3 grid, hipStream_t stream)
{
    const int groups = (ns + (BASE ? 5 : 0) + 15) / 16;
    const dim3 block(256);
    // bf16 matrix cores on the exact three-way split where they paid (tools/wide_mma_ab.sh, 500k Gaussians, bwd_render ms, fp32 -> split):
    // 4 column groups 0.495 -> 0.486, 5 groups 0.606 -> 0.573 (1920x1080, 2M: 2.168 -> 2.058); 3 groups lose (0.407 -> 0.419: 24 B
    // registers per group push the kernel from 3 waves per SIMD to 2); 2 groups (K <= 27) tie even at 3 waves; 6 and 7 groups do not fit.
#ifdef HSR_ABLATE
    static const char* e_mma = getenv("HSR_BWD_WIDE_MMA");   // A/B selector, ablate build only (parity-tested there): "f32" = fp32 matrix instructions
    const bool bf = !(e_mma && !strcmp(e_mma, "f32"));
#else
    constexpr bool bf = true;
#endif
    if (groups <= 1) render_bwd_subw_kernel<1, BASE, 224, false><<<grid, block, 0, stream>>>(a, c0, ns);
    else if (groups == 2) render_bwd_subw_kernel<2, BASE, 224, false><<<grid, block, 0, stream>>>(a, c0, ns);
    else if (groups == 3) render_bwd_subw_kernel<3, BASE, 224, false><<<grid, block, 0, stream>>>(a, c0, ns);
    else if (groups == 4 && bf) render_bwd_subw_kernel<4, BASE, 224, true><<<grid, block, 0, stream>>>(a, c0, ns);
    else if (groups == 5 && bf) render_bwd_subw_kernel<5, BASE, 224, true><<<grid, block, 0, stream>>>(a, c0, ns);
#ifdef HSR_ABLATE
    else if (groups == 4) render_bwd_subw_kernel<4, BASE, 224, false><<<grid, block, 0, stream>>>(a, c0, ns);
    else if (groups == 5) render_bwd_subw_kernel<5, BASE, 224, false><<<grid, block, 0, stream>>>(a, c0, ns);
#endif
    else if (groups == 6) render_bwd_subw_kernel<6, BASE, 224, false><<<grid, block, 0, stream>>>(a, c0, ns);
    else render_bwd_subw_kernel<7, BASE, 224, false><<<grid, block, 0, stream>>>(a, c0, ns);
}
}  // namespace

// geometry-only gradients (a.grow_stride == 16): any K
int hsr_launch_render_backward_geo(const RenderBwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    render_bwd_geo_kernel<224><<<dim3(hsr_tile_grid(tiles)), dim3(256), 0, stream>>>(a);
    return HSR_OK;
}

// semantic variant with K > 27, packed mode, P * grow_stride < 2^30: BASE pass (59 channels + the base sums) + SEM passes of 64
int hsr_launch_render_backward_subw(const RenderBwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles));
    const int K = a.K;
    // One pass while the B operand (16 registers per 16 columns) fits two waves per SIMD: K + 5 <= 112 columns.  Every pass
    // re-derives alpha and T for every (pixel, splat) pair, and that — not the matrix-core work, which is the same in total —
    // is most of a pass: K = 74 in ONE pass of 80 columns at 2 waves per SIMD instead of 64 + 22 columns at 3 and 4.
    static const char* e_pass = getenv("HSR_BWD_WIDE_PASS");   // kernel-family selector (parity-tested): "split" = 64-column passes
    const bool split = e_pass && !strcmp(e_pass, "split");
    // (Other splits were measured too, tools/wide_split_sweep.sh on 500k Gaussians: every extra pass costs ~0.3-0.5 ms whatever its
    // width — K = 74: one pass 0.62 ms, 27 + 47 channels 0.89 ms, 43 + 31 0.82 ms; K = 102: 0.79 vs 1.16-1.34 ms.)
    const int first = split ? (K < 59 ? K : 59) : (K < 107 ? K : 107);
    const int chunk = 64;
    launch_subw_pass<true>(a, 0, first, grid, stream);
    for (int c0 = first; c0 < K; c0 += chunk) launch_subw_pass<false>(a, c0, K - c0 < chunk ? K - c0 : chunk, grid, stream);
    return HSR_OK;
}
