// hsr_render_fwd_mma.hip — semantic tile forward whose channel accumulation runs on the matrix cores (gfx950, round 3).
// EXPERIMENT, ablate build only (HSR_FWD_IMPL=mma): parity-green on every case of tools/fwd_mma_ab.sh, and slower than the per-lane forward
// at every width (500k Gaussians, fwd_render ms, per-lane -> this: K = 16 0.136 -> 0.166, K = 26 0.171 -> 0.186, K = 48 0.274 -> 0.381,
// K = 74 0.356 -> 0.561, K = 102 0.495 -> 0.711; profiles/r03_fwd_mma_ab.log).  Why, measured (tools/micro/valu_rate.hip,
// profiles/r03_valu_rate_dpp.jsonl): v_mfma_f32_16x16x4_f32 retires 1024 MACs in 24-32 SIMD cycles = 32-43 MAC per cycle, a plain
// v_fmac_f32 64 MACs in 1.9 = 34, v_pk_fma_f32 128 in 3.56 = 36 — the fp32 matrix pipe is no faster than the vector pipe, and matrix and
// vector instructions of different waves do not overlap on a SIMD (4 MFMA + 32 FMA: 201 cycles against 96 + 69 alone).  Only against the
// DPP FMAs of the wide per-lane kernel (3.5 cycles, 18 MAC per cycle) is there a factor to win, and the 4 NCG operand registers + 24 record
// registers of a round push those widths to two waves per SIMD or into spills.  Kept for its transposition scheme and the measurement.
//
// Same per-pixel semantics as hsr_render_fwd.hip (reference renderCUDA_SEM, cuda_rasterizer/forward.cu:400-538): front-to-back over
// the tile's depth-sorted list, power > 0 and alpha < 1/255 skipped, alpha clamped at 0.99, pixel terminated when T (1 - alpha) < 1e-4,
// median depth = depth of the splat where T crosses 0.5 (default 15), no background blend.  Same staging pipeline, same 4x4 sub-block
// lists (hsr_tile_common.h), same outputs and saved state (final_T, n_contrib, median_pos, sub-block masks).
//
// What is different: WHO multiplies weights and features.  In hsr_render_fwd.hip every lane owns a pixel and its K + 4 accumulators;
// with per-lane rows the LDS crossbar is the bound (each of a group's 16 lanes reads the same row), with quad-shared rows the feature
// reaches the FMA through its DPP operand — and a DPP FMA issues at 3.5 SIMD cycles per wave instruction (5.25 at two waves per SIMD)
// against 1.9 for a plain one (tools/micro/valu_rate.hip, profiles/r03_valu_rate_dpp.jsonl): at K = 74, 76 of the 110 vector
// instructions of a visit are DPP FMAs — 46 % of the kernel's issue cycles.
// Here the lanes only evaluate alpha and the transmittance chain (the ~30 instructions per visit that are inherently per pixel); every
// weighted sum — the K features, blue, depth, red, green — is the small matrix product
//       D_g[16 pixels][16 channels] += W_g[16 pixels][4 splats] . F_g[4 splats][16 channels]
// per 16-lane group g (one 4x4 sub-block, walking ITS list) and round of four list entries: one v_mfma_f32_16x16x4_f32 per
// (group, 16 channels, round) — an exact fp32 fma chain over the four splats in list order, i.e. the same sums in the same order.
//   * A operand (lane i + 16 k = weight of pixel i for the round's k-th splat): each lane has computed its pixel's four weights into four
//     registers; a 4 x 4 transpose of (16-lane row, register) — two v_permlane32_swap + two v_permlane16_swap — turns them into the four
//     groups' A operands.
//   * B operand (lane n + 16 k = channel n of the k-th splat's row): ONE 4-byte LDS read per (group, 16 channels, round); every staged
//     feature is read once per group that visits it — 4x fewer LDS bytes than quad-shared rows, 16x fewer than per-lane rows.
//   * D (lane n + 16 y, register x = pixel (x, y) of the sub-block, channel n): 4 (K + 4) / 16 accumulator registers per lane, as many as
//     the per-lane kernel's, but nothing else is parked in registers (no feature words): three waves per SIMD at K = 102 instead of two.
//   * The staged row is { s_0 .. s_(K-1), b, depth, r, g, 0 .. } padded to whole 16-channel groups; the epilogue stores the D registers
//     as 16-byte pieces (four x-adjacent pixels of one channel plane), the same granularity as the per-lane kernel's stores.
// K is a run-time argument; the kernel is instantiated per number of 16-channel groups (K + 4 <= 16 NCG).
#include <stdlib.h>
#include <string.h>

#include "hsr_tile_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

// (a, b) -> (rows of a and b interleaved by HALF: a' = [a.lo, b.lo], b' = [a.hi, b.hi])
__device__ __forceinline__ void swap32(float& a, float& b)
{
    const uint2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// (a, b) -> (a' = [a.r0, b.r0, a.r2, b.r2], b' = [a.r1, b.r1, a.r3, b.r3]), r = 16-lane row
__device__ __forceinline__ void swap16(float& a, float& b)
{
    const uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

constexpr int mma_occupancy(int ncg) { return ncg <= 2 ? 4 : (ncg <= 3 ? 3 : 2); }
constexpr int mma_batch(int ncg) { return ncg <= 2 ? 224 : (ncg <= 4 ? 160 : (ncg <= 5 ? 128 : (ncg == 6 ? 112 : (ncg == 7 ? 144 : 112)))); }

template <int NCG, bool ALIGNED>
__global__ void __launch_bounds__(256, mma_occupancy(NCG)) render_fwd_mma_kernel(RenderFwdArgs a)
{
    constexpr int RWP = 16 * NCG;              // floats per staged row: K features, blue, depth, red, green, zeros
    constexpr int BATCH = mma_batch(NCG);      // list entries staged at a time (slots are bytes)
    static_assert(BATCH <= 256, "batch slots are bytes");
    constexpr int NPF = (RWP - 4 + 15) / 16 + 1;   // touches per row: every 16th float and the last one
    __shared__ float4 s_geo[BATCH];            // x, y, A', B'   (pre-scaled conic, hsr_tile_common.h)
    __shared__ float2 s_co[BATCH];             // C', opacity
    __shared__ float4 s_row[BATCH * (RWP / 4)];
    __shared__ __attribute__((aligned(16))) uint8_t s_sublist[16 * HSR_SUB_LSTRIDE];
    __shared__ uint8_t s_subcnt[4][16];
    __shared__ int s_wdone[4];

    const int tiles_all = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const int tile = hsr_block_tile(blockIdx.x, tiles_all);
    if (tile >= tiles_all) return;
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63, grp = lane >> 4, l16 = lane & 15;
    const int K = a.K;
    if (a.bin.base) {   // speculative forward: the list lives where num_rendered says
        BinState bs;
        if (!hsr_bin_resolve(a.bin, *a.bin.R_dev, &bs)) {
            hsr_poison_tile(a, tile, t, true, 0, K);   // the binning buffer cannot hold num_rendered (hsr_tile_common.h)
            return;
        }
        a.point_list = bs.vals;
        a.masks = bs.vals_unsorted;
    }
    const TileGeom tg = tile_geom_sub(tile, a.W, a.H, t);
    const bool inside = tg.inside;
    float pfx = tg.pfx, pfy = tg.pfy;
    asm volatile("" : "+v"(pfx), "+v"(pfy));
    const float tile_x0 = (float)(tg.tx * HSR_TILE_X), tile_y0 = (float)(tg.ty * HSR_TILE_Y);
    const size_t N = (size_t)a.W * a.H;
    const uint2 range = a.ranges[tile];
    const int n = (int)(range.y - range.x);

    // ---- per-pixel state (this lane's pixel) ----
    float T = 1.0f;
    uint32_t last_contributor = 0;
    uint32_t median_at = 0;   // 1 + list position of the splat at which T crossed 0.5 (ImgState::median_pos)
    float median_D = 15.0f;   // its depth (forward.cu:511-515; default 15.0, :450)
    bool done = !inside;
    // ---- accumulators: D[g][c][x] = pixel (x, lane >> 4) of this wave's sub-block g, channel 16 c + (lane & 15) ----
    f32x4 D[4][NCG];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int c = 0; c < NCG; c++) D[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- software-pipelined staging (lane t <-> splat t of a batch): ids two batches ahead, the 64-byte record one batch ahead,
    // the semantic row TOUCHED one batch ahead (one word per 64-byte line: the lines are in this XCD's L2 when the row is read for
    // real at staging time) — hsr_render_fwd.hip, PF ----
    int id_next = 0, id_cur = 0;
    float2 p_xy = {0, 0};
    float4 p_co = {0, 0, 0, 0};
    float p_r = 0, p_g = 0, p_b = 0, p_d = 0;
    float pf_t[NPF];
#pragma unroll
    for (int c = 0; c < NPF; c++) pf_t[c] = 0.f;
    float pf_sink = 0.f;
    // unconditional, clamped loads: a load inside a divergent `if` is waited for where it is issued (EXPERIMENTS.md §4e)
    auto fetch_id = [&](int start) -> int { return (int)a.point_list[range.x + min(max(start + t, 0), max(n - 1, 0))]; };
    auto load_record = [&](int id_of) {
        const size_t id = (size_t)id_of;
        id_cur = id_of;
        const float4* rec = a.rec + 4 * id;
        const float4 r0 = rec[0], r2 = rec[2];
        p_co = rec[1];
        p_xy = make_float2(r0.x, r0.y);
        p_d = r0.z;
        p_r = r2.x; p_g = r2.y; p_b = r2.z;
        if (K > 0) {
            const float* row = a.semantics + id * (size_t)K;
#pragma unroll
            for (int c = 0; c < NPF; c++) pf_t[c] = row[min(16 * c, K - 1)];
        }
    };
    if (n > 0) {
        const int id0 = fetch_id(0);
        id_next = fetch_id(BATCH);
        load_record(id0);
    }
    // the depth of the splat at which T crossed 0.5: read from its staged row (channel K + 1) right after the round in which it
    // happened, while the batch is still in LDS
    const float* rowf = reinterpret_cast<const float*>(s_row);

    for (int start = 0; start < n; start += BATCH) {
        const bool wave_done = __ballot(!done) == 0ull;
        if (lane == 0) s_wdone[wv] = wave_done;
        __syncthreads();   // also: everyone has finished reading the previous batch
        if (s_wdone[0] & s_wdone[1] & s_wdone[2] & s_wdone[3]) break;
        const int cnt = min(BATCH, n - start);
        uint32_t qmask = 0u;
        if (t < cnt) {
            const uint32_t mask16 = subblock_mask(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0, !(a.debug_flags & 16));
            qmask = mask16;
            a.masks[range.x + start + t] = mask16;   // the backward stages the same entries: it reads the mask instead of deriving it again
            s_geo[t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_co[t] = make_float2((-0.5f * HSR_LOG2E) * p_co.z, p_co.w);
#pragma unroll
            for (int c = 0; c < NPF; c++) pf_sink += pf_t[c];   // the touches landed a batch ago: consume them so that they stay real loads
            // an entry no sub-block will visit (a fifth to a third of a tile's list) needs no row — except slot 0, which stands in for
            // the entries past the end of a group's list (weight 0): an unstaged row is whatever the LDS held, and NaN * 0 is NaN
            if (qmask != 0u || t == 0) {
                const float* grow = a.semantics + (size_t)id_cur * (size_t)K;
                float4* row = &s_row[t * (RWP / 4)];
#pragma unroll
                for (int g0 = 0; g0 < RWP; g0 += 32) {   // 32 floats in flight at a time
                    float rv[32];
#pragma unroll
                    for (int c = 0; c < 32; c++) {
                        const int ch = g0 + c;
                        if (ch >= RWP) continue;
                        if (ALIGNED) {   // K even: rows are 8-byte aligned
                            if ((c & 1) == 0) {
                                float2 v = make_float2(0.f, 0.f);
                                if (ch < K) v = reinterpret_cast<const float2*>(grow)[ch / 2];
                                rv[c] = v.x;
                                rv[c + 1] = v.y;
                            }
                        } else {
                            rv[c] = ch < K ? grow[ch] : 0.f;
                        }
                    }
                    // blue, depth, red, green ride behind the features
#pragma unroll
                    for (int c = 0; c < 32; c++) {
                        const int ch = g0 + c;
                        if (ch < RWP && ch >= K && ch < K + 4) rv[c] = ch == K ? p_b : (ch == K + 1 ? p_d : (ch == K + 2 ? p_r : p_g));
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (g0 / 4 + q < RWP / 4) row[g0 / 4 + q] = make_float4(rv[4 * q], rv[4 * q + 1], rv[4 * q + 2], rv[4 * q + 3]);
                    asm volatile("" ::: "memory");   // next group's loads stay behind this group's stores
                }
            }
        }
        publish_subblock_lists(qmask, t, s_sublist, s_subcnt);
        __syncthreads();
        // next batch's gathers go out now and land while this batch is blended
        {
            const int id_use = id_next;
            id_next = fetch_id(start + 2 * BATCH);
            load_record(id_use);
        }
        if (wave_done) continue;

        // ---- four lists per wave, one per 16-lane group; rounds of four entries ----
        const int sb = wv * 4 + grp;
        const int total = flatten_sublist(sb, lane, s_sublist, s_subcnt);
        const int tot0 = __builtin_amdgcn_readlane(total, 0), tot1 = __builtin_amdgcn_readlane(total, 16),
                  tot2 = __builtin_amdgcn_readlane(total, 32), tot3 = __builtin_amdgcn_readlane(total, 48);
        const int m = max(max(tot0, tot1), max(tot2, tot3));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the B side reads the other groups' lists
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint8_t* lists = s_sublist + (wv * 4) * HSR_SUB_LSTRIDE;
        const int kb = lane >> 4;   // B side: this lane's row of the operand = the round's kb-th splat
        const int tots[4] = {tot0, tot1, tot2, tot3};
        // the four groups' slot words of a round (wave-uniform addresses: LDS broadcasts), fetched one round ahead
        uint32_t lw[4];
#pragma unroll
        for (int g = 0; g < 4; g++) lw[g] = *reinterpret_cast<const uint32_t*>(lists + g * HSR_SUB_LSTRIDE);
        for (int pos = 0; pos < m; pos += 4) {
            // ---- phase 1: every LDS read of the round goes out at once — the records of this lane's four entries first (the weights
            // wait for those only: LDS returns in order), then the 4 NCG B-operand words, which land while the weights are computed ----
            const uint32_t own = grp == 0 ? lw[0] : (grp == 1 ? lw[1] : (grp == 2 ? lw[2] : lw[3]));
            float4 gk[4];
            float2 ck[4];
            int jk[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                jk[k] = (pos + k < total) ? (int)((own >> (8 * k)) & 0xFFu) : 0;
                gk[k] = s_geo[jk[k]];
                ck[k] = s_co[jk[k]];
            }
            float bop[4][NCG];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int jb = (pos + kb < tots[g]) ? (int)((lw[g] >> (8 * kb)) & 0xFFu) : 0;
                const float* brow = rowf + jb * RWP + l16;
#pragma unroll
                for (int c = 0; c < NCG; c++) bop[g][c] = brow[16 * c];
            }
#pragma unroll
            for (int g = 0; g < 4; g++) lw[g] = *reinterpret_cast<const uint32_t*>(lists + g * HSR_SUB_LSTRIDE + min(pos + 4, 252));
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 2: this lane's pixel against its group's four entries: the weights ----
            float w[4];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool valid = pos + k < total;
                const int j = jk[k];
                const float4 g = gk[k];
                const float2 co = ck[k];
                const float dx = g.x - pfx, dy = g.y - pfy;
                const float power2 = fmaf(co.x, dy * dy, fmaf(g.w, dx * dy, g.z * (dx * dx)));   // log2(G)
                const float alpha = fminf(0.99f, co.y * __builtin_amdgcn_exp2f(power2));
                bool contrib = valid && !done && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                const float test_T = T * (1.0f - alpha);
                if (contrib && test_T < 0.0001f) {
                    done = true;
                    contrib = false;
                }
                w[k] = contrib ? alpha * T : 0.f;
                if (contrib) {
                    if (T > 0.5f && test_T < 0.5f) median_at = (uint32_t)(start + j + 1);   // its depth: after the batch's rounds
                    T = test_T;
                    last_contributor = (uint32_t)(start + j + 1);
                    any = true;
                }
            }
            // ---- (row, register) transpose: w[k] of group g  ->  A operand of group g, row k ----
            swap32(w[0], w[2]);
            swap32(w[1], w[3]);
            swap16(w[0], w[1]);
            swap16(w[2], w[3]);
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 3: the products (skipped when no pixel of the wave accepted any of the round's entries) ----
            if (__ballot(any) != 0ull) {
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int c = 0; c < NCG; c++) D[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g], bop[g][c], D[g][c], 0, 0, 0);
            }
        }
        // the depth of the splat at which T crossed 0.5 (once per pixel): from its staged row, while the batch is still in LDS
        if (median_at > (uint32_t)start) median_D = rowf[((int)median_at - 1 - start) * RWP + K + 1];
    }
    if (pf_sink == 1.2345678e-30f) T = pf_sink;   // never true for data that matters; keeps the touch loads alive

    // ---- per-pixel outputs of this lane's pixel ----
    if (inside) {
        const size_t pix_id = (size_t)a.W * (size_t)(int)pfy + (size_t)(int)pfx;
        a.final_T[pix_id] = T;
        a.n_contrib[pix_id] = last_contributor;
        a.median_pos[pix_id] = median_at;
        a.out_median_depth[pix_id] = median_D;
        a.out_opacity[pix_id] = 1.0f - T;
    }
    // ---- the weighted sums: lane (n = lane & 15, y = lane >> 4) holds channel 16 c + n of the four pixels (0..3, y) of sub-block g ----
    const bool quad_store = (a.W & 3) == 0;   // then a sub-block row is 16-byte aligned in every plane and never straddles the right edge
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int px0 = tg.tx * HSR_TILE_X + (wv & 1) * 8 + (g & 1) * 4;
        const int py = tg.ty * HSR_TILE_Y + (wv >> 1) * 8 + (g >> 1) * 4 + (lane >> 4);
        if (py >= a.H || px0 >= a.W) continue;
        const size_t pix0 = (size_t)a.W * py + px0;
#pragma unroll
        for (int c = 0; c < NCG; c++) {
            const int ch = 16 * c + l16;
            float* plane = nullptr;
            if (ch < K) plane = a.out_semantic + (size_t)ch * N;
            else if (ch == K) plane = a.out_color + 2 * N;       // blue
            else if (ch == K + 1) plane = a.out_depth;
            else if (ch == K + 2) plane = a.out_color;           // red
            else if (ch == K + 3) plane = a.out_color + N;       // green
            if (!plane) continue;
            const f32x4 v = D[g][c];
            if (quad_store) {
                *reinterpret_cast<float4*>(plane + pix0) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int x = 0; x < 4; x++)
                    if (px0 + x < a.W) plane[pix0 + x] = v[x];
            }
        }
    }
}

template <int NCG>
void launch_mma(const RenderFwdArgs& a, dim3 grid, hipStream_t stream)
{
    if ((a.K & 1) == 0) render_fwd_mma_kernel<NCG, true><<<grid, dim3(256), 0, stream>>>(a);
    else render_fwd_mma_kernel<NCG, false><<<grid, dim3(256), 0, stream>>>(a);
}

}  // namespace

// semantic variant, K + 4 <= 144 channels in one pass; false: not taken (non-semantic, wider trees)
bool hsr_launch_render_forward_mma(const RenderFwdArgs& a, hipStream_t stream)
{
    if (!a.semantic || a.K < 0 || a.K + 4 > 144) return false;
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles));
    switch ((a.K + 4 + 15) / 16) {
    case 1: launch_mma<1>(a, grid, stream); break;
    case 2: launch_mma<2>(a, grid, stream); break;
    case 3: launch_mma<3>(a, grid, stream); break;
    case 4: launch_mma<4>(a, grid, stream); break;
    case 5: launch_mma<5>(a, grid, stream); break;
    case 6: launch_mma<6>(a, grid, stream); break;
    case 7: launch_mma<7>(a, grid, stream); break;
    case 8: launch_mma<8>(a, grid, stream); break;
    default: launch_mma<9>(a, grid, stream); break;
    }
    return true;
}
