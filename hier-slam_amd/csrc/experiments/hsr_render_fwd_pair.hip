// hsr_render_fwd_pair.hip — forward tile kernel, K <= 27: blend accumulation on the matrix cores, software-pipelined
// over PAIRS of list entries.
//
// Per-pixel semantics are those of hsr_render_fwd.hip (reference forward.cu:261-538).  Two things bound that kernel:
// ~52 VALU instructions per (wave, splat) of which 15 are the packed FMAs of the 30 blended channels, and two dependent
// LDS round trips per splat (list slot -> record) that only occupancy hides.  Here
//   * the 30 channels (sem[K], r, g, b, depth, mask) are accumulated by v_mfma_f32_32x32x2_f32 — an exact fp32 fmaf
//     chain — two list entries per instruction pair: OUT[64 px][32 ch] += W[64 px][2] . F[2][32 ch].  The A operand is the
//     two per-lane weights after one v_permlane32_swap, the B operand one ds_read_b32 per lane;
//   * every list entry goes through the pair (no per-entry "nobody contributes" branch), which makes the loop body one
//     basic block: the two MFMAs of pair p-1 are issued between the alpha evaluations of pair p, so the matrix pipe runs
//     under the VALU work instead of after it, and the records of pair p+1 / the list slots of pair p+2 are fetched
//     while pair p is evaluated (flat per-quadrant list, hsr_tile_common.h);
//   * an odd list is padded with a dummy slot of opacity 0 and an all-zero feature row.
// Batches are 128 splats (two threads stage one splat, half a feature row each) so that four workgroups fit a CU.
#include "hsr_tile_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned uint2p_ __attribute__((ext_vector_type(2)));

constexpr int PF_BATCH = 128;
constexpr int PF_DUMMY = PF_BATCH;       // slot of the padding record
constexpr int PF_FM = 32;                // feature row: sem[KC], r, g, b, depth, (1.0 for the mask variant), zeros
constexpr int PF_FS = PF_FM + 4;         // LDS row stride (16-byte staging stores of consecutive rows on different banks)

// KC <= 27 semantic channels [0, KC) (channels >= a.K read as 0) + base outputs.  MASK: non-semantic variant.
template <int KC, bool MASK>
__global__ void __launch_bounds__(256, 4) render_fwd_pair_kernel(RenderFwdArgs a)
{
    static_assert(KC + 5 <= PF_FM, "feature row holds sem[KC], r, g, b, depth, mask");
    __shared__ float4 s_geo[PF_BATCH + 1];   // x, y, A, B (pre-scaled conic)
    __shared__ float4 s_cd[PF_BATCH + 1];    // C, opacity, depth, -
    __shared__ float s_feat[(PF_BATCH + 1) * PF_FS > 4 * 32 * 33 ? (PF_BATCH + 1) * PF_FS : 4 * 32 * 33];
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ uint8_t s_flat[4][PF_BATCH + 8];
    __shared__ int s_wdone[4];

    const int tile = hsr_block_tile(blockIdx.x, ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y));
    if (tile >= ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (a.bin.base) {   // speculative forward: the list lives where num_rendered says
        BinState bs;
        if (!hsr_bin_resolve(a.bin, *a.bin.R_dev, &bs)) return;
        a.point_list = bs.vals;
    }
    const TileGeom tg = tile_geom(tile, a.W, a.H, t);
    const bool inside = tg.inside;
    const size_t N = (size_t)a.W * a.H;
    const size_t pix_id = (size_t)a.W * tg.py + tg.px;
    const float pfx = tg.pfx, pfy = tg.pfy;
    const float tile_x0 = (float)(tg.tx * HSR_TILE_X), tile_y0 = (float)(tg.ty * HSR_TILE_Y);
    const uint2 range = a.ranges[tile];
    const int n = (int)(range.y - range.x);

    float T = 1.0f;
    uint32_t last_contributor = 0;
    float median_D = 15.0f;
    bool done = !inside;
    f32x16 D0, D1;  // OUT[pixels 0-31][32 ch], OUT[pixels 32-63][32 ch] of this wave's quadrant
#pragma unroll
    for (int i = 0; i < 16; i++) { D0[i] = 0.f; D1[i] = 0.f; }

    // the padding record: opacity 0 (never contributes), zero feature row (0 * w stays finite)
    if (t == 0) {
        s_geo[PF_DUMMY] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_cd[PF_DUMMY] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (t < PF_FS) s_feat[PF_DUMMY * PF_FS + t] = 0.f;

    // ---- staging: thread t -> entry e = t & 127, half h = t >> 7 of the 32-float feature row; ids two batches ahead,
    // records one batch ahead (registers), as in hsr_render_fwd.hip ----
    const int e = t & (PF_BATCH - 1), half = t >> 7;
    int id_next = 0;
    float2 p_xy = {0, 0};
    float4 p_co = {0, 0, 0, 0};
    float p_d = 0.f;
    float p_row[16];
#pragma unroll
    for (int c = 0; c < 16; c++) p_row[c] = 0.f;
    const bool aligned_rows = (a.K == KC) && (KC % 2 == 0);
    auto load_id = [&](int start) {
        const int i = start + e;
        if (i < n) id_next = (int)a.point_list[range.x + i];
    };
    auto load_record = [&](int start) {
        const int i = start + e;
        if (i < n) {
            const size_t id = (size_t)id_next;
            p_xy = a.means2D[id];
            p_co = a.conic_opacity[id];
            p_d = a.depths[id];
            const float* row = a.semantics + id * (size_t)a.K;   // only dereferenced for channels < a.K
            const float cr = a.colors[3 * id], cg = a.colors[3 * id + 1], cb = a.colors[3 * id + 2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                if (hh != half) continue;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int c0 = 16 * hh + 2 * q;   // compile-time after unrolling
                    float v0 = 0.f, v1 = 0.f;
                    if (c0 + 1 < KC && aligned_rows) {
                        const float2 x = *reinterpret_cast<const float2*>(row + c0);
                        v0 = x.x; v1 = x.y;
                    } else {
                        if (c0 < KC) v0 = c0 < a.K ? row[c0] : 0.f;
                        if (c0 + 1 < KC) v1 = c0 + 1 < a.K ? row[c0 + 1] : 0.f;
                    }
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const int c = c0 + s;
                        float& v = s ? v1 : v0;
                        if (c == KC) v = cr;
                        if (c == KC + 1) v = cg;
                        if (c == KC + 2) v = cb;
                        if (c == KC + 3) v = p_d;
                        if (MASK && c == KC + 4) v = 1.0f;
                    }
                    p_row[2 * q] = v0;
                    p_row[2 * q + 1] = v1;
                }
            }
        }
    };
    load_id(0);
    load_record(0);
    load_id(PF_BATCH);

    // pending pair: A operands (after the permlane swap) and B operand, all in registers
    float pa0 = 0.f, pa1 = 0.f, pb = 0.f;

    for (int start = 0; start < n; start += PF_BATCH) {
        const bool wave_done = __ballot(!done) == 0ull;
        if (lane == 0) s_wdone[wv] = wave_done;
        __syncthreads();  // also: everyone has finished reading the previous batch
        if (s_wdone[0] & s_wdone[1] & s_wdone[2] & s_wdone[3]) break;
        const int cnt = min(PF_BATCH, n - start);
        uint32_t qmask = 0u;
        if (e < cnt) {
            if (half == 0) {
                qmask = quadrant_mask(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0);
                s_geo[e] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
                s_cd[e] = make_float4((-0.5f * HSR_LOG2E) * p_co.z, p_co.w, p_d, 0.f);
            }
            float4* dst = reinterpret_cast<float4*>(&s_feat[e * PF_FS + 16 * half]);
#pragma unroll
            for (int q = 0; q < 4; q++) dst[q] = make_float4(p_row[4 * q], p_row[4 * q + 1], p_row[4 * q + 2], p_row[4 * q + 3]);
        }
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);   // slots 0..127 are staged by waves 0 and 1
        __syncthreads();
        load_record(start + PF_BATCH);
        load_id(start + 2 * PF_BATCH);
        if (wave_done) continue;

        // flat list of this quadrant, padded to an even length (+ slack for the prefetch) with the dummy slot
        int total = 0;
#pragma unroll
        for (int seg = 0; seg < 2; seg++) {
            const int c = s_lcnt[wv][seg];
            if (lane < c) s_flat[wv][total + lane] = s_list[wv][seg * 64 + lane];
            total += c;
        }
        if (lane < 6) s_flat[wv][total + lane] = (uint8_t)PF_DUMMY;
        __builtin_amdgcn_wave_barrier();
        const int npairs = (total + 1) >> 1;
        if (npairs == 0) continue;

        const uint8_t* fl = s_flat[wv];
        int j0 = fl[0], j1 = fl[1];            // slots of the current pair
        int n0 = fl[2], n1 = fl[3];            // slots of the next pair
        float4 g0 = s_geo[j0], g1 = s_geo[j1];
        float4 c0 = s_cd[j0], c1 = s_cd[j1];
        float bcur = s_feat[(lane < 32 ? j0 : j1) * PF_FS + (lane & 31)];
        for (int p = 0; p < npairs; p++) {
            // prefetch: records of pair p+1, slots of pair p+2
            const float4 ng0 = s_geo[n0], ng1 = s_geo[n1];
            const float4 nc0 = s_cd[n0], nc1 = s_cd[n1];
            const float bnext = s_feat[(lane < 32 ? n0 : n1) * PF_FS + (lane & 31)];
            const int m0 = fl[2 * p + 4], m1 = fl[2 * p + 5];

            // ---- first entry of the pair ----
            float w0, w1;
            {
                const float dx = g0.x - pfx, dy = g0.y - pfy;
                const float power2 = fmaf(c0.x, dy * dy, fmaf(g0.w, dx * dy, g0.z * (dx * dx)));  // log2(G)
                const float alpha = fminf(0.99f, c0.y * __builtin_amdgcn_exp2f(power2));
                bool contrib = !done && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                const float test_T = T * (1.0f - alpha);
                const bool stop = contrib && test_T < 0.0001f;
                done = done || stop;
                contrib = contrib && !stop;
                w0 = contrib ? alpha * T : 0.f;
                median_D = (contrib && T > 0.5f && test_T < 0.5f) ? c0.z : median_D;
                T = contrib ? test_T : T;
                last_contributor = contrib ? (uint32_t)(start + j0 + 1) : last_contributor;
            }
            D0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa0, pb, D0, 0, 0, 0);   // pair p-1, pixels 0-31
            // ---- second entry ----
            {
                const float dx = g1.x - pfx, dy = g1.y - pfy;
                const float power2 = fmaf(c1.x, dy * dy, fmaf(g1.w, dx * dy, g1.z * (dx * dx)));
                const float alpha = fminf(0.99f, c1.y * __builtin_amdgcn_exp2f(power2));
                bool contrib = !done && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                const float test_T = T * (1.0f - alpha);
                const bool stop = contrib && test_T < 0.0001f;
                done = done || stop;
                contrib = contrib && !stop;
                w1 = contrib ? alpha * T : 0.f;
                median_D = (contrib && T > 0.5f && test_T < 0.5f) ? c1.z : median_D;
                T = contrib ? test_T : T;
                last_contributor = contrib ? (uint32_t)(start + j1 + 1) : last_contributor;
            }
            D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa1, pb, D1, 0, 0, 0);   // pair p-1, pixels 32-63
            // this pair becomes the pending one
            const uint2p_ sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w0), __float_as_uint(w1), false, false);
            pa0 = __uint_as_float(sw[0]);
            pa1 = __uint_as_float(sw[1]);
            pb = bcur;
            // rotate the pipeline
            j0 = n0; j1 = n1; n0 = m0; n1 = m1;
            g0 = ng0; g1 = ng1; c0 = nc0; c1 = nc1;
            bcur = bnext;
            if (__ballot(!done) == 0ull) break;   // every pixel of the quadrant has terminated
        }
    }
    // retire the pending pair
    D0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa0, pb, D0, 0, 0, 0);
    D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa1, pb, D1, 0, 0, 0);

    // ---- D[i][j]: lane l holds channel j = l & 31, pixels i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r = 0..15 ----
    // transpose through LDS (row stride 33) back to lane = pixel, 32 pixels at a time
    __syncthreads();  // all waves are past their last read of s_feat
    float* tp = s_feat + wv * (32 * 33);
    if (inside) {
        a.final_T[pix_id] = T;
        a.n_contrib[pix_id] = last_contributor;
        a.out_median_depth[pix_id] = median_D;
        a.out_opacity[pix_id] = 1.0f - T;
    }
#pragma unroll
    for (int hsel = 0; hsel < 2; hsel++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            tp[i * 33 + (lane & 31)] = hsel ? D1[r] : D0[r];
        }
        __builtin_amdgcn_wave_barrier();
        if (inside && (lane >> 5) == hsel) {
            const float* mine = tp + (lane & 31) * 33;
            a.out_color[pix_id] = mine[KC];
            a.out_color[N + pix_id] = mine[KC + 1];
            a.out_color[2 * N + pix_id] = mine[KC + 2];
            a.out_depth[pix_id] = mine[KC + 3];
            if (MASK) a.out_mask[pix_id] = mine[KC + 4];
#pragma unroll
            for (int c = 0; c < KC; c++)
                if (c < a.K) a.out_semantic[(size_t)c * N + pix_id] = mine[c];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// non-semantic variant and semantic K <= 27; returns false otherwise
bool hsr_launch_render_forward_pair(const RenderFwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles)), block(256);
    if (!a.semantic) { render_fwd_pair_kernel<0, true><<<grid, block, 0, stream>>>(a); return true; }
    if (a.K > 27) return false;
    if (a.K == 0) render_fwd_pair_kernel<0, false><<<grid, block, 0, stream>>>(a);
    else if (a.K == 16) render_fwd_pair_kernel<16, false><<<grid, block, 0, stream>>>(a);
    else if (a.K == 26) render_fwd_pair_kernel<26, false><<<grid, block, 0, stream>>>(a);
    else render_fwd_pair_kernel<27, false><<<grid, block, 0, stream>>>(a);
    return true;
}
