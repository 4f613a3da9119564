// hsr_render_fwd_mfma.hip — forward tile kernel with the blend accumulation on the matrix cores.
//
// Per-pixel semantics are those of hsr_render_fwd.hip (reference forward.cu:261-538).  The difference is where
//        out[pixel][channel] += w[pixel] * feature[splat][channel]       (forward.cu:365-368, :503-508)
// is evaluated for the K semantic channels + r, g, b + depth (+ the mask of the non-semantic variant): that is a
// rank-1 update per accepted splat, i.e. a dense product  OUT[64 pixels][32 ch] += W[64 px][splats] . F[splats][32 ch].
// The tile kernels are VALU-issue bound on gfx950 while the MFMA pipe idles, and v_mfma_f32_32x32x2_f32 is an
// exact fp32 fmaf chain, so two accepted splats at a time go there:
//   * the per-lane weights of two consecutive accepted splats (lane = pixel) are exactly the A operand of a
//     32x32x2 MFMA after ONE v_permlane32_swap: {w0.lo | w1.lo} = A[pixels 0-31][k = 0,1], {w0.hi | w1.hi} for 32-63;
//   * the B operand is one ds_read_b32 per lane: lane l reads feature row (l < 32 ? splat0 : splat1), column l & 31
//     — instead of seven ds_read_b128 broadcasts and 15 packed FMAs per accepted splat;
//   * the two 32x32 accumulators (32 VGPRs) replace the 30 per-lane accumulators; at the end of the tile they are
//     transposed through LDS back to lane = pixel and stored with the same coalesced pattern as before.
// T, alpha tests, termination, n_contrib and median depth stay on the VALU path unchanged.
#include "hsr_tile_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned uint2v_ __attribute__((ext_vector_type(2)));

constexpr int FM_CH = 32;  // feature row: sem[KC], r, g, b, depth, (1.0 for the mask variant), zero padding

// KC <= 27 semantic channels [0, KC) (channels >= a.K masked) + base outputs.  MASK: non-semantic variant.
template <int KC, bool MASK>
__global__ void __launch_bounds__(256) render_fwd_mfma_kernel(RenderFwdArgs a)
{
    constexpr int BATCH = 256;
    static_assert(KC + 5 <= FM_CH, "feature row holds sem[KC], r, g, b, depth, mask");
    __shared__ float4 s_geo[BATCH];          // x, y, A, B (pre-scaled conic)
    __shared__ float2 s_co[BATCH];           // C, opacity
    __shared__ float s_feat[4 * 64 * 33 > BATCH * FM_CH ? 4 * 64 * 33 : BATCH * FM_CH];  // feature rows; output transpose at the end
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ int s_wdone[4];

    const int tile = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
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
    float pend_w = 0.f;  // weights of an accepted splat waiting for a partner
    int pend_j = -1;     // its batch slot (wave-uniform)

    // ---- software-pipelined staging registers (lane t <-> splat t of a batch), as in hsr_render_fwd.hip ----
    int id_next = 0;
    float2 p_xy = {0, 0};
    float4 p_co = {0, 0, 0, 0};
    float p_r = 0, p_g = 0, p_b = 0, p_d = 0;
    float p_sem[KC > 0 ? KC : 1];
#pragma unroll
    for (int c = 0; c < (KC > 0 ? KC : 1); c++) p_sem[c] = 0.f;
    const bool aligned_rows = (a.K == KC) && (KC % 2 == 0);
    auto load_id = [&](int start) {
        const int i = start + t;
        if (i < n) id_next = (int)a.point_list[range.x + i];
    };
    auto load_record = [&](int start) {
        const int i = start + t;
        if (i < n) {
            const size_t id = (size_t)id_next;
            p_xy = a.means2D[id];
            p_co = a.conic_opacity[id];
            p_d = a.depths[id];
            p_r = a.colors[3 * id];
            p_g = a.colors[3 * id + 1];
            p_b = a.colors[3 * id + 2];
            if (KC > 0) {
                if (aligned_rows) {
                    const float2* row = reinterpret_cast<const float2*>(a.semantics + id * (size_t)KC);
#pragma unroll
                    for (int q = 0; q < KC / 2; q++) {
                        const float2 v = row[q];
                        p_sem[2 * q] = v.x;
                        p_sem[2 * q + 1] = v.y;
                    }
                } else {
                    const float* row = a.semantics + id * (size_t)a.K;
#pragma unroll
                    for (int c = 0; c < KC; c++) p_sem[c] = c < a.K ? row[c] : 0.f;
                }
            }
        }
    };
    load_id(0);
    load_record(0);
    load_id(BATCH);

    // one rank-2 update: OUT += w0 (x) F[j0] + w1 (x) F[j1]
    auto mfma_pair = [&](float w0, int j0, float w1, int j1) {
        const uint2v_ sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w0), __float_as_uint(w1), false, false);
        const float b = s_feat[(lane < 32 ? j0 : j1) * FM_CH + (lane & 31)];
        D0 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[0]), b, D0, 0, 0, 0);
        D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[1]), b, D1, 0, 0, 0);
    };

    for (int start = 0; start < n; start += BATCH) {
        const bool wave_done = __ballot(!done) == 0ull;
        if (lane == 0) s_wdone[wv] = wave_done;
        __syncthreads();  // also: everyone has finished reading the previous batch
        if (s_wdone[0] & s_wdone[1] & s_wdone[2] & s_wdone[3]) break;
        const int cnt = min(BATCH, n - start);
        uint32_t qmask = 0u;
        if (t < cnt) {
            qmask = quadrant_mask(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0);
            s_geo[t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_co[t] = make_float2((-0.5f * HSR_LOG2E) * p_co.z, p_co.w);
            float row[FM_CH];
#pragma unroll
            for (int c = 0; c < FM_CH; c++) row[c] = 0.f;
#pragma unroll
            for (int c = 0; c < KC; c++) row[c] = p_sem[c];
            row[KC] = p_r; row[KC + 1] = p_g; row[KC + 2] = p_b; row[KC + 3] = p_d;
            if (MASK) row[KC + 4] = 1.0f;
            float4* dst = reinterpret_cast<float4*>(&s_feat[t * FM_CH]);
#pragma unroll
            for (int q = 0; q < FM_CH / 4; q++) dst[q] = make_float4(row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]);
        }
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        __syncthreads();
        load_record(start + BATCH);
        load_id(start + 2 * BATCH);
        if (wave_done) continue;

        for (int seg = 0; seg < 4; seg++) {
            const int m = s_lcnt[wv][seg];
            for (int k = 0; k < m; k++) {
                const int j = s_list[wv][seg * 64 + k];
                const float4 g = s_geo[j];
                const float2 co = s_co[j];
                const float dx = g.x - pfx, dy = g.y - pfy;
                const float power2 = fmaf(co.x, dy * dy, fmaf(g.w, dx * dy, g.z * (dx * dx)));  // log2(G)
                const float alpha = fminf(0.99f, co.y * __builtin_amdgcn_exp2f(power2));
                bool contrib = !done && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                const float test_T = T * (1.0f - alpha);
                if (contrib && test_T < 0.0001f) {
                    done = true;
                    contrib = false;
                }
                if (__ballot(contrib) == 0ull) continue;
                const float w = contrib ? alpha * T : 0.f;
                const bool cross = contrib && T > 0.5f && test_T < 0.5f;
                if (__ballot(cross) != 0ull) {
                    const float dep = s_feat[j * FM_CH + KC + 3];
                    if (cross) median_D = dep;
                }
                if (contrib) {
                    T = test_T;
                    last_contributor = (uint32_t)(start + j + 1);
                }
                if (pend_j < 0) {
                    pend_w = w;
                    pend_j = j;
                } else {
                    mfma_pair(pend_w, pend_j, w, j);
                    pend_j = -1;
                }
            }
        }
        // the feature rows of this batch are about to be overwritten: retire a waiting splat with a zero partner
        if (pend_j >= 0) {
            mfma_pair(pend_w, pend_j, 0.f, pend_j);
            pend_j = -1;
        }
    }

    // ---- D[i][j]: lane l holds channel j = l & 31, pixels i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r = 0..15 ----
    // transpose through LDS (row stride 33) back to lane = pixel, then the usual coalesced stores
    __syncthreads();  // all waves are past their last read of s_feat
    float* tp = s_feat + wv * (64 * 33);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        tp[i * 33 + (lane & 31)] = D0[r];
        tp[(32 + i) * 33 + (lane & 31)] = D1[r];
    }
    __builtin_amdgcn_wave_barrier();
    if (inside) {
        a.final_T[pix_id] = T;
        a.n_contrib[pix_id] = last_contributor;
        const float* mine = tp + lane * 33;
        a.out_color[pix_id] = mine[KC];
        a.out_color[N + pix_id] = mine[KC + 1];
        a.out_color[2 * N + pix_id] = mine[KC + 2];
        a.out_depth[pix_id] = mine[KC + 3];
        a.out_median_depth[pix_id] = median_D;
        a.out_opacity[pix_id] = 1.0f - T;
        if (MASK) a.out_mask[pix_id] = mine[KC + 4];
#pragma unroll
        for (int c = 0; c < KC; c++)
            if (c < a.K) a.out_semantic[(size_t)c * N + pix_id] = mine[c];
    }
}

}  // namespace

// covers the base outputs and the first min(K, 27) semantic channels; returns how many channels it produced
int hsr_launch_render_forward_mfma(const RenderFwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(tiles), block(256);
    if (!a.semantic) { render_fwd_mfma_kernel<0, true><<<grid, block, 0, stream>>>(a); return 0; }
    if (a.K == 0) { render_fwd_mfma_kernel<0, false><<<grid, block, 0, stream>>>(a); return 0; }
    if (a.K == 16) { render_fwd_mfma_kernel<16, false><<<grid, block, 0, stream>>>(a); return 16; }
    if (a.K == 26) { render_fwd_mfma_kernel<26, false><<<grid, block, 0, stream>>>(a); return 26; }
    render_fwd_mfma_kernel<27, false><<<grid, block, 0, stream>>>(a);
    return a.K < 27 ? a.K : 27;
}
