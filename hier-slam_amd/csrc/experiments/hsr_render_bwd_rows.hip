// hsr_render_bwd_rows.hip — experimental backward tile kernel: matrix-core sums, NO global atomics.
//
// The atomics of the other two backward kernels are what bounds them on MI355X: float atomics execute at
// the memory side in 64-byte requests at a fixed chip-wide rate (MI355X guide, "Global float atomics"), and a
// splat's 10+K sums live in six different arrays, so one (wave, splat) costs ~6 requests — 12-17 M requests
// per backward at the headline workload, 0.16-0.48 ms (measured by dropping them, tools/ablate.sh).
// This kernel keeps the arithmetic of hsr_render_bwd_mfma.hip (same per-pixel semantics as the reference's
// backward.cu:472-899) but
//   * combines the four quadrant waves of a tile in an LDS accumulator s_acc[batch slot][ROW] with ds_add_f32
//     (butterfly lanes and MFMA result columns hit distinct LDS addresses: conflict-free), and
//   * writes each list position's ROW floats once, with plain coalesced stores, to rows[position][ROW].
// The per-Gaussian sum over its rows (contiguous in emission order, reached through the inverse permutation
// built by inverse_map_kernel) is fused into preprocess_backward_kernel.  Consequences: no zero-fill of the
// gradient arrays and no global atomics anywhere in the backward.  NOT yet bit-reproducible: the four waves
// still meet in LDS through ds_add_f32, whose arrival order varies.  Status (r01): 0.64 ms + 0.12 ms gather vs
// 0.68 ms for the atomic VALU kernel — more barriers (128-entry batches) and 3 blocks/CU cost what the atomics
// saved, so this path is opt-in (diff_gaussian_rasterization._C.deterministic_backward / HSR_BWD_IMPL=rows).
#include "hsr_tile_common.h"
#include "hsr_wave_reduce.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_SLOTS = 16;
constexpr int MF_STRIDE = 66;
constexpr int MF_PANEL = 64 * 17;
constexpr int ROWS_BATCH = 128;  // list entries staged per round (LDS: 128 * ROW * 4 B accumulator)

// KC semantic channels [0, KC) together with the 10 base sums; channels >= a.K are masked.  Nothing is added
// to global memory atomically: sums are combined over the tile's four waves in an LDS accumulator
// s_acc[slot][ROW] (ds_add_f32 on distinct addresses) and each list position's row is written ONCE with
// plain coalesced stores to rows[range.x + pos][ROW]; the per-Gaussian sum over its rows happens in
// preprocess_backward_kernel.  Row layout: [0..6] butterfly values, [8 + ch] the MFMA channels.
template <int KC>
__global__ void __launch_bounds__(256) render_bwd_rows_kernel(RenderBwdArgs a)
{
    constexpr int BATCH = ROWS_BATCH;
    constexpr int ROW = 8 + 16 * ((KC + 5 + 15) / 16);
    constexpr int NCH = KC + 5;               // sem[KC], r, g, b, depth, opacity(direct)
    constexpr int NG = (NCH + 15) / 16;       // 16-channel groups
    static_assert(NG <= 2, "at most 32 direct channels per launch");
    constexpr int NV = 7;                     // VALU butterfly: mean2D.xy, conic.xyw, opacity(alpha path), depth(median)
    __shared__ float4 s_geo[BATCH];
    __shared__ float2 s_co[BATCH];
    __shared__ float4 s_col[BATCH];
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ int s_wmax[4];
    __shared__ float s_panel[4][MF_PANEL];
    __shared__ int s_slot_id[4][MF_SLOTS];   // batch slot j of each panel row
    __shared__ float s_acc[BATCH * ROW];

    const int tile = hsr_block_tile(blockIdx.x, ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y));
    if (tile >= ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const TileGeom tg = tile_geom(tile, a.W, a.H, t);
    const bool inside = tg.inside;
    const size_t N = (size_t)a.W * a.H;
    const size_t pix_id = (size_t)a.W * tg.py + tg.px;
    const float pfx = tg.pfx, pfy = tg.pfy;
    const float tile_x0 = (float)(tg.tx * HSR_TILE_X), tile_y0 = (float)(tg.ty * HSR_TILE_Y);
    const uint2 range = a.ranges[tile];
    float* panel = s_panel[wv];

    const float T_final = inside ? a.final_T[pix_id] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)a.n_contrib[pix_id] : 0;

    int wmax = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    if (lane == 0) s_wmax[wv] = wmax;

    // ---- upstream gradients of this lane's pixel, then the MFMA B operand (G transposed through LDS) ----
    float dpx0 = 0, dpx1 = 0, dpx2 = 0, dpd = 0, dpm = 0, dpo = 0;
    if (inside) {
        dpx0 = a.dL_dpix[pix_id];
        dpx1 = a.dL_dpix[N + pix_id];
        dpx2 = a.dL_dpix[2 * N + pix_id];
        dpd = a.dL_dpix_depth[pix_id];
        dpm = a.dL_dpix_median[pix_id];
        dpo = a.dL_dpix_opacity[pix_id];
    }
    float Breg[NG][16];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        // channels [16g, 16g+16) of this lane's pixel -> panel[pixel][c] (row stride 17)
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int ch = 16 * g + c;
            float v = 0.f;
            if (ch < KC) {
                if (inside && ch < a.K) v = a.dL_dpix_sem[(size_t)ch * N + pix_id];
            } else if (ch == KC) v = dpx0;
            else if (ch == KC + 1) v = dpx1;
            else if (ch == KC + 2) v = dpx2;
            else if (ch == KC + 3) v = dpd;
            else if (ch == KC + 4) v = dpo;
            panel[lane * 17 + c] = v;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) Breg[g][m] = panel[(4 * m + (lane >> 4)) * 17 + (lane & 15)];
        __syncthreads();
    }
    const int hi_all = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    const float bg_dot = a.bg[0] * dpx0 + a.bg[1] * dpx1 + a.bg[2] * dpx2;
    const float kx = (0.5f * a.W) / HSR_LOG2E, ky = (0.5f * a.H) / HSR_LOG2E;
    float Rb = 0.f, last_h = 0.f, last_alpha = 0.f;

    const int myv = reduce_slot(lane);  // butterfly value (= row column) this lane ends up holding

    int nslot = 0;  // wave-uniform: accepted splats waiting in the panel
    auto flush = [&]() {
        f32x4 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* arow = panel + (lane & 15) * MF_STRIDE + (lane >> 4);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const float av = arow[4 * m];
#pragma unroll
            for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Breg[g][m], acc[g], 0, 0, 0);
        }
        // D[row = 4*(lane>>4) + r][col = lane&15]: row = panel slot, col = channel within the group
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int slot = 4 * (lane >> 4) + r;
            if (slot < nslot) {
                float* dst = s_acc + s_slot_id[wv][slot] * ROW + 8 + (lane & 15);
#pragma unroll
                for (int g = 0; g < NG; g++) atomicAdd(dst + 16 * g, acc[g][r]);  // ds_add_f32
            }
        }
        nslot = 0;
    };

    // ---- software-pipelined staging, as in hsr_render_bwd.hip ----
    int id_next = 0;
    float2 p_xy = {0, 0};
    float4 p_co = {0, 0, 0, 0};
    float p_r = 0, p_g = 0, p_b = 0, p_d = 0;
    auto load_id = [&](int hi) {
        if (t < BATCH && hi - 1 - t >= 0) id_next = (int)a.point_list[range.x + hi - 1 - t];
    };
    auto load_record = [&](int hi) {
        if (t < BATCH && hi - 1 - t >= 0) {
            const size_t id = (size_t)id_next;
            p_xy = a.means2D[id];
            p_co = a.conic_opacity[id];
            p_r = a.colors[3 * id];
            p_g = a.colors[3 * id + 1];
            p_b = a.colors[3 * id + 2];
            p_d = a.depths[id];
        }
    };
    load_id(hi_all);
    load_record(hi_all);
    load_id(hi_all - BATCH);

    float* rows = a.rows;
    // list entries behind the tile's farthest contributor are never visited: their rows are zero
    {
        const int n = (int)(range.y - range.x);
        float* z = rows + ((size_t)range.x + hi_all) * ROW;
        for (int e = t; e < (n - hi_all) * ROW; e += 256) z[e] = 0.f;
    }
    for (int e = t; e < BATCH * ROW; e += 256) s_acc[e] = 0.f;
    // writes the rows of the batch whose back end is list position `hi_prev` (slot j <-> position hi_prev-1-j)
    // and clears the accumulator; callers put a barrier before (all waves done adding) and after
    auto store_rows = [&](int hi_prev) {
        const int cnt_prev = min(BATCH, hi_prev);
        for (int e = t; e < cnt_prev * ROW; e += 256) {
            const int j = e / ROW, c = e - j * ROW;
            rows[((size_t)range.x + hi_prev - 1 - j) * ROW + c] = s_acc[e];
            s_acc[e] = 0.f;
        }
    };
    for (int hi = hi_all; hi > 0; hi -= BATCH) {
        const int cnt = min(BATCH, hi);
        __syncthreads();
        if (hi != hi_all) {
            if (nslot > 0) flush();
            __syncthreads();  // every wave's pending panel rows are in s_acc
            store_rows(hi + BATCH);
        }
        uint32_t qmask = 0u;
        if (t < cnt) {
            qmask = quadrant_mask(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0);
            s_geo[t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_co[t] = make_float2((-0.5f * HSR_LOG2E) * p_co.z, p_co.w);
            s_col[t] = make_float4(p_r, p_g, p_b, p_d);
        }
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        __syncthreads();
        load_record(hi - BATCH);
        load_id(hi - 2 * BATCH);
        if (hi - cnt < wmax) {  // else: this wave's pixels all stopped in front of this batch
            for (int seg = 0; seg < 4; seg++) {
                const int m = s_lcnt[wv][seg];
                for (int k = 0; k < m; k++) {
                    const int j = s_list[wv][seg * 64 + k];
                    const float4 g = s_geo[j];
                    const float2 co = s_co[j];
                    const int pos = hi - 1 - j;
                    const float dx = g.x - pfx, dy = g.y - pfy;
                    const float dxx = dx * dx, dxy = dx * dy, dyy = dy * dy;
                    const float power2 = fmaf(co.x, dyy, fmaf(g.w, dxy, g.z * dxx));
                    const float G = __builtin_amdgcn_exp2f(power2);
                    const float alpha = fminf(0.99f, co.y * G);
                    const bool active = pos < last_contributor && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                    if (__ballot(active) == 0ull) continue;

                    const float inv_one_m_a = __builtin_amdgcn_rcpf(1.0f - alpha);
                    const float test_T = T * inv_one_m_a;
                    const float w = active ? alpha * test_T : 0.f;
                    // direct sums go through the panel -> MFMA
                    panel[nslot * MF_STRIDE + lane] = w;
                    if (lane == 0) s_slot_id[wv][nslot] = j;

                    const float4 cd = s_col[j];
                    const float h = fmaf(cd.x, dpx0, fmaf(cd.y, dpx1, fmaf(cd.z, dpx2, fmaf(cd.w, dpd, dpo))));
                    const float Rn = fmaf(last_alpha, last_h - Rb, Rb);
                    float dL_dalpha = (h - Rn) * test_T;
                    dL_dalpha += (-T_final * inv_one_m_a) * bg_dot;
                    const float Gs = active ? G : 0.f;
                    const float gda = Gs * dL_dalpha;
                    const float q = co.y * gda;
                    float v[NV];
                    v[0] = q * fmaf(2.0f * g.z, dx, g.w * dy) * kx;
                    v[1] = q * fmaf(2.0f * co.x, dy, g.w * dx) * ky;
                    const float hq = -0.5f * q;
                    v[2] = hq * dxx;
                    v[3] = hq * dxy;
                    v[4] = hq * dyy;
                    v[5] = gda;
                    v[6] = (active && test_T > 0.5f && T < 0.5f) ? dpm : 0.f;
                    if (active) {
                        Rb = Rn;
                        last_h = h;
                        last_alpha = alpha;
                        T = test_T;
                    }
                    const float total = wave_reduce_transpose<NV>(v, lane);
                    if (myv < NV) atomicAdd(s_acc + j * ROW + myv, total);  // ds_add_f32, 7 distinct addresses
                    nslot++;
                    if (nslot == MF_SLOTS) flush();
                }
            }
        }
    }
    if (nslot > 0) flush();
    __syncthreads();
    if (hi_all > 0) {
        const int last_hi = hi_all - ((hi_all - 1) / BATCH) * BATCH;  // back end of the last batch processed
        store_rows(last_hi);
    }
}


// inv[emission index u] = sorted list position i.  Emission order (rasterizer_impl.cu:98-108) keeps the
// instances of one Gaussian contiguous: u = offsets[g-1] + (ty - y0) * (x1 - x0) + (tx - x0).
__global__ void __launch_bounds__(256) inverse_map_kernel(int R, int tiles_x, int tiles_y, const uint64_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ vals, const float2* __restrict__ means2D,
                                                          const int* __restrict__ radii, const uint32_t* __restrict__ offsets,
                                                          uint32_t* __restrict__ inv)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const uint32_t gid = vals[i];
    const uint32_t tile = (uint32_t)(keys[i] >> 32);
    const int tx = (int)(tile % (uint32_t)tiles_x), ty = (int)(tile / (uint32_t)tiles_x);
    const float2 xy = means2D[gid];
    const float r = (float)radii[gid];
    // same rect as the key emission (reference getRect, auxiliary.h:46-56)
    int v;
    v = (int)((xy.x - r) / 16.0f); v = v < 0 ? 0 : v; const int x0 = v < tiles_x ? v : tiles_x;
    v = (int)((xy.y - r) / 16.0f); v = v < 0 ? 0 : v; const int y0 = v < tiles_y ? v : tiles_y;
    v = (int)((((xy.x + r) + 16.0f) - 1.0f) / 16.0f); v = v < 0 ? 0 : v; const int x1 = v < tiles_x ? v : tiles_x;
    const uint32_t base = gid == 0 ? 0u : offsets[gid - 1];
    inv[base + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0))] = (uint32_t)i;
}

}  // namespace

int hsr_rows_row_floats(int K) { return 8 + 16 * ((K + 5 + 15) / 16); }
bool hsr_rows_supported(int K) { return K >= 0 && K <= 27; }

int hsr_launch_inverse_map(int R, int tiles_x, int tiles_y, const uint64_t* keys, const uint32_t* vals, const float2* means2D,
                           const int* radii, const uint32_t* offsets, uint32_t* inv, hipStream_t stream)
{
    if (R > 0) inverse_map_kernel<<<(R + 255) / 256, 256, 0, stream>>>(R, tiles_x, tiles_y, keys, vals, means2D, radii, offsets, inv);
    return HSR_OK;
}

// returns the number of semantic channels the kernel was instantiated for (its row layout)
int hsr_launch_render_backward_rows(const RenderBwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles)), block(256);
    const int K = a.semantic ? a.K : 0;
    if (K <= 11) { render_bwd_rows_kernel<11><<<grid, block, 0, stream>>>(a); return 11; }
    if (K == 16) { render_bwd_rows_kernel<16><<<grid, block, 0, stream>>>(a); return 16; }
    if (K == 26) { render_bwd_rows_kernel<26><<<grid, block, 0, stream>>>(a); return 26; }
    render_bwd_rows_kernel<27><<<grid, block, 0, stream>>>(a);
    return 27;
}
