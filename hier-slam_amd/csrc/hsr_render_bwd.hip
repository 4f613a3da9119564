// hsr_render_bwd.hip — 16x16-tile backward alpha compositing for gfx950 (wave64, 4 waves per tile).
//
// Per-pixel semantics follow the reference's backward renderCUDA / renderCUDA_SEM
// (cuda_rasterizer/backward.cu:472-666, :669-899): back-to-front re-traversal from final_T /
// n_contrib, T recovered as T/(1-alpha), the same alpha thresholds, the bg term, the median-depth
// gradient at the T=0.5 crossing, final opacity treated as a blended channel of ones whose colour
// gradient is ADDED to dL_dopacity (backward.cu:628-632, :859-864), and — as observed in the
// reference — no semantic->alpha term (backward.cu:834 reads a scratch buffer that is never
// written, rasterizer_impl.cu:673-674): the semantic loss reaches dL_dsemantics only.
//
// The CDNA4 design point is the accumulation.  The reference issues 11+K fp32 global atomicAdds per
// (pixel, splat) pair (backward.cu:616-663, :828-896).  Here the 10+K per-lane terms of one splat
// are summed across the 64 lanes of a wave with a *transposing* butterfly: each stage pairs two
// registers and halves the lane set (v_permlane32_swap, v_permlane16_swap, then DPP row_ror:8 /
// row_half_mirror / quad_perm), so N values cost ~3 N instructions instead of 6 N and end up one
// per lane.  One global_atomic_add_f32 wave-instruction
// per (wave, splat) then carries all 10+K sums.  Waves own 8x8 quadrants and walk compacted
// per-quadrant lists (hsr_tile_common.h); of the survivors, a splat no lane accepts is skipped.
#include <stdlib.h>
#include <string.h>

#include "hsr_tile_common.h"
#include "hsr_wave_reduce.h"

namespace {

// out[c] = w * in[c] with v_pk_mul_f32 (two products per VALU issue slot; the kernel is VALU-issue bound)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void mul_packed(float w, const float* in, float* out)
{
    const f32x2 w2 = {w, w};
#pragma unroll
    for (int c = 0; c + 1 < N; c += 2) {
        const f32x2 a = {in[c], in[c + 1]};
        const f32x2 r = a * w2;
        out[c] = r.x;
        out[c + 1] = r.y;
    }
    if (N & 1) out[N - 1] = w * in[N - 1];
}

// BASE: this launch produces the 10 geometric/colour sums (and the first KC semantic channels);
// !BASE: semantic channels [c0, c0+KC) only (generic-K chunking).
template <int KC, bool BASE>
__global__ void __launch_bounds__(256) render_bwd_kernel(RenderBwdArgs a, int c0)
{
    constexpr int BATCH = 256;
    constexpr int NV = BASE ? 10 + KC : (KC > 0 ? KC : 1);
    __shared__ float4 s_geo[BATCH];  // x, y, A, B  (pre-scaled conic, hsr_tile_common.h)
    __shared__ float2 s_co[BATCH];   // C, opacity
    __shared__ float4 s_col[BATCH];  // r, g, b, depth
    __shared__ int s_id[BATCH];
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ uint8_t s_flat[4][256];
    __shared__ int s_wmax[4];

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

    const float T_final = inside ? a.final_T[pix_id] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)a.n_contrib[pix_id] : 0;
    // list position of the splat at which this pixel's T crossed 0.5 in the forward (-1: never): it receives dL_dmedian_depth
    const int median_at = (inside && BASE ? (int)a.median_pos[pix_id] : 0) - 1;

    // nothing behind the tile's farthest contributor can receive gradient: start there
    int wmax = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    if (lane == 0) s_wmax[wv] = wmax;
    __syncthreads();
    const int hi_all = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    float dpx0 = 0, dpx1 = 0, dpx2 = 0, dpd = 0, dpm = 0, dpo = 0;
    float dsem[KC > 0 ? KC : 1];
#pragma unroll
    for (int c = 0; c < (KC > 0 ? KC : 1); c++) dsem[c] = 0.f;
    if (inside) {
        if (BASE) {
            dpx0 = a.dL_dpix[pix_id];
            dpx1 = a.dL_dpix[N + pix_id];
            dpx2 = a.dL_dpix[2 * N + pix_id];
            dpd = a.dL_dpix_depth[pix_id];
            dpm = a.dL_dpix_median[pix_id];
            dpo = a.dL_dpix_opacity[pix_id];
        }
#pragma unroll
        for (int c = 0; c < KC; c++)
            if (c0 + c < a.K) dsem[c] = a.dL_dpix_sem[(size_t)(c0 + c) * N + pix_id];
    }
    const float bg_dot = BASE ? a.bg[0] * dpx0 + a.bg[1] * dpx1 + a.bg[2] * dpx2 : 0.f;
    // d(pixel)/d(ndc) = 0.5*W, 0.5*H (backward.cu:550-551); 1/log2(e) undoes the conic pre-scale
    const float kx = (0.5f * a.W) / HSR_LOG2E, ky = (0.5f * a.H) / HSR_LOG2E;

    float Rb = 0.f, last_h = 0.f, last_alpha = 0.f;  // blended "colour . gradient" behind the current splat

    // per-lane atomic target for the value this lane ends up holding after the transposing reduction
    const int myv = reduce_slot(lane);
    float* tgt_base = nullptr;
    int tgt_stride = 0;
    if (a.grow) {
        // packed per-Gaussian row (hsr_tile_common.h): one update = 144 contiguous bytes = 3 lines instead of 6
        tgt_stride = a.grow_stride;
        if (BASE) {
            if (myv < 6) tgt_base = a.grow + myv;                                   // mean2D.xy, conic.xyw, opacity (total)
            else if (myv < 9) tgt_base = a.grow + hsr_grow_direct0(a.K) + (myv - 6); // rgb
            else if (myv == 9) tgt_base = a.grow + 6;                               // depth (total)
            else if (myv < NV && c0 + (myv - 10) < a.K) tgt_base = a.grow + HSR_GROW_SEM0 + c0 + (myv - 10);
        } else if (myv < NV && c0 + myv < a.K) {
            tgt_base = a.grow + HSR_GROW_SEM0 + c0 + myv;
        }
    } else if (BASE) {
        if (myv < 2) { tgt_base = a.dL_dmean2D + myv; tgt_stride = 3; }
        else if (myv < 5) { tgt_base = a.dL_dconic + (myv == 4 ? 3 : myv - 2); tgt_stride = 4; }
        else if (myv == 5) { tgt_base = a.dL_dopacity; tgt_stride = 1; }
        else if (myv < 9) { tgt_base = a.dL_dcolor + (myv - 6); tgt_stride = 3; }
        else if (myv == 9) { tgt_base = a.dL_ddepth; tgt_stride = 1; }
        else if (myv < NV && c0 + (myv - 10) < a.K) { tgt_base = a.dL_dsemantics + c0 + (myv - 10); tgt_stride = a.K; }
    } else {
        if (myv < NV && c0 + myv < a.K) { tgt_base = a.dL_dsemantics + c0 + myv; tgt_stride = a.K; }
    }

    // ---- software-pipelined staging (lane t <-> list position hi-1-t of a batch): ids two batches
    // ahead, records one batch ahead, so the dependent global round trips hide behind the blend ----
    int id_next = 0, id_cur = 0;
    float2 p_xy = {0, 0};
    float4 p_co = {0, 0, 0, 0};
    float p_r = 0, p_g = 0, p_b = 0, p_d = 0;
    auto load_id = [&](int hi) {
        if (hi - 1 - t >= 0) id_next = (int)a.point_list[range.x + hi - 1 - t];
    };
    auto load_record = [&](int hi) {
        if (hi - 1 - t >= 0) {
            const size_t id = (size_t)id_next;
            id_cur = id_next;
            p_xy = a.means2D[id];
            p_co = a.conic_opacity[id];
            if (BASE) {
                p_r = a.colors[3 * id];
                p_g = a.colors[3 * id + 1];
                p_b = a.colors[3 * id + 2];
                p_d = a.depths[id];
            }
        }
    };
    load_id(hi_all);
    load_record(hi_all);
    load_id(hi_all - BATCH);

    for (int hi = hi_all; hi > 0; hi -= BATCH) {
        const int cnt = min(BATCH, hi);
        __syncthreads();
        uint32_t qmask = 0u;
        if (t < cnt) {
            // slot 0 is the farthest entry of this batch (list position hi-1), like the reference's
            // reverse staging (backward.cu:562, :771)
            qmask = quadrant_mask_exact(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0);
            s_id[t] = id_cur;
            s_geo[t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_co[t] = make_float2((-0.5f * HSR_LOG2E) * p_co.z, p_co.w);
            if (BASE) s_col[t] = make_float4(p_r, p_g, p_b, p_d);
        }
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        __syncthreads();
        load_record(hi - BATCH);
        load_id(hi - 2 * BATCH);
        if (hi - cnt >= wmax) continue;  // this wave's pixels all stopped in front of this batch

        const int total = build_flat_list(wv, lane, s_list, s_lcnt, s_flat);
        {
            // slot + record of the next splat are fetched one iteration ahead (stale tail entries are
            // fetched and never used)
            int j1 = s_flat[wv][0];
            float4 gn = s_geo[j1];
            float2 con = s_co[j1];
            float4 cdn = s_col[j1];
            int j2 = s_flat[wv][1];
            for (int i = 0; i < total; i++) {
                const int j = j1;
                const float4 g = gn;
                const float2 co = con;
                const float4 cd = cdn;
                j1 = j2;
                gn = s_geo[j1];
                con = s_co[j1];
                cdn = s_col[j1];
                j2 = s_flat[wv][(i + 2) & 255];
                const int pos = hi - 1 - j;  // 0-based list position == reference `contributor` after decrement
                const float dx = g.x - pfx, dy = g.y - pfy;
                const float dxx = dx * dx, dxy = dx * dy, dyy = dy * dy;
                const float power2 = fmaf(co.x, dyy, fmaf(g.w, dxy, g.z * dxx));  // log2(G)
                const float G = __builtin_amdgcn_exp2f(power2);
                const float alpha = fminf(0.99f, co.y * G);
                const bool active = pos < last_contributor && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                if (__ballot(active) == 0ull) continue;

                // one v_rcp_f32 (1 ulp) instead of the reference's two IEEE divisions by (1 - alpha)
                // (backward.cu:594, :644): well inside the 1e-4 bar, ~20 instructions per pair cheaper
                const float inv_one_m_a = __builtin_amdgcn_rcpf(1.0f - alpha);
                const float test_T = T * inv_one_m_a;
                const float w = active ? alpha * test_T : 0.f;
                float v[NV];
                if (BASE) {
                    // The reference keeps one "colour behind me" accumulator per channel (rgb, depth and the
                    // opacity channel of ones: backward.cu:604-632) and dots each with its upstream gradient.
                    // The recurrence is linear, so the dot product can be taken first: one scalar
                    //   h = c . dL_dpixel,   R <- last_alpha * last_h + (1 - last_alpha) * R,
                    // gives dL_dalpha = (h - R) * T_before, identical up to fp32 rounding with 3 state
                    // registers instead of 11 and a third of the instructions.
                    const float h = fmaf(cd.x, dpx0, fmaf(cd.y, dpx1, fmaf(cd.z, dpx2, fmaf(cd.w, dpd, dpo))));
                    const float Rn = fmaf(last_alpha, last_h - Rb, Rb);
                    float dL_dalpha = (h - Rn) * test_T;
                    v[6] = w * dpx0;
                    v[7] = w * dpx1;
                    v[8] = w * dpx2;
                    // depth (+ median-depth gradient at the T = 0.5 crossing, backward.cu:618-626)
                    v[9] = w * dpd + ((active && pos == median_at) ? dpm : 0.f);
                    dL_dalpha += (-T_final * inv_one_m_a) * bg_dot;
                    // rejected lanes contribute nothing (and exp2 of a positive power may be inf)
                    const float Gs = active ? G : 0.f;
                    const float gda = Gs * dL_dalpha;        // G * dL_dalpha
                    const float q = co.y * gda;              // G * dL_dG
                    // dG/ddelx = -G*(dx*cx + dy*cy) = G*(2A*dx + B*dy)/log2e  (backward.cu:648-651)
                    v[0] = q * fmaf(2.0f * g.z, dx, g.w * dy) * kx;
                    v[1] = q * fmaf(2.0f * co.x, dy, g.w * dx) * ky;
                    const float hq = -0.5f * q;
                    v[2] = hq * dxx;
                    v[3] = hq * dxy;
                    v[4] = hq * dyy;
                    v[5] = fmaf(w, dpo, gda);
                    if (active) {
                        Rb = Rn;
                        last_h = h;
                        last_alpha = alpha;
                    }
                    mul_packed<KC>(w, dsem, v + 10);
                } else {
                    mul_packed<KC>(w, dsem, v);
                }
                if (active) T = test_T;

                const float total = wave_reduce_transpose<NV>(v, lane);
                if (a.debug_flags & 1) {
                    asm volatile("" ::"v"(total));  // timing experiment: keep the sum alive, drop the atomic
                } else if (tgt_base) {
                    atomicAdd(tgt_base + (size_t)s_id[j] * tgt_stride, total);
                }
            }
        }
    }
}

}  // namespace

// Which kernel family takes packed rows of K semantic channels: shared by the launcher below and by hsr_api.hip, which sizes the rows
// (the compact layout exists only in the Q-panel kernels).
static bool backward_takes_q(int Ksem)
{
    static const char* impl = getenv("HSR_BWD_IMPL");
    static const bool other = impl && (!strcmp(impl, "valu") || !strcmp(impl, "sub") || !strcmp(impl, "mfma") || !strcmp(impl, "mom"));
    return !other && Ksem <= 27;
}
int hsr_backward_row_layout(int K_semantic, bool packed, int P)
{
    // (beyond 2^30 row elements the all-VALU kernel takes over, with classic rows: same test as in the launcher)
    return (packed && backward_takes_q(K_semantic) && hsr_grow_compact_pays(K_semantic) &&
            (size_t)P * (size_t)hsr_grow_stride_l(1, K_semantic) < ((size_t)1 << 30)) ? 1 : 0;
}

int hsr_launch_render_backward(const RenderBwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles)), block(256);
    // Default (packed accumulation rows, a.grow): the matrix-core kernels on 4x4 sub-block lists (hsr_render_bwd_sub.hip) — `sub` for
    // K <= 27, one-pass `subw` beyond.  They address the packed rows with 32-bit element indices; beyond 2^30 row elements (P > 22 M
    // Gaussians at K = 26), in the legacy accumulation mode (no scratch: atomics straight into the six output arrays, like the
    // reference) and under HSR_BWD_IMPL=valu (A/B timing, tests) the all-VALU kernel of this file takes over: 64-bit addressing, any K.
    // Round 1's quadrant-list matrix-core kernels live in experiments/ (HSR_BWD_IMPL=mfma in the ablate build).
    const bool rows_fit_32bit = !a.grow || (size_t)a.P * (size_t)a.grow_stride < ((size_t)1 << 30);
    static const char* impl = getenv("HSR_BWD_IMPL");
    static const bool force_valu = impl && !strcmp(impl, "valu");
    const int Ksem = a.semantic ? a.K : 0;
#ifdef HSR_ABLATE
    static const bool use_quad = impl && !strcmp(impl, "mfma");
    static const bool use_mom = impl && !strcmp(impl, "mom");
    if (!force_valu && rows_fit_32bit) {
        if (use_mom && a.grow && Ksem <= 27) return hsr_launch_render_backward_mom(a, stream);
        if (use_quad) return Ksem <= 27 ? hsr_launch_render_backward_mfma(a, stream) : hsr_launch_render_backward_wide(a, stream);
    }
#endif
    static const bool old_sub = impl && !strcmp(impl, "sub");   // round 3's butterfly kernel (A/B timing, parity-tested)
    if (!force_valu && rows_fit_32bit && a.grow) {
        if (Ksem > 27) return hsr_launch_render_backward_subw(a, stream);
        return old_sub ? hsr_launch_render_backward_sub(a, stream) : hsr_launch_render_backward_q(a, stream);
    }
    if (!a.semantic || a.K == 0) {
        render_bwd_kernel<0, true><<<grid, block, 0, stream>>>(a, 0);
        return HSR_OK;
    }
    switch (a.K) {
    case 16: render_bwd_kernel<16, true><<<grid, block, 0, stream>>>(a, 0); break;
    case 26: render_bwd_kernel<26, true><<<grid, block, 0, stream>>>(a, 0); break;
    default:
        // base sums + first 32 channels, then 32-channel chunks (K = 74, 102 and any other K)
        render_bwd_kernel<32, true><<<grid, block, 0, stream>>>(a, 0);
        for (int c0 = 32; c0 < a.K; c0 += 32) render_bwd_kernel<32, false><<<grid, block, 0, stream>>>(a, c0);
        break;
    }
    return HSR_OK;
}
