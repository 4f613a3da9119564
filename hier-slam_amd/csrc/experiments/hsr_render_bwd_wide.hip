// hsr_render_bwd_wide.hip — backward tile kernel for WIDE semantic trees (K > 27: the reference's 74- and 102-channel
// configurations, config.h:18), per-splat sums on the matrix cores, in channel PASSES.
//
// Same per-pixel semantics as hsr_render_bwd.hip (reference backward.cu:669-899) and the same scheme as
// hsr_render_bwd_mfma.hip (which serves K <= 27 in one launch and whose header explains the layouts):
//        D[splat][channel] = W[splat][pixel] . G[pixel][channel]      on v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chain)
// with the per-wave LDS weight panel (16 splat slots x 64 pixels, row stride 66) as A operand and the upstream
// gradients G of the wave's 64 pixels in registers in B-operand layout.  G for 100+ channels does not fit in
// registers next to the rest of the kernel, so a wide tree is covered by passes of up to 64 channels (16*NG
// registers of G each):
//   BASE pass   : the 7 alpha-path moments (VALU butterfly) + r, g, b, depth, opacity + the first <= 52 semantic channels
//                 (7 + 52 + 5 = 64 = one lane per packed-row column at emission);
//   SEM passes  : up to 64 further semantic channels each; they only re-derive alpha and T (about 20 VALU instructions
//                 per splat) and feed the panel — the reference's separate atomics per channel (backward.cu:845)
//                 become one row-segment atomic per (wave, splat).
// K = 74: BASE(52) + SEM(22);  K = 102: BASE(52) + SEM(50).  Before: one all-VALU launch per 32 channels (3-4 launches
// of ~0.4 ms at 500k Gaussians, each with a 32-value butterfly per splat).
#include "hsr_tile_common.h"
#include "hsr_wave_reduce.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WF_SLOTS = 16;        // accepted splats per MFMA flush (M dimension)
constexpr int WF_STRIDE = 66;       // floats per panel row: slot*66 + pixel -> conflict-free transposed reads
constexpr int WF_PANEL = 64 * 17;   // floats per wave: max(16 * 66, 64 * 17 for the G transpose)
constexpr int WF_BASE_SEM = 52;     // semantic channels of the BASE pass
constexpr int WF_SEM_SEM = 64;      // semantic channels of a SEM pass

// semantic channels [c0, c0 + ns) of the image; BASE adds the ten base sums.  16*NG >= ns + (BASE ? 5 : 0).
template <int NG, bool BASE>
__global__ void __launch_bounds__(256, 2) render_bwd_wide_kernel(RenderBwdArgs a, int c0, int ns)
{
    constexpr int BATCH = 256;
    constexpr int NV = 7;  // VALU butterfly: mean2D.xy, conic.xyw, opacity(alpha path), depth(median)
    __shared__ float4 s_geo[BATCH];
    __shared__ float2 s_co[BATCH];
    __shared__ float4 s_col[BATCH];
    __shared__ int s_id[BATCH];
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ int s_wmax[4];
    __shared__ float s_panel[4][WF_PANEL];
    __shared__ int s_slot_id[4][WF_SLOTS];
    __shared__ int s_prev_id[4][WF_SLOTS];
    __shared__ float s_u7[4][WF_SLOTS * 8];

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
    const int nch = ns + (BASE ? 5 : 0);  // live columns of D

    // prologue loads: unconditional (out-of-image lanes read pixel 0 and are zeroed afterwards), all issued first
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

    // ---- the MFMA B operand: G transposed through LDS, one 16-channel group at a time ----
    float Breg[NG][16];
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
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) Breg[g][m] = panel[(4 * m + (lane >> 4)) * 17 + (lane & 15)];
        __syncthreads();
    }
    const int hi_all = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    const float bg_dot = a.bg[0] * dpx0 + a.bg[1] * dpx1 + a.bg[2] * dpx2;
    const float kx = (0.5f * a.W) / HSR_LOG2E, ky = (0.5f * a.H) / HSR_LOG2E;
    float Rb = 0.f, last_h = 0.f, last_alpha = 0.f;

    const bool packed = a.grow != nullptr;
    // legacy arrays: atomic target of the butterfly value this lane ends up holding (BASE pass)
    const int myv = reduce_slot(lane);
    float* tgt_base = nullptr;
    int tgt_stride = 0;
    if (BASE && !packed) {
        if (myv < 2) { tgt_base = a.dL_dmean2D + myv; tgt_stride = 3; }
        else if (myv < 5) { tgt_base = a.dL_dconic + (myv == 4 ? 3 : myv - 2); tgt_stride = 4; }
        else if (myv == 5) { tgt_base = a.dL_dopacity; tgt_stride = 1; }
        else if (myv == 6) { tgt_base = a.dL_ddepth; tgt_stride = 1; }
    }
    // packed mode: a finished splat's sums are parked in its panel row at index p and leave as ONE atomic
    // wave-instruction, lane p -> packed-row column emit_col:  BASE: p 0..6 moments | 7..7+ns-1 semantics | then the 5
    // direct sums;  SEM: p = channel
    constexpr int P0 = BASE ? 7 : 0;
    int emit_col = -1;
    if (packed) {
        if (BASE && lane < 7) emit_col = lane;
        else if (lane >= P0 && lane < P0 + ns) emit_col = HSR_GROW_SEM0 + c0 + (lane - P0);
        else if (BASE && lane >= P0 + ns && lane < P0 + ns + 5) emit_col = hsr_grow_direct0(a.K) + (lane - P0 - ns);
    }
    int nslot = 0;   // wave-uniform: accepted splats waiting in the panel
    int prev_n = 0;  // wave-uniform: finished rows of the previous group still parked in the panel (packed mode)

    auto emit_row = [&](int srow) {
        const float val = panel[srow * WF_STRIDE + lane];
        if (emit_col >= 0 && !(a.debug_flags & 1))
            atomicAdd(a.grow + ((uint32_t)s_prev_id[wv][srow] * (uint32_t)a.grow_stride + (uint32_t)emit_col), val);   // 32-bit index: launcher guards P * stride < 2^30
    };
    auto flush = [&]() {
        f32x4 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* arow = panel + (lane & 15) * WF_STRIDE + (lane >> 4);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const float av = arow[4 * m];
#pragma unroll
            for (int g = 0; g < NG; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Breg[g][m], acc[g], 0, 0, 0);
        }
        // D[row = 4*(lane>>4) + r][col = lane&15]: row = panel slot, col = channel within the group
        if (packed) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int slot = 4 * (lane >> 4) + r;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const int ch = 16 * g + (lane & 15);
                    if (ch < nch) panel[slot * WF_STRIDE + P0 + ch] = acc[g][r];
                }
            }
            if (BASE) {
#pragma unroll
                for (int h = 0; h < 2; h++) {  // the 7 butterfly sums of each slot -> indices 0..6
                    const int slot = 8 * h + (lane >> 3), vv = lane & 7;
                    if (vv < 7) panel[slot * WF_STRIDE + vv] = s_u7[wv][slot * 8 + vv];
                }
            }
            if (lane < WF_SLOTS) s_prev_id[wv][lane] = s_slot_id[wv][lane];
            prev_n = nslot;
            nslot = 0;
            return;
        }
        // legacy arrays: one atomic per (slot, channel) straight into the reference's arrays
        if (!(a.debug_flags & 1)) {
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const int ch = 16 * g + (lane & 15);
                float* base = nullptr;
                int stride = 0;
                if (ch < ns) { base = a.dL_dsemantics + c0 + ch; stride = a.K; }
                else if (BASE && ch < ns + 3) { base = a.dL_dcolor + (ch - ns); stride = 3; }
                else if (BASE && ch == ns + 3) { base = a.dL_ddepth; stride = 1; }
                else if (BASE && ch == ns + 4) { base = a.dL_dopacity; stride = 1; }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int slot = 4 * (lane >> 4) + r;
                    if (base && slot < nslot) atomicAdd(base + (size_t)s_slot_id[wv][slot] * stride, acc[g][r]);
                }
            }
        }
        nslot = 0;
    };

    // ---- software-pipelined staging, as in hsr_render_bwd_mfma.hip ----
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
        if (hi - cnt < wmax) {  // else: this wave's pixels all stopped in front of this batch
            for (int seg = 0; seg < 4; seg++) {
                const int m = s_lcnt[wv][seg];
                int j_next = s_list[wv][seg * 64];
                for (int k = 0; k < m; k++) {
                    const int j = j_next;   // slot fetched one iteration ahead (slot -> record are dependent LDS round trips)
                    j_next = s_list[wv][seg * 64 + min(k + 1, 63)];
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
                    // wave-uniform counters pinned to scalars: the panel / row addressing then runs on the scalar unit
                    nslot = __builtin_amdgcn_readfirstlane(nslot);
                    prev_n = __builtin_amdgcn_readfirstlane(prev_n);
                    // packed mode: the previous group's row parked in this panel row leaves now
                    if (nslot < prev_n) emit_row(nslot);
                    panel[nslot * WF_STRIDE + lane] = w;
                    s_slot_id[wv][nslot] = s_id[j];

                    if (BASE) {
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
                        v[6] = (active && pos == median_at) ? dpm : 0.f;
                        if (active) {
                            Rb = Rn;
                            last_h = h;
                            last_alpha = alpha;
                        }
                        const float total = wave_reduce_transpose<NV>(v, lane);
                        if (packed) {
                            if (myv < NV) s_u7[wv][nslot * 8 + myv] = total;  // joins its row at the flush
                        } else if (tgt_base && !(a.debug_flags & 1)) {
                            atomicAdd(tgt_base + (size_t)s_id[j] * tgt_stride, total);
                        }
                    }
                    if (active) T = test_T;
                    nslot++;
                    if (nslot == WF_SLOTS) flush();
                }
            }
        }
    }
    if (packed) {
        for (int sr = nslot; sr < prev_n; sr++) emit_row(sr);  // rows of the previous group not displaced yet
        prev_n = 0;
        if (nslot > 0) {
            flush();
            for (int sr = 0; sr < prev_n; sr++) emit_row(sr);
        }
    } else if (nslot > 0) {
        flush();
    }
}

template <bool BASE>
void launch_pass(const RenderBwdArgs& a, int c0, int ns, dim3 grid, hipStream_t stream)
{
    const int groups = (ns + (BASE ? 5 : 0) + 15) / 16;
    const dim3 block(256);
    if (groups <= 1) render_bwd_wide_kernel<1, BASE><<<grid, block, 0, stream>>>(a, c0, ns);
    else if (groups == 2) render_bwd_wide_kernel<2, BASE><<<grid, block, 0, stream>>>(a, c0, ns);
    else if (groups == 3) render_bwd_wide_kernel<3, BASE><<<grid, block, 0, stream>>>(a, c0, ns);
    else render_bwd_wide_kernel<4, BASE><<<grid, block, 0, stream>>>(a, c0, ns);
}

}  // namespace

// semantic variant with K > 27 (any K): BASE pass + as many SEM passes as the width needs
int hsr_launch_render_backward_wide(const RenderBwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles));
    const int K = a.K;
    const int first = K < WF_BASE_SEM ? K : WF_BASE_SEM;
    launch_pass<true>(a, 0, first, grid, stream);
    for (int c0 = first; c0 < K; c0 += WF_SEM_SEM) launch_pass<false>(a, c0, K - c0 < WF_SEM_SEM ? K - c0 : WF_SEM_SEM, grid, stream);
    return HSR_OK;
}
