// hsr_render_bwd_mom.hip — backward tile kernel, K <= 27, packed rows: the six alpha-path moments on the matrix cores too.
//
// hsr_render_bwd_mfma.hip forms K+5 of the 10+K per-splat sums as D = W . G on the matrix cores and leaves 7 on a VALU
// butterfly (28 instructions per accepted splat) fed by ~12 instructions of per-pixel products.  Six of those seven are
// polynomial moments of ONE per-pixel weight q = opacity * G * dL/dalpha over the wave's 64 pixels:
//     sum q*dx, sum q*dy, sum q*dx^2, sum q*dx*dy, sum q*dy^2, sum q        with  dx = a - x,  dy = b - y
// (a, b = splat centre relative to the quadrant, x, y = 0..7 the pixel inside it), i.e. linear combinations — with
// per-SPLAT coefficients — of   S[m] = sum_pixels q[pixel] * mono_m(x, y),   mono = {1, x, y, x^2, xy, y^2}:
// one more dense contraction over pixels, with a CONSTANT right-hand side.  Here every accepted splat also writes its 64
// q values to a second, 8-row LDS panel; every 8 splats 16 v_mfma_f32_16x16x4_f32 produce their S[m] (B operand generated
// on the fly from the lane index), and one lane per (splat, moment) combines them with the splat's own coefficients.
// Quadrant-local coordinates keep the expansion well conditioned (|a|, |b| <~ 20: two digits of cancellation at most).
// The seventh sum (median depth: one pixel -> one splat, ever) leaves as one atomic per pixel at the end.
// W . G keeps its 16-splat groups (2 MFMAs per splat); the moments add 2 per splat.  Packed accumulation mode only.
#include "hsr_tile_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_SLOTS = 16;        // accepted splats per MFMA flush (M dimension)
constexpr int MF_STRIDE = 66;       // floats per panel row: slot*66 + pixel -> conflict-free transposed reads
constexpr int MF_PANEL = 64 * 17;   // floats per wave: max(16 * 66, 64 * 17 for the G transpose)

// KC semantic channels [0, KC) are produced here together with the 10 base sums; channels >= a.K are masked.
template <int KC>
__global__ void __launch_bounds__(256, 4) render_bwd_mom_kernel(RenderBwdArgs a)
{
    constexpr int BATCH = 256;
    constexpr int NCH = KC + 5;               // sem[KC], r, g, b, depth, opacity(direct)
    constexpr int NG = (NCH + 15) / 16;       // 16-channel groups
    static_assert(NG <= 2, "at most 32 direct channels per launch");
    constexpr int QS = 8;                     // splats per moment flush (rows of the q panel)
    __shared__ float4 s_geo[BATCH];
    __shared__ float2 s_co[BATCH];
    __shared__ float4 s_col[BATCH];
    __shared__ int s_id[BATCH];
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ int s_wmax[4];
    __shared__ float s_panel[4][MF_PANEL];
    __shared__ int s_slot_id[4][MF_SLOTS];
    __shared__ int s_prev_id[4][MF_SLOTS];
    __shared__ float s_u7[4][MF_SLOTS * 6];      // the six finished moments of each slot (joins its row at the W flush)
    __shared__ float s_qpanel[4][QS * MF_STRIDE];
    __shared__ int s_slot_j[4][MF_SLOTS];        // batch slot of each panel slot: its geometry record, read at the moment flush

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

    // Every prologue load is unconditional (out-of-image lanes read pixel 0 and are zeroed afterwards) and
    // issued before anything consumes one: a per-lane guard makes hipcc branch around each load and wait for it
    // separately (38 serialised global round trips per wave otherwise).
    const size_t pix_ld = inside ? pix_id : 0;
    const float inm = inside ? 1.f : 0.f;
    const float T_final_ld = a.final_T[pix_ld];
    const int last_contributor_ld = (int)a.n_contrib[pix_ld];
    float dpx0 = a.dL_dpix[pix_ld], dpx1 = a.dL_dpix[N + pix_ld], dpx2 = a.dL_dpix[2 * N + pix_ld];
    float dpd = a.dL_dpix_depth[pix_ld], dpm = a.dL_dpix_median[pix_ld], dpo = a.dL_dpix_opacity[pix_ld];
    float semv[KC > 0 ? KC : 1];
#pragma unroll
    for (int c = 0; c < KC; c++) semv[c] = a.dL_dpix_sem[(size_t)min(c, a.K - 1) * N + pix_ld];
    dpx0 *= inm; dpx1 *= inm; dpx2 *= inm; dpd *= inm; dpm *= inm; dpo *= inm;
    const float T_final = T_final_ld * inm;
    float T = T_final;
    const int last_contributor = inside ? last_contributor_ld : 0;

    int wmax = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    if (lane == 0) s_wmax[wv] = wmax;

    // ---- the MFMA B operand: G transposed through LDS ----
    float Breg[NG][16];
    {
        // issue every global load of this lane's upstream gradients first (independent, one wait), THEN go through
        // LDS: a load -> ds_write pair per channel would serialise 32 global round trips in the prologue
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
#pragma unroll
        for (int g = 0; g < NG; g++) {
            // channels [16g, 16g+16) of this lane's pixel -> panel[pixel][c] (row stride 17), read back transposed
#pragma unroll
            for (int c = 0; c < 16; c++) panel[lane * 17 + c] = gv[g][c];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) Breg[g][m] = panel[(4 * m + (lane >> 4)) * 17 + (lane & 15)];
            __syncthreads();
        }
    }
    const int hi_all = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    const float bg_dot = a.bg[0] * dpx0 + a.bg[1] * dpx1 + a.bg[2] * dpx2;
    const float kx = (0.5f * a.W) / HSR_LOG2E, ky = (0.5f * a.H) / HSR_LOG2E;
    float Rb = 0.f, last_h = 0.f, last_alpha = 0.f;

    int nslot = 0;   // wave-uniform: accepted splats waiting in the panel
    int prev_n = 0;  // wave-uniform: finished rows of the previous group still parked in the panel (packed mode)
    constexpr bool packed = true;
    float* qpanel = s_qpanel[wv];
    int q_lo = 0;       // wave-uniform: first slot of the current group whose moments are not finished yet
    int med_id = -1;    // the splat at which this pixel's transmittance crossed 0.5 (median depth): at most one
    // packed mode: which packed-row column this lane emits, if any (cols 0..6 | 16..16+K-1 | 16+K..16+K+4)
    const bool emit_lane = lane < a.grow_stride && (lane < 6 || (lane >= HSR_GROW_SEM0 && lane < HSR_GROW_SEM0 + a.K + 5));
    // B operand of the moment MFMAs, generated on the fly: lane l supplies B[k = pixel 4m + (l>>4)][n = l&15] = mono_n(x, y)
    // with x = 4*(m&1) + (l>>4), y = m>>1 (quadrant-local), mono = {1, x, y, x^2, x*y, y^2, 0...}
    const int mono = lane & 15;
    const float x_even = (float)(lane >> 4), x_odd = (float)(4 + (lane >> 4));
    const float e1 = mono == 0 ? 1.f : 0.f, ex = mono == 1 ? 1.f : 0.f, ey = mono == 2 ? 1.f : 0.f;
    const float exx = mono == 3 ? 1.f : 0.f, exy = mono == 4 ? 1.f : 0.f, eyy = mono == 5 ? 1.f : 0.f;
    const float P_even = fmaf(exx * x_even, x_even, fmaf(ex, x_even, e1)), Q_even = fmaf(exy, x_even, ey);
    const float P_odd = fmaf(exx * x_odd, x_odd, fmaf(ex, x_odd, e1)), Q_odd = fmaf(exy, x_odd, ey);
    // moments of slots [q_lo, nslot): 16 MFMAs over the q panel, then one lane per (slot, moment)
    auto flush_moments = [&]() {
        if (q_lo >= nslot) return;
        f32x4 accm = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* arow = qpanel + (lane & 7) * MF_STRIDE + (lane >> 4);   // rows 8-15 mirror rows 0-7 (unused)
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const float yv = (float)(m >> 1);
            const float bm = fmaf(yv, (m & 1) ? Q_odd : Q_even, (m & 1) ? P_odd : P_even) + eyy * (yv * yv);
            accm = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * m], bm, accm, 0, 0, 0);
        }
        // D[row = 4*(lane>>4) + r][col = lane&15]: q-panel rows 0-7 live in lanes 0-31
        if (lane < 32 && (lane & 15) < 6) {
#pragma unroll
            // S[m] of q row i goes into columns 0-5 of that (now consumed) q-panel row
            for (int r = 0; r < 4; r++) qpanel[(4 * (lane >> 4) + r) * MF_STRIDE + (lane & 15)] = accm[r];
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int qrow = lane >> 3, out = lane & 7;          // 8 q rows x 8 lanes
            const int slot = (q_lo & ~(QS - 1)) + qrow;          // panel slot of this q row
            if (slot >= q_lo && slot < nslot && out < 6) {
                const int js = s_slot_j[wv][slot];
                const float4 g = s_geo[js];
                const float2 co = s_co[js];
                const float* S = qpanel + qrow * MF_STRIDE;
                const float S0 = S[0], Sx = S[1], Sy = S[2], Sxx = S[3], Sxy = S[4], Syy = S[5];
                const float av = g.x - tg.qx0, bv = g.y - tg.qy0;   // splat centre relative to the quadrant origin
                const float qdx = fmaf(av, S0, -Sx), qdy = fmaf(bv, S0, -Sy);   // sum q*dx, sum q*dy
                float v;
                if (out == 0) v = fmaf(2.0f * g.z, qdx, g.w * qdy) * kx;
                else if (out == 1) v = fmaf(2.0f * co.x, qdy, g.w * qdx) * ky;
                else if (out == 2) v = -0.5f * fmaf(av, fmaf(av, S0, -2.0f * Sx), Sxx);
                else if (out == 3) v = -0.5f * fmaf(av, fmaf(bv, S0, -Sy), fmaf(-bv, Sx, Sxy));
                else if (out == 4) v = -0.5f * fmaf(bv, fmaf(bv, S0, -2.0f * Sy), Syy);
                else v = S0 / co.y;                                  // sum G*dL_dalpha
                s_u7[wv][slot * 6 + out] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        q_lo = nslot;
    };

    // Packed mode, deferred emission: a finished group's 16 rows stay in the panel (row s in the 66-float panel
    // row s, which the MFMAs have already consumed) and row s is added to global memory — ONE atomic
    // wave-instruction covering the Gaussian's whole packed row — just before the next group's s-th splat
    // overwrites that panel row.  Atomics are thereby spaced one per accepted splat instead of bursts of 24.
    auto emit_row = [&](int srow) {
        const float val = panel[srow * MF_STRIDE + lane];
        if (emit_lane && !(a.debug_flags & 1))
            atomicAdd(a.grow + ((uint32_t)s_prev_id[wv][srow] * (uint32_t)a.grow_stride + (uint32_t)lane), val);   // 32-bit index: launcher guards P * stride < 2^30
    };
    auto flush = [&]() {
        flush_moments();   // slots of this group whose moments are still pending
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
        // park the finished rows in the panel, in packed-row column order
        int colg[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int ch = 16 * g + (lane & 15);
            colg[g] = ch < KC ? (ch < a.K ? HSR_GROW_SEM0 + ch : -1) : (ch < KC + 5 ? hsr_grow_direct0(a.K) + (ch - KC) : -1);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int slot = 4 * (lane >> 4) + r;
#pragma unroll
            for (int g = 0; g < NG; g++)
                if (colg[g] >= 0) panel[slot * MF_STRIDE + colg[g]] = acc[g][r];
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {  // the 6 moment sums of each slot -> columns 0..5
            const int slot = 8 * h + (lane >> 3), vv = lane & 7;
            if (vv < 6) panel[slot * MF_STRIDE + vv] = s_u7[wv][slot * 6 + vv];
        }
        if (lane < MF_SLOTS) s_prev_id[wv][lane] = s_slot_id[wv][lane];
        prev_n = nslot;
        nslot = 0;
        q_lo = 0;
    };

    // ---- software-pipelined staging, as in hsr_render_bwd.hip ----
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
            p_r = a.colors[3 * id];
            p_g = a.colors[3 * id + 1];
            p_b = a.colors[3 * id + 2];
            p_d = a.depths[id];
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
            qmask = quadrant_mask(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0);
            s_id[t] = id_cur;
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
                int j_next = s_list[wv][seg * 64];
                for (int k = 0; k < m; k++) {
                    // the slot of the NEXT entry is fetched one iteration ahead: slot -> record is otherwise two dependent
                    // LDS round trips per entry
                    const int j = j_next;
                    j_next = s_list[wv][seg * 64 + min(k + 1, 63)];
                    const float4 g = s_geo[j];
                    const float2 co = s_co[j];
                    // colour/depth of the splat fetched together with its geometry: read after the "anyone active?" branch
                    // it would be a third dependent LDS round trip per accepted splat
                    const float4 cd = s_col[j];
                    asm volatile("" ::"v"(cd.x), "v"(cd.y), "v"(cd.z), "v"(cd.w));
                    const int pos = hi - 1 - j;
                    const float dx = g.x - pfx, dy = g.y - pfy;
                    const float power2 = fmaf(co.x, dy * dy, fmaf(g.w, dx * dy, g.z * (dx * dx)));
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
                    const float h = fmaf(cd.x, dpx0, fmaf(cd.y, dpx1, fmaf(cd.z, dpx2, fmaf(cd.w, dpd, dpo))));
                    const float Rn = fmaf(last_alpha, last_h - Rb, Rb);
                    float dL_dalpha = (h - Rn) * test_T;
                    dL_dalpha += (-T_final * inv_one_m_a) * bg_dot;
                    const float q = active ? co.y * (G * dL_dalpha) : 0.f;
                    // both weights of the splat go to their panels: w -> D = W.G, q -> the six moments
                    panel[nslot * MF_STRIDE + lane] = w;
                    qpanel[(nslot & (QS - 1)) * MF_STRIDE + lane] = q;
                    s_slot_id[wv][nslot] = s_id[j];
                    s_slot_j[wv][nslot] = j;
                    if (active) {
                        if (test_T > 0.5f && T < 0.5f) med_id = s_id[j];
                        Rb = Rn;
                        last_h = h;
                        last_alpha = alpha;
                        T = test_T;
                    }
                    nslot++;
                    if (nslot == MF_SLOTS) flush();
                    else if ((nslot & (QS - 1)) == 0) flush_moments();
                }
            }
            // the finishing lanes read this batch's geometry records: moments may not wait across a restaging
            flush_moments();
        }
    }
    for (int sr = nslot; sr < prev_n; sr++) emit_row(sr);  // rows of the previous group not displaced yet
    prev_n = 0;
    if (nslot > 0) {
        flush();
        for (int sr = 0; sr < prev_n; sr++) emit_row(sr);
    }
    // median-depth term (reference backward.cu:853-857): one pixel -> one splat -> one atomic
    if (med_id >= 0 && dpm != 0.f && !(a.debug_flags & 1)) atomicAdd(a.grow + (size_t)med_id * a.grow_stride + 6, dpm);
}

}  // namespace

// semantic K <= 27 / non-semantic, packed accumulation mode only
int hsr_launch_render_backward_mom(const RenderBwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles)), block(256);
    const int K = a.semantic ? a.K : 0;
    if (K == 0) render_bwd_mom_kernel<0><<<grid, block, 0, stream>>>(a);
    else if (K <= 11) render_bwd_mom_kernel<11><<<grid, block, 0, stream>>>(a);   // one 16-channel group
    else if (K == 16) render_bwd_mom_kernel<16><<<grid, block, 0, stream>>>(a);
    else if (K == 26) render_bwd_mom_kernel<26><<<grid, block, 0, stream>>>(a);
    else render_bwd_mom_kernel<27><<<grid, block, 0, stream>>>(a);
    return HSR_OK;
}
