// hsr_render_bwd_mfma.hip — backward tile kernel with the "direct" per-splat sums on the matrix cores.
//
// Same per-pixel semantics as hsr_render_bwd.hip (reference backward.cu:472-899, see that file's header).
// What changes is WHERE the per-splat sums over the 64 pixels of a wave are formed.  Of the 10+K values a
// splat receives from a wave, K+5 have the form
//        sum_pixels  w[splat][pixel] * g[pixel][channel]        w = alpha*T   (backward.cu:616, :622, :632, :845)
// with g the upstream gradient of the pixel (K semantic channels, r, g, b, depth, final opacity) — a dense
// contraction over pixels: D[splat][channel] = W[splat][pixel] . G[pixel][channel].  The tile kernels are
// VALU-issue bound (≈4 cycles per wave64 VALU instruction on gfx950), the MFMA pipe is idle, and
// v_mfma_f32_16x16x4_f32 is an exact fp32 fmaf chain (MI355X guide, "FP32-input MFMA"), so the contraction moves
// there without touching numerics policy:
//   * G is loaded once per tile in its natural lane = pixel layout, transposed through LDS into the MFMA
//     B-operand layout (lane l holds G[pixel 4m + (l>>4)][channel 16g + (l&15)]) and kept in 16*NG registers;
//   * each accepted splat writes its 64 weights to one row of a per-wave LDS panel (one ds_write_b32);
//   * every 16 accepted splats the panel is read back in A-operand layout (lane l: W[slot l&15][pixel 4m + (l>>4)],
//     row stride 66 floats = conflict-free) and 16*NG MFMAs produce D[16 splats][16*NG channels];
//   * D's lanes hold (4 splats x 16 channels) per register: one global_atomic_add_f32 wave-instruction per
//     register adds four 64-byte row segments — a well-shaped atomic.
// Only 7 values per splat stay on the VALU butterfly (mean2D.xy, conic.xyw, the alpha-path opacity term and the
// median-depth term), 28 instructions instead of ~105, and the K+5 multiplies w*g disappear.
#include "hsr_tile_common.h"
#include "hsr_wave_reduce.h"

#ifdef HSR_TRACE
// Diagnostic build only (make -C hier-slam_amd/csrc trace -> libhsr_rast_trace.so, tools/trace_bwd.py): per-wave cycle counts of
// the phases of this kernel, s_memtime deltas accumulated in registers and dumped at the end.  Never compiled into the product.
#define HSR_TRACE_SLOTS 8
__device__ unsigned long long g_hsr_trace[16384 * HSR_TRACE_SLOTS];
extern "C" int hsr_debug_read_trace(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hsr_trace), sizeof(unsigned long long) * (size_t)n);
}
#define TR_NOW() clock64()
#define TR_ADD(acc, t0) (acc) += (unsigned long long)(clock64() - (t0))
#else
#define TR_NOW() 0ll
#define TR_ADD(acc, t0) ((void)0)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_SLOTS = 16;        // accepted splats per MFMA flush (M dimension)
constexpr int MF_STRIDE = 66;       // floats per panel row: slot*66 + pixel -> conflict-free transposed reads
constexpr int MF_PANEL = 64 * 17;   // floats per wave: max(16 * 66, 64 * 17 for the G transpose)

// KC semantic channels [0, KC) are produced here together with the 10 base sums; channels >= a.K are masked.
template <int KC>
__global__ void __launch_bounds__(256, 4) render_bwd_mfma_kernel(RenderBwdArgs a)
{
    constexpr int BATCH = 256;
    constexpr int NCH = KC + 5;               // sem[KC], r, g, b, depth, opacity(direct)
    constexpr int NG = (NCH + 15) / 16;       // 16-channel groups
    static_assert(NG <= 2, "at most 32 direct channels per launch");
    constexpr int NV = 7;                     // VALU butterfly: mean2D.xy, conic.xyw, opacity(alpha path), depth(median)
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
    __shared__ float s_u7[4][MF_SLOTS * 8];

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
    unsigned long long tr_stage = 0, tr_loop = 0, tr_flush = 0, tr_emit = 0, tr_visits = 0, tr_accepted = 0;
    const long long tr_t0 = TR_NOW();
    (void)tr_stage; (void)tr_loop; (void)tr_flush; (void)tr_emit; (void)tr_visits; (void)tr_accepted; (void)tr_t0;

    // Every prologue load is unconditional (out-of-image lanes read pixel 0 and are zeroed afterwards) and
    // issued before anything consumes one: a per-lane guard makes hipcc branch around each load and wait for it
    // separately (38 serialised global round trips per wave otherwise).
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
    const long long tr_t1 = TR_NOW();   // end of the prologue
    (void)tr_t1;

    const float bg_dot = a.bg[0] * dpx0 + a.bg[1] * dpx1 + a.bg[2] * dpx2;
    const float kx = (0.5f * a.W) / HSR_LOG2E, ky = (0.5f * a.H) / HSR_LOG2E;
    float Rb = 0.f, last_h = 0.f, last_alpha = 0.f;

    // atomic target of the butterfly value this lane ends up holding
    const int myv = reduce_slot(lane);
    float* tgt_base = nullptr;
    int tgt_stride = 0;
    if (a.grow) {  // packed per-Gaussian row: the 7 butterfly values are columns 0..6 of one 64-byte line
        if (myv < 7) { tgt_base = a.grow + myv; tgt_stride = a.grow_stride; }
    } else if (myv < 2) { tgt_base = a.dL_dmean2D + myv; tgt_stride = 3; }
    else if (myv < 5) { tgt_base = a.dL_dconic + (myv == 4 ? 3 : myv - 2); tgt_stride = 4; }
    else if (myv == 5) { tgt_base = a.dL_dopacity; tgt_stride = 1; }
    else if (myv == 6) { tgt_base = a.dL_ddepth; tgt_stride = 1; }
    int nslot = 0;   // wave-uniform: accepted splats waiting in the panel
    int prev_n = 0;  // wave-uniform: finished rows of the previous group still parked in the panel (packed mode)
    const bool packed = a.grow != nullptr;
    // packed mode: which packed-row column this lane emits, if any (cols 0..6 | 16..16+K-1 | 16+K..16+K+4)
    const bool emit_lane = packed && lane < a.grow_stride && (lane < 7 || (lane >= HSR_GROW_SEM0 && lane < HSR_GROW_SEM0 + a.K + 5));

    // Packed mode, deferred emission: a finished group's 16 rows stay in the panel (row s in the 66-float panel
    // row s, which the MFMAs have already consumed) and row s is added to global memory — ONE atomic
    // wave-instruction covering the Gaussian's whole packed row — just before the next group's s-th splat
    // overwrites that panel row.  Atomics are thereby spaced one per accepted splat instead of bursts of 24.
    auto emit_row = [&](int srow) {
        const long long te = TR_NOW();
        (void)te;
        const float val = panel[srow * MF_STRIDE + lane];
        // 32-bit element index (host guarantees P * grow_stride < 2^30): scalar base + 32-bit vector offset instead of a 64-bit
        // multiply-add per emission
        if (emit_lane && !(a.debug_flags & 1))
            atomicAdd(a.grow + ((uint32_t)s_prev_id[wv][srow] * (uint32_t)a.grow_stride + (uint32_t)lane), val);
        if (emit_lane && (a.debug_flags & 8))   // timing experiment: twice the atomic requests (second one to a neighbouring row)
            atomicAdd(a.grow + ((uint32_t)(s_prev_id[wv][srow] ^ 1) * (uint32_t)a.grow_stride + (uint32_t)lane), val);
        if (emit_lane && (a.debug_flags & 16))  // three times
            atomicAdd(a.grow + ((uint32_t)(s_prev_id[wv][srow] ^ 2) * (uint32_t)a.grow_stride + (uint32_t)lane), val);
        TR_ADD(tr_emit, te);
    };
    auto flush = [&]() {
        const long long tf = TR_NOW();
        (void)tf;
        if (a.debug_flags & 2) {  // timing experiment: no MFMA / no flush atomics
            nslot = 0;
            return;
        }
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
        if (packed) {
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
            for (int h = 0; h < 2; h++) {  // the 7 butterfly sums of each slot -> columns 0..6
                const int slot = 8 * h + (lane >> 3), vv = lane & 7;
                if (vv < 7) panel[slot * MF_STRIDE + vv] = s_u7[wv][slot * 8 + vv];
            }
            if (lane < MF_SLOTS) s_prev_id[wv][lane] = s_slot_id[wv][lane];
            prev_n = nslot;
            nslot = 0;
            TR_ADD(tr_flush, tf);
            return;
        }
        // legacy arrays: atomic targets of the columns this lane holds, recomputed once per 16 splats
        float* mf_base[NG];
        int mf_stride[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int ch = 16 * g + (lane & 15);
            mf_base[g] = nullptr;
            mf_stride[g] = 0;
            if (ch < KC) { if (ch < a.K) { mf_base[g] = a.dL_dsemantics + ch; mf_stride[g] = a.K; } }
            else if (ch < KC + 3) { mf_base[g] = a.dL_dcolor + (ch - KC); mf_stride[g] = 3; }
            else if (ch == KC + 3) { mf_base[g] = a.dL_ddepth; mf_stride[g] = 1; }
            else if (ch == KC + 4) { mf_base[g] = a.dL_dopacity; mf_stride[g] = 1; }
        }
        if (!(a.debug_flags & 1)) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int slot = 4 * (lane >> 4) + r;
                if (slot < nslot) {
                    const size_t id = (size_t)s_slot_id[wv][slot];
#pragma unroll
                    for (int g = 0; g < NG; g++)
                        if (mf_base[g]) atomicAdd(mf_base[g] + id * mf_stride[g], acc[g][r]);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < NG; g++) asm volatile("" ::"v"(acc[g]));
        }
        nslot = 0;
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
        const long long ts = TR_NOW();
        (void)ts;
        __syncthreads();
        uint32_t qmask = 0u;
        if (t < cnt) {
            qmask = quadrant_mask_exact(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0);
            s_id[t] = id_cur;
            s_geo[t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_co[t] = make_float2((-0.5f * HSR_LOG2E) * p_co.z, p_co.w);
            s_col[t] = make_float4(p_r, p_g, p_b, p_d);
        }
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        __syncthreads();
        load_record(hi - BATCH);
        load_id(hi - 2 * BATCH);
        TR_ADD(tr_stage, ts);
        const long long tl = TR_NOW();
        (void)tl;
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
                    const float dxx = dx * dx, dxy = dx * dy, dyy = dy * dy;
                    const float power2 = fmaf(co.x, dyy, fmaf(g.w, dxy, g.z * dxx));
                    const float G = __builtin_amdgcn_exp2f(power2);
                    const float alpha = fminf(0.99f, co.y * G);
                    const bool active = pos < last_contributor && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
#ifdef HSR_TRACE
                    tr_visits++;
#endif
                    if (__ballot(active) == 0ull) continue;
#ifdef HSR_TRACE
                    tr_accepted++;
#endif

                    const float inv_one_m_a = __builtin_amdgcn_rcpf(1.0f - alpha);
                    const float test_T = T * inv_one_m_a;
                    const float w = active ? alpha * test_T : 0.f;
                    // nslot / prev_n are wave-uniform but live in vector registers (they change under divergent-looking
                    // control flow): pin them to scalars here so that the panel / row addressing runs on the scalar unit —
                    // the blend loop is VALU-port bound
                    nslot = __builtin_amdgcn_readfirstlane(nslot);
                    prev_n = __builtin_amdgcn_readfirstlane(prev_n);
                    // packed mode: the previous group's row parked in this panel row leaves now
                    if (nslot < prev_n) emit_row(nslot);
                    // direct sums go through the panel -> MFMA
                    panel[nslot * MF_STRIDE + lane] = w;
                    s_slot_id[wv][nslot] = s_id[j];   // every lane stores the same value to the same address: no exec juggling

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
                        T = test_T;
                    }
                    if (a.debug_flags & 4) {  // timing experiment: no butterfly, no per-splat atomic
#pragma unroll
                        for (int c = 0; c < NV; c++) asm volatile("" ::"v"(v[c]));
                        nslot++;
                        if (nslot == MF_SLOTS) flush();
                        continue;
                    }
                    const float total = wave_reduce_transpose<NV>(v, lane);
                    if (packed) {
                        if (myv < NV) s_u7[wv][nslot * 8 + myv] = total;  // joins its row at the flush
                    } else if (a.debug_flags & 1) {
                        asm volatile("" ::"v"(total));
                    } else if (tgt_base) {
                        atomicAdd(tgt_base + (size_t)s_id[j] * tgt_stride, total);
                    }
                    nslot++;
                    if (nslot == MF_SLOTS) flush();
                }
            }
        }
        TR_ADD(tr_loop, tl);
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
#ifdef HSR_TRACE
    if (lane == 0) {
        const int wid = tile * 4 + wv;
        if (wid < 16384) {
            unsigned long long* o = g_hsr_trace + (size_t)wid * HSR_TRACE_SLOTS;
            o[0] = (unsigned long long)(clock64() - tr_t0);
            o[1] = (unsigned long long)(tr_t1 - tr_t0);
            o[2] = tr_stage; o[3] = tr_loop; o[4] = tr_flush; o[5] = tr_emit; o[6] = tr_visits; o[7] = tr_accepted;
        }
    }
#endif
}

}  // namespace

// K <= 27 is covered in one launch; for larger K the caller adds VALU chunk launches for channels >= 27.
int hsr_launch_render_backward_mfma(const RenderBwdArgs& a, hipStream_t stream)
{
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles)), block(256);
    const int K = a.semantic ? a.K : 0;
    if (K == 0) render_bwd_mfma_kernel<0><<<grid, block, 0, stream>>>(a);
    else if (K <= 11) render_bwd_mfma_kernel<11><<<grid, block, 0, stream>>>(a);   // one 16-channel group
    else if (K == 16) render_bwd_mfma_kernel<16><<<grid, block, 0, stream>>>(a);
    else if (K == 26) render_bwd_mfma_kernel<26><<<grid, block, 0, stream>>>(a);
    else render_bwd_mfma_kernel<27><<<grid, block, 0, stream>>>(a);
    return HSR_OK;
}
