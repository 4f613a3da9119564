// (Since late round 2 the per-lane kernel of hsr_render_fwd.hip — sub-block lists, rows touched into L2, quad-shared row reads — is faster at
// every width: this kernel runs only under HSR_FWD_IMPL=wide, as a parity-tested second implementation.)
// hsr_render_fwd_wide.hip — forward tile kernel for WIDE semantic trees (29 <= K <= 124: the reference's 74- and
// 102-channel configurations, config.h:18), blend accumulation on the matrix cores.
//
// Per-pixel semantics are those of hsr_render_fwd.hip (reference forward.cu:400-538).  Why a separate kernel: with
// per-lane accumulators the K = 74 / 102 instantiations need 222 / 256+ registers (2 / 1 waves per SIMD) and 19 / 26
// broadcast ds_read_b128 per (wave, splat) — each costs ~4.5 LDS clocks on gfx950 whatever the address pattern
// (tools/micro/lds_bw.hip), so the blend loop is LDS-bound at ~120 clocks per splat.  Here
//        OUT[64 px][32*NB ch] += W[64 px][2 splats] . F[2 splats][32*NB ch]
// runs as 2*NB v_mfma_f32_32x32x2_f32 per PAIR of accepted splats (an exact fp32 fmaf chain, so numerics are those
// of the VALU kernel): the A operand is the two per-lane weights after one v_permlane32_swap, the B operand is ONE
// conflict-free ds_read_b32 per lane per 32-channel block.  The accumulators (32*NB registers) sit in the MFMA
// result registers, the VALU only evaluates alpha / T / termination.
// LDS: feature rows of one batch, BATCH x 32*NB floats (BATCH = 128: 48 KB at NB = 3, 64 KB at NB = 4 -> 2-3
// workgroups per CU, which is also what the register budget allows).  Semantic rows are gathered one row per wave-level load.
#include "hsr_tile_common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned uint2w_ __attribute__((ext_vector_type(2)));

template <int NB>
__global__ void __launch_bounds__(256, 2) render_fwd_wide_kernel(RenderFwdArgs a)
{
    constexpr int BATCH = 128;
    constexpr int FM = 32 * NB;        // feature row: sem[K], r, g, b, depth, zero padding
    constexpr int FS = FM + 4;         // row stride in LDS: +4 floats so that the 16-byte staging stores of consecutive rows
                                       // hit different banks (stride FM put all 64 lanes on the same 4 banks: LDS was 54 %
                                       // bank-conflict cycles and the kernel LDS-bound)
    constexpr int ROWS_PER_ROUND = 16; // semantic rows in flight per wave (one float2 register each)
    __shared__ float4 s_geo[BATCH];    // x, y, A, B (pre-scaled conic)
    __shared__ float2 s_co[BATCH];     // C, opacity
    __shared__ float s_feat[BATCH * FS > 4 * 64 * 33 ? BATCH * FS : 4 * 64 * 33];  // feature rows; output transpose at the end
    __shared__ int s_ids[BATCH];
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ int s_wdone[4];

    const int K = a.K;
    const int tile = hsr_block_tile(blockIdx.x, ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y));
    if (tile >= ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (a.bin.base) {   // speculative forward: the list lives where num_rendered says
        BinState bs;
        if (!hsr_bin_resolve(a.bin, *a.bin.R_dev, &bs)) {
            hsr_poison_tile(a, tile, t, true, 0, a.K);
            return;
        }
        a.point_list = bs.vals;
        a.masks = bs.vals_unsorted;
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
    uint32_t median_at = 0;   // 1 + list position of the splat at which T crossed 0.5 (ImgState::median_pos)
    bool done = !inside;
    f32x16 D[2][NB];  // OUT[pixels 0-31 | 32-63][channel block]
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) D[h][b][i] = 0.f;
    float pend_w = 0.f;  // weights of an accepted splat waiting for a partner
    int pend_j = -1;     // its batch slot (wave-uniform)

    // staging: thread t < 128 owns entry e = t of the batch (geometry, colour, depth, culling mask); the K-float semantic
    // rows are gathered COOPERATIVELY: wave w copies rows 32w .. 32w+31, one row per load instruction with lane l on
    // floats 2l, 2l+1 — one contiguous 4K-byte request per row instead of 64 lanes on 64 different rows (which made the
    // gather 16x more L1 transactions than bytes warranted).  Row ids travel through v_readlane.
    const int e = t & (BATCH - 1);
    const bool even_rows = (K & 1) == 0;  // rows of an even K are 8-byte aligned
    int id_next = 0;    // id of entry e of the NEXT batch to stage (threads < 128)
    {
        if (t < BATCH && t < n) id_next = (int)a.point_list[range.x + t];
    }
    // zero the rows once: columns >= K + 4 are never written again and must read as 0 in the B operand
    for (int i = t; i < BATCH * FS; i += 256) s_feat[i] = 0.f;

    const int dbg = a.debug_flags;
    auto mfma_pair = [&](float w0, int j0, float w1, int j1) {
        if (dbg & 1) return;
        const uint2w_ sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w0), __float_as_uint(w1), false, false);
        const float* brow = s_feat + (lane < 32 ? j0 : j1) * FS + (lane & 31);
        float bv[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) bv[b] = brow[32 * b];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            D[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[0]), bv[b], D[0][b], 0, 0, 0);
            D[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(sw[1]), bv[b], D[1][b], 0, 0, 0);
        }
    };

    for (int start = 0; start < n; start += BATCH) {
        const bool wave_done = __ballot(!done) == 0ull;
        if (lane == 0) s_wdone[wv] = wave_done;
        __syncthreads();  // also: everyone has finished reading the previous batch
        if (s_wdone[0] & s_wdone[1] & s_wdone[2] & s_wdone[3]) break;
        const int cnt = min(BATCH, n - start);
        uint32_t qmask = 0u;
        {
            // per-entry part (threads 0..127); tail entries re-read entry 0 of the list and are never listed
            const bool live = t < BATCH && e < cnt;
            const int my_id = live ? id_next : (int)a.point_list[range.x];
            const size_t id = (size_t)my_id;
            const float2 xy = a.means2D[id];
            const float4 co = a.conic_opacity[id];
            const float dep = a.depths[id];
            const float cr = a.colors[3 * id], cg = a.colors[3 * id + 1], cb = a.colors[3 * id + 2];
            // row ids of this wave's 32 rows sit in lanes 0..31 of waves 0/1 only: pass them through LDS-free shuffles
            // (wave w needs entries 32w..32w+31 = lanes (32w & 63).. of wave w >> 1) -> publish via s_ids
            if (t < BATCH) s_ids[t] = my_id;
            {
                const int i = start + BATCH + t;
                if (t < BATCH && i < n) id_next = (int)a.point_list[range.x + i];
            }
            __syncthreads();
#pragma unroll 1
            for (int r0 = 0; r0 < ((dbg & 2) ? 0 : 32); r0 += ROWS_PER_ROUND) {
                float2 v[ROWS_PER_ROUND];
#pragma unroll
                for (int r = 0; r < ROWS_PER_ROUND; r++) {
                    const int row_id = __builtin_amdgcn_readfirstlane(s_ids[wv * 32 + r0 + r]);
                    const float* row = a.semantics + (size_t)row_id * (size_t)K;
                    const int c = min(2 * lane, K - 2 + (K & 1));  // clamp so that the address stays inside the row
                    if (even_rows) {
                        v[r] = *reinterpret_cast<const float2*>(row + c);
                    } else {
                        v[r].x = row[min(2 * lane, K - 1)];
                        v[r].y = row[min(2 * lane + 1, K - 1)];
                    }
                }
#pragma unroll
                for (int r = 0; r < ROWS_PER_ROUND; r++) {
                    float* dst = &s_feat[(wv * 32 + r0 + r) * FS + 2 * lane];
                    if (even_rows) {
                        if (2 * lane < K) *reinterpret_cast<float2*>(dst) = v[r];
                    } else {
                        if (2 * lane < K) dst[0] = v[r].x;
                        if (2 * lane + 1 < K) dst[1] = v[r].y;
                    }
                }
            }
            if (t < BATCH) {
                // r, g, b, depth directly behind the semantics (columns the row copies never touch)
                float* dst = &s_feat[e * FS + K];
                dst[0] = cr; dst[1] = cg; dst[2] = cb; dst[3] = dep;
                if (live) {
                    const uint32_t mask16 = subblock_mask(xy.x, xy.y, co.x, co.y, co.z, co.w, tile_x0, tile_y0);
                    qmask = quadrant_bits(mask16);
                    a.masks[range.x + start + e] = mask16;   // for the backward's staging (RenderFwdArgs::masks)
                }
                s_geo[e] = make_float4(xy.x, xy.y, (-0.5f * HSR_LOG2E) * co.x, -HSR_LOG2E * co.y);
                s_co[e] = make_float2((-0.5f * HSR_LOG2E) * co.z, co.w);
            }
        }
        // slots are 0..127: staged by waves 0 and 1; waves 2 and 3 publish empty segments
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        __syncthreads();
        if (wave_done || (dbg & 4)) continue;

        for (int seg = 0; seg < 2; seg++) {
            const int m = s_lcnt[wv][seg];
            int j_next = s_list[wv][seg * 64];
            for (int k = 0; k < m; k++) {
                const int j = j_next;   // slot fetched one iteration ahead (slot -> record are dependent LDS round trips)
                j_next = s_list[wv][seg * 64 + min(k + 1, 63)];
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
                    const float dep = s_feat[j * FS + K + 3];
                    if (cross) {
                        median_D = dep;
                        median_at = (uint32_t)(start + j + 1);
                    }
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
    // transpose one 32-channel block at a time through LDS (row stride 33) back to lane = pixel
    __syncthreads();  // all waves are past their last read of s_feat
    float* tp = s_feat + wv * (64 * 33);
    if (inside) {
        a.final_T[pix_id] = T;
        a.n_contrib[pix_id] = last_contributor;
        a.median_pos[pix_id] = median_at;
        a.out_median_depth[pix_id] = median_D;
        a.out_opacity[pix_id] = 1.0f - T;
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            tp[i * 33 + (lane & 31)] = D[0][b][r];
            tp[(32 + i) * 33 + (lane & 31)] = D[1][b][r];
        }
        __builtin_amdgcn_wave_barrier();
        if (inside) {
            const float* mine = tp + lane * 33;
#pragma unroll 4
            for (int c = 0; c < 32; c++) {
                const int ch = 32 * b + c;
                float* dst = ch < K ? a.out_semantic + (size_t)ch * N : (ch < K + 3 ? a.out_color + (size_t)(ch - K) * N : a.out_depth);
                if (ch <= K + 3) dst[pix_id] = mine[c];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// semantic variant, 29 <= K <= 124.  Returns false when K is outside that range (caller falls back).
bool hsr_launch_render_forward_wide(const RenderFwdArgs& a_, hipStream_t stream)
{
    RenderFwdArgs a = a_;
    static const int dbg = hsr_ablate_env("HSR_FWD_DEBUG") ? atoi(hsr_ablate_env("HSR_FWD_DEBUG")) : 0;   // 0 in the product build
    a.debug_flags = dbg;
    if (!a.semantic || a.K < 29 || a.K > 124) return false;  // K + 4 channels must need 2..4 blocks of 32
    const int tiles = ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    const dim3 grid(hsr_tile_grid(tiles)), block(256);
    const int nb = (a.K + 4 + 31) / 32;  // K semantic channels + r, g, b, depth
    if (nb == 2) render_fwd_wide_kernel<2><<<grid, block, 0, stream>>>(a);
    else if (nb == 3) render_fwd_wide_kernel<3><<<grid, block, 0, stream>>>(a);
    else render_fwd_wide_kernel<4><<<grid, block, 0, stream>>>(a);
    return true;
}
