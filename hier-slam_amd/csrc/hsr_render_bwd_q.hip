// hsr_render_bwd_q.hip — backward tile kernel, packed rows, K <= 27 and the geometry-only (tracking) variant: round 4.
//
// Same per-pixel semantics as hsr_render_bwd.hip (reference backward.cu:472-899) and the same decomposition as round 2/3's
// hsr_render_bwd_sub.hip: one workgroup per 16x16 tile, a wave per 8x8 quadrant, a 16-lane group per 4x4 sub-block, the wave
// walks its quadrant list 16 entries (a "chunk") at a time and every group visits only the entries that reach its sub-block.
//
// What changed: WHERE the ten per-(splat, quadrant) sums that are not channel sums are formed.  A visit used to end with
// twelve per-pixel products and a 20-instruction transposing DPP butterfly over the group's 16 lanes (7 values), i.e. 32 of the
// ~70 vector instructions of a visit — the part of the kernel that does not depend on K.  Six of the
// seven are polynomial moments of ONE per-pixel number, gda = G * dL/dalpha:
//     sum gda,  sum gda*dx,  sum gda*dy,  sum gda*dx^2,  sum gda*dx*dy,  sum gda*dy^2          (dx = splat centre - pixel centre)
// A visit now just stores its two per-pixel factors — w (the blend weight, as before) and gda — into LDS and moves on: ~28 vector
// instructions.  At the end of the chunk
//   * W . G on the matrix cores gives the K + 5 channel sums (as before);
//   * lane (row, group) reads the group's 16 gda values of that row and forms their six moments ABOUT THE SUB-BLOCK'S CENTRE
//     (|u|, |v| <= 1.5 px, the basis values are small exact constants, the sums are separable: 57 instructions per lane and
//     chunk instead of 32 per lane and visit), then shifts them to the splat's centre with the splat's own coefficients:
//         S_x  = (A' ex + B'/2 ey) M0 - A' Mu - B'/2 Mv        (ex, ey = splat centre - sub-block centre)
//         S_xx = ex^2 M0 - 2 ex Mu + Muu     ...
//     Round 2 summed RAW moments (sum q dx, |dx| up to the splat's extent) and cancelled afterwards: 15x the error of the
//     reference's per-pixel formulation on elongated splats.  Here the cancelling products (A' ex against
//     B'/2 ey) are combined ONCE per (row, group) in a single FMA — the same rounding the reference pays per pixel — and what
//     is summed in fp32 before a cancellation is bounded by 1.5 px, not by the splat's size; tests/test_gpu_truth.py is the gate
//     (all of its cases, both 3 000-case fuzz seeds: DESIGN.md §2).
//   * the seventh value (the median-depth gradient: one pixel -> one splat, ever) never enters the loop: a per-batch table (see
//     "median" in the kernel).
// Idle groups (a group whose list is shorter than the longest of the four) visit a DUMMY entry of opacity 0 and write to a
// dummy segment: no validity predicate, no exec juggling in the loop.  Inactive pixels are handled by masking alpha and G
// to zero (T * rcp(1 - 0) = T, fma(0, x, R) = R): two selects per visit instead of four.  The loop is software-pipelined by
// hand (record of visit i + 2 in flight, alpha stage of visit i + 1, blend stage of visit i).
//
// What the measurements of the round say (EXPERIMENTS.md §10; profiles/r04_*):
//   * the kernels are bound by instruction ISSUE — a wave issues one instruction every 7-10 cycles whatever its kind, a SIMD ~2.4
//     cycles per instruction with four waves — so a visit costs its instruction count, and the chip delivers that count times the
//     waves in flight: the first version of this file (two [16][64] row panels, 53 KB of LDS, three workgroups per CU) had 25 % fewer
//     vector instructions than round 3's kernel and exactly its run time.  Hence the panels are stored as SEGMENTS, one per
//     (row, group) pair a chunk actually visits (29.5 of 64 on average): 40 KB, four workgroups per CU, nothing to clear;
//   * with K > 0 the kernel is bound by its gradient atomics, not by any of this: three 64-byte lines per (splat, quadrant) row,
//     4.57 M requests at the headline = 0.225 ms at the memory side's rate, 0.28 measured with or without the instruction savings.
//     What helps there is fewer LINES: compact rows (hsr_tile_common.h) make K = 0 one line instead of two and 12 <= K <= 20 two
//     instead of three; K = 26 needs 36 floats and stays at three.
#include "hsr_tile_common.h"
#include <stdlib.h>
#include <string.h>

#ifdef HSR_TRACE
// Diagnostic build only (make -C hier-slam_amd/csrc trace -> libhsr_rast_trace.so, tools/trace_bwd.py): per-wave cycle counts of
// the phases of render_bwd_q_kernel, clock64 deltas accumulated in registers and dumped at the end.  Never in the product.
#define HSR_TRACE_SLOTS 8
__device__ unsigned long long g_hsr_trace_q[16384 * HSR_TRACE_SLOTS];
extern "C" int hsr_debug_read_trace_q(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hsr_trace_q), sizeof(unsigned long long) * (size_t)n);
}
#define TR_NOW() clock64()
#define TR_ADD(acc, t0) (acc) += (unsigned long long)(clock64() - (t0))
#else
#define TR_NOW() 0ll
#define TR_ADD(acc, t0) ((void)0)
#endif

namespace {

// see hsr_render_bwd_sub.hip: makes the staging registers of the next batch "used" before the first atomics of this batch are issued
#define HSR_SETTLE_STAGING()                                                                                                   \
    asm volatile("" ::"v"(id_next), "v"(p_xy.x), "v"(p_xy.y), "v"(p_co.x), "v"(p_co.y), "v"(p_co.z), "v"(p_co.w), "v"(p_r), "v"(p_g), \
                 "v"(p_b), "v"(p_d), "v"(p_mask))

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Timing ablations (diagnostic build only; results are WRONG with any of them set): HSR_DEBUG_FLAGS bit 1 (2) no panel stores, bit 2 (4) no
// median add, bit 3 (8) no flush at all, bit 5 (32) no moments / emission table in the flush, bit 6 (64) no matrix instructions, bit 7 (128) no
// panel clear, bit 8 (256) no visit loop, bit 9 (512) the visit's record is not read (the chunk's first one is reused), bit 10 (1024) no
// v_exp_f32 / v_rcp_f32, bit 11 (2048) no list-element reads.  tools/r04_ablate_q.sh
#ifdef HSR_ABLATE
#define QAB(a, bit) ((a).debug_flags & (bit))
#else
#define QAB(a, bit) 0
#endif

// orders the LDS accesses of ONE wave (stores before it are visible to the wave's loads after it); no workgroup barrier
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int Q_ROWS = 16;                        // quadrant-list entries per chunk (M dimension of the matrix-core flush)
constexpr int Q_CAP = 40;                         // (row, group) pairs a chunk may visit: one 16-pixel SEGMENT of each panel per pair.  A chunk of 16 rows
                                                  // visits 29.5 pairs on average at the headline; a chunk that would visit more than Q_CAP is cut to Q_CAP / 4 rows
constexpr int Q_SEG_DUMMY = Q_CAP;                // segment idle groups write to
constexpr int Q_SEG_ZERO = Q_CAP + 1;             // segment nobody writes: what an unvisited (row, group) pair reads as
constexpr int Q_NSEG = Q_CAP + 2;
constexpr int Q_SEGW = 17;                        // words per segment: 16 pixels + 1, so that the A-operand reads of 16 rows (one word of 16 different
                                                  // segments) and the 16 lanes' stores of a visit spread over the banks
constexpr int Q_SEGB = Q_SEGW * 4;
constexpr int Q_PANEL = Q_NSEG * Q_SEGW;          // words per panel and wave
constexpr int Q_ORD = 20;                         // elements per group visit list: 16 + look-ahead padding (the pipelined loop reads up to element 18)
constexpr int Q_TB = 36;                          // floats per row of the emission table [row][value 0..7][group]
constexpr int Q_ENTB = 48;                        // bytes per staged record
static_assert(2 * Q_PANEL >= 64 * 17, "the two panels double as the prologue's transposition scratch");
static_assert(Q_ROWS * Q_TB <= Q_SEG_ZERO * Q_SEGW, "the emission table overlays the W panel below its zero segment");

// KC semantic channels [0, KC) + r, g, b, depth, opacity(direct) on the matrix cores (KC + 5 <= 32); GEO: none of them (a TRACKING
// iteration of Hier-SLAM: only the camera pose is optimised, scripts/hierslam.py:1683-1860, so autograd asks for dL_dmeans3D /
// dL_dmeans2D alone): the depth sum joins column 6 and a row is ONE 64-byte line.
// Four workgroups per CU (128 registers) where the B operand is one column group or none; with two column groups (32 registers of B
// operand) the kernel would spill at 128 registers, and its time is set by the gradient atomics anyway (three lines per row): three.
// CL: compact rows (hsr_tile_common.h, hsr_grow_col): the last min(9, K + 5) channel columns leave through line 0 together with columns 0..6.
// SEMA (with GEO): the opt-in "exact" semantic -> alpha pass (hsr_set_semantic_alpha_mode; the reference's backward.cu:834-845 reads an
// unwritten scratch there, so its semantic loss never reaches alpha — DESIGN.md §9).  The term is linear in dL/dalpha, so it is its own
// pass over the same lists: h := sum over channels [a.sem_c0, a.sem_c0 + KC) of feature(splat) * dL_dpixel_semantic(pixel) in place of
// the colour / depth / opacity dot product, no background term, no median, no direct sums; its moments land in columns 0..5 of the rows
// the main pass fills (a.grow_stride is the main pass's).  The splat's KC features ride behind the 48-byte record.
template <int KC, int BATCH, bool GEO, bool CL = false, bool SEMA = false>
__global__ void __launch_bounds__(256, (SEMA || (!GEO && KC + 5 > 16)) ? 3 : 4) render_bwd_q_kernel(RenderBwdArgs a)
{
    static_assert(!SEMA || (GEO && KC % 4 == 0 && KC > 0), "the semantic -> alpha pass is a geometry-only pass over whole float4s of features");
    constexpr int ENTB = SEMA ? Q_ENTB + 4 * KC : Q_ENTB;   // bytes per staged record
    constexpr int REC4 = ENTB / 16;
    constexpr int NCH = GEO ? 0 : KC + 5;          // sem[KC], r, g, b, depth, opacity(direct)
    constexpr int NG = GEO ? 0 : (NCH + 15) / 16;  // 16-channel groups
    constexpr int NGA = NG > 0 ? NG : 1;
    static_assert(NG <= 2, "at most 32 direct channels per launch");
    static_assert(BATCH <= 256 && (BATCH + 1) * ENTB < 65536, "batch slots are bytes, record offsets 16 bits");
    // one 48-byte record per staged splat { x, y, A', B' | r, g, b, depth | C', opacity, B'/2, - }; record BATCH is the dummy
    __shared__ float4 s_ent[REC4 * (BATCH + 1)];
    __shared__ int s_id[BATCH];
    __shared__ uint16_t s_mask[BATCH + 2];             // sub-block mask of each staged splat
    __shared__ uint8_t s_list[4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ uint8_t s_flat[4][256];
    __shared__ int s_wmax[4];
    __shared__ __attribute__((aligned(16))) float s_pan[4][2 * Q_PANEL];   // per wave: the W panel, then the Q panel, in segments
    __shared__ uint2 s_ord[4][4][Q_ORD];               // per (wave, group): { record offset, segment byte offset } of the entries the group visits: no decoding in the loop
    __shared__ uint32_t s_rowent[4][Q_ROWS];           // record offset of each chunk row
    __shared__ __attribute__((aligned(16))) uint32_t s_cid[4][Q_ROWS];   // packed-row offset (Gaussian id x row stride) of each chunk row
    __shared__ float s_medj[BATCH];                    // median-depth gradient of each staged splat: see "median" below
    __shared__ uint8_t s_segtab[4][Q_ROWS][4];         // segment of each (row, group) pair of the chunk, Q_SEG_ZERO where the group does not visit the row

    const int tile = HSR_TILE_OF_BLOCK(blockIdx.x, (a.W + HSR_TILE_X - 1) / HSR_TILE_X, (a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    if (tile >= ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, gq = lane >> 4, l16 = lane & 15;
    const TileGeom tg = tile_geom_sub(tile, a.W, a.H, t);
    const bool inside = tg.inside;
    const size_t N = (size_t)a.W * a.H;
    const size_t pix_id = (size_t)a.W * tg.py + tg.px;
    float pfx = tg.pfx, pfy = tg.pfy;
    asm volatile("" : "+v"(pfx), "+v"(pfy));   // opaque floats: see render_fwd_kernel
    const uint2 range = a.ranges[tile];
    float* pw = s_pan[wv];
    float* pq = s_pan[wv] + Q_PANEL;
    unsigned long long tr_stage = 0, tr_loop = 0, tr_flush = 0, tr_iters = 0, tr_chunks = 0, tr_accepted = 0, tr_setup = 0;
    (void)tr_setup;
    const long long tr_t0 = TR_NOW();
    (void)tr_stage; (void)tr_loop; (void)tr_flush; (void)tr_iters; (void)tr_chunks; (void)tr_accepted; (void)tr_t0;

    if (t < REC4) s_ent[REC4 * BATCH + t] = make_float4(0.f, 0.f, 0.f, 0.f);   // the dummy record: opacity 0 -> alpha 0 -> never active

    // every prologue load unconditional and issued before anything consumes one
    const size_t pix_ld = inside ? pix_id : 0;
    const float inm = inside ? 1.f : 0.f;
    const float T_final_ld = a.final_T[pix_ld];
    const int last_contributor_ld = (int)a.n_contrib[pix_ld];
    const int median_at_ld = (int)a.median_pos[pix_ld];
    float dpx0 = a.dL_dpix[pix_ld], dpx1 = a.dL_dpix[N + pix_ld], dpx2 = a.dL_dpix[2 * N + pix_ld];
    float dpd = a.dL_dpix_depth[pix_ld], dpm = a.dL_dpix_median[pix_ld], dpo = a.dL_dpix_opacity[pix_ld];
    float semv[KC > 0 ? KC : 1];
#pragma unroll
    for (int c = 0; c < KC; c++) semv[c] = a.dL_dpix_sem[(size_t)min((SEMA ? a.sem_c0 : 0) + c, a.K - 1) * N + pix_ld];
    if (SEMA) {
#pragma unroll
        for (int c = 0; c < KC; c++) semv[c] = (a.sem_c0 + c < a.K) ? semv[c] * inm : 0.f;
        dpx0 = dpx1 = dpx2 = dpd = dpm = dpo = 0.f;   // nothing but the semantic term in this pass
    }
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
    float Breg[NGA][16];
    if (!GEO) {
        float gv[NGA][16];
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
        // the panels are private to the wave: a wave-level fence orders its LDS stores and loads, the four waves do not have to meet
#pragma unroll
        for (int g = 0; g < NG; g++) {
#pragma unroll
            for (int c = 0; c < 16; c++) pw[lane * 17 + c] = gv[g][c];
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < 16; m++) Breg[g][m] = pw[(4 * m + (lane >> 4)) * 17 + (lane & 15)];
            wave_lds_fence();
        }
    }
    if (lane < 2 * Q_SEGW) {   // the zero segments (GEO: one, of float2)
        if (GEO) pw[Q_SEG_ZERO * 2 * Q_SEGW + lane] = 0.f;
        else (lane < Q_SEGW ? pw : pq - Q_SEGW)[Q_SEG_ZERO * Q_SEGW + lane] = 0.f;
    }
    __syncthreads();   // s_wmax, the dummy record
    const int hi_all = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    const long long tr_t1 = TR_NOW();   // end of the prologue
    (void)tr_t1;

    const float ntfbg = -T_final * (a.bg[0] * dpx0 + a.bg[1] * dpx1 + a.bg[2] * dpx2);   // the background term of dL/dalpha, x 1 / (1 - alpha)
    const float kx2 = (a.W) / HSR_LOG2E, ky2 = (a.H) / HSR_LOG2E;                        // 2 kx, 2 ky of hsr_render_bwd_sub.hip
    float Racc = 0.f;   // the reference's accum_rec AFTER the last accepted splat (backward.cu:630-640, h = colour . dL_dpixel)
    char* const panb = reinterpret_cast<char*>(&s_pan[0][0]);
    const char* const entb = reinterpret_cast<const char*>(&s_ent[0]);
    // Panel layout.  K > 0 / K = 0: a W panel and a Q panel of Q_NSEG segments of 17 words (16 pixels + 1).  GEO: ONE panel of segments of
    // 17 float2 { gda, w * dL_ddepth }: one 8-byte store per visit, and the flush reads both factors of a pixel with one 8-byte load.
    constexpr uint32_t SEGB = GEO ? 2 * Q_SEGB : Q_SEGB;                     // bytes per segment
    const uint32_t wave_off = (uint32_t)(wv * 2 * Q_PANEL * 4);              // byte offset of the wave's panels in s_pan
    const uint32_t lane16_off = wave_off + (uint32_t)(l16 * (GEO ? 8 : 4));  // ... of this lane's pixel in segment 0
    const uint2 DUMMY_ELEM = make_uint2((uint32_t)(BATCH * ENTB), (uint32_t)Q_SEG_DUMMY * SEGB);
    uint2* const ordp = &s_ord[wv][gq][0];

    // ---- software-pipelined staging ----
    int id_next = 0, id_cur = 0;
    float2 p_xy = {0, 0};
    float4 p_co = {0, 0, 0, 0};
    float p_r = 0, p_g = 0, p_b = 0, p_d = 0;
    uint32_t p_mask = 0u;
    const int n_list = (int)(range.y - range.x);
    auto fetch_id = [&](int hi) -> int { return (int)a.point_list[range.x + min(max(hi - 1 - t, 0), max(n_list - 1, 0))]; };
    auto fetch_mask = [&](int hi) -> uint32_t { return a.masks[range.x + min(max(hi - 1 - t, 0), max(n_list - 1, 0))]; };
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

    // the chunk's segments -> packed rows
    auto flush = [&](int nrows) {
        wave_lds_fence();
        const uint32_t st = *reinterpret_cast<const uint32_t*>(&s_segtab[wv][l16][0]);   // row l16: the segments of its four groups
        // (per-lane constants of the flush, recomputed here rather than kept live through the visit loop: registers)
        // centre of this lane's sub-block
        const float cxg = pfx - (float)(l16 & 3) + 1.5f, cyg = pfy - (float)(l16 >> 2) + 1.5f;
        // packed-row columns of the accumulator columns this lane holds (col = lane & 15 of channel group g), or -1; compact rows: columns
        // below 16 leave through the emission table (slot = column - 7)
        constexpr bool ALL_IN_LINE0 = CL && NCH <= 9;   // K <= 4 with compact rows: every channel sum rides in line 0, no atomics of their own
        int colg[NGA], tslot[NGA];
#pragma unroll
        for (int g = 0; g < NGA; g++) {
            const int ch = 16 * g + l16;                                                        // MFMA column: sem 0..KC-1, r, g, b, depth, opacity
            const int chl = ch < KC ? (ch < a.K ? ch : -1) : (ch < KC + 5 ? a.K + (ch - KC) : -1);   // the row's channel column, or none
            const int col = (GEO || chl < 0) ? -1 : hsr_grow_col(CL ? 1 : 0, a.K, chl);
            colg[g] = (col >= 16 && !ALL_IN_LINE0) ? col : -1;
            tslot[g] = (CL && col >= 7 && col < 16) ? col - 7 : -1;
        }
        // (1) D[16 entries][16 NG channels] = W . G: lane (row l16, k-slice gq) reads pixel 4 m + gq, i.e. word 4 (m & 3) + gq of the
        //     row's segment of group m >> 2 (the zero segment where that group does not visit the row)
        f32x4 acc[NGA];
#pragma unroll
        for (int g = 0; g < NGA; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!GEO) {
            // all sixteen A-operand words requested before the first matrix instruction waits for one
            float av[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const float* sp = pw + ((st >> (8 * g4)) & 0xFFu) * Q_SEGW + gq;
#pragma unroll
                for (int i = 0; i < 4; i++) av[4 * g4 + i] = sp[4 * i];
            }
            asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]));
            asm volatile("" : "+v"(av[8]), "+v"(av[9]), "+v"(av[10]), "+v"(av[11]), "+v"(av[12]), "+v"(av[13]), "+v"(av[14]), "+v"(av[15]));
            if (NG == 1) {
                // one column group: two accumulation chains (even / odd k-steps) so that a matrix instruction does not wait for the one before
                f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 16; m += 2) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], Breg[0][m], acc[0], 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m + 1], Breg[0][m + 1], acc2, 0, 0, 0);
                }
                acc[0] += acc2;
            } else {
#pragma unroll
                for (int m = 0; m < 16; m++)
#pragma unroll
                    for (int g = 0; g < NG; g++)
                        if (!QAB(a, 64)) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], Breg[g][m], acc[g], 0, 0, 0);
                        else acc[g][0] += av[m] * Breg[g][m];
            }
        }
        // (2) lane (row l16, group gq): the six moments of the group's 16 gda values of that row about the sub-block centre.
        //     pixel 4 yy + xx of the group sits at (u, v) = (xx - 1.5, yy - 1.5)
        float o[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!QAB(a, 32)) {
            const uint32_t myseg = (st >> (8 * gq)) & 0xFFu;
            float qv[16], wd[GEO ? 16 : 1], medsum;
            if (GEO) {
                const float2* qp = reinterpret_cast<const float2*>(pw) + myseg * Q_SEGW;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float2 v = qp[i];
                    qv[i] = v.x;
                    wd[GEO ? i : 0] = v.y;
                }
                medsum = 0.f;
            } else {
                const float* qp = pq + myseg * Q_SEGW;
#pragma unroll
                for (int i = 0; i < 16; i++) qv[i] = qp[i];
                medsum = 0.f;
            }
            const uint32_t eo = s_rowent[wv][l16];
            // median: whatever has been added for this row's splat so far, by any wave — taken (exchanged with zero), so that it is emitted once
            if (!SEMA && gq == 0 && eo < (uint32_t)(BATCH * ENTB)) medsum = atomicExch(&s_medj[eo / ENTB], 0.f);
            const float4* ent = reinterpret_cast<const float4*>(entb + eo);
            const float4 e0 = ent[0], e2 = ent[2];
            float s0[4], s1[4], s2[4];
#pragma unroll
            for (int yy = 0; yy < 4; yy++) {
                const float qa = qv[4 * yy], qb = qv[4 * yy + 1], qc = qv[4 * yy + 2], qd = qv[4 * yy + 3];
                const float ad = qa + qd, bc = qb + qc;
                s0[yy] = ad + bc;
                s1[yy] = fmaf(1.5f, qd - qa, 0.5f * (qc - qb));
                s2[yy] = fmaf(2.25f, ad, 0.25f * bc);
            }
            const float M0 = (s0[0] + s0[3]) + (s0[1] + s0[2]);
            const float Mu = (s1[0] + s1[3]) + (s1[1] + s1[2]);
            const float Muu = (s2[0] + s2[3]) + (s2[1] + s2[2]);
            const float Mv = fmaf(1.5f, s0[3] - s0[0], 0.5f * (s0[2] - s0[1]));
            const float Mvv = fmaf(2.25f, s0[0] + s0[3], 0.25f * (s0[1] + s0[2]));
            const float Muv = fmaf(1.5f, s1[3] - s1[0], 0.5f * (s1[2] - s1[1]));
            // shift to the splat's centre: dx = ex - u, dy = ey - v (reference backward.cu:881-896 sums these per pixel)
            const float ex = e0.x - cxg, ey = e0.y - cyg;
            const float Ap = e0.z, Cp = e2.x, op = e2.y, hB = e2.z;
            const float Sx = fmaf(fmaf(Ap, ex, hB * ey), M0, -fmaf(Ap, Mu, hB * Mv));     // sum gda (A' dx + B'/2 dy)
            const float Sy = fmaf(fmaf(Cp, ey, hB * ex), M0, -fmaf(Cp, Mv, hB * Mu));     // sum gda (C' dy + B'/2 dx)
            const float Sxx = fmaf(ex, fmaf(ex, M0, -2.0f * Mu), Muu);                     // sum gda dx^2
            const float Sxy = fmaf(ex, fmaf(ey, M0, -Mv), fmaf(-ey, Mu, Muv));             // sum gda dx dy
            const float Syy = fmaf(ey, fmaf(ey, M0, -2.0f * Mv), Mvv);                     // sum gda dy^2
            const float mh = -0.5f * op;
            o[0] = (kx2 * op) * Sx;   // dL_dmean2D.x
            o[1] = (ky2 * op) * Sy;   // dL_dmean2D.y
            o[2] = mh * Sxx;          // dL_dconic.x
            o[3] = mh * Sxy;          // dL_dconic.y
            o[4] = mh * Syy;          // dL_dconic.w
            o[5] = M0;                // dL_dopacity, alpha path
            // column 6: the median-depth sum taken above (group 0); GEO: + the segment's depth sum (the panel's .y holds w * dL_ddepth)
            float s6 = medsum;
            if (GEO) {
                float wsum[4];
#pragma unroll
                for (int yy = 0; yy < 4; yy++) wsum[yy] = (wd[GEO ? 4 * yy : 0] + wd[GEO ? 4 * yy + 1 : 0]) + (wd[GEO ? 4 * yy + 2 : 0] + wd[GEO ? 4 * yy + 3 : 0]);
                s6 += (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
            }
            o[6] = s6;
        }
        // D[row = 4*(lane>>4) + r][col = lane&15]: one atomic wave-instruction per register = 4 rows x 64 bytes
        if (!GEO && !ALL_IN_LINE0) {
            const uint4 b4 = *reinterpret_cast<const uint4*>(&s_cid[wv][4 * (lane >> 4)]);   // the four row offsets in one LDS read
            const uint32_t bb[4] = {b4.x, b4.y, b4.z, b4.w};
            const int nr = nrows - 4 * (lane >> 4);   // how many of this lane's four rows exist
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int g = 0; g < NG; g++)
                    if (r < nr && colg[g] >= 0 && !(a.debug_flags & 1))
                        atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(a.grow) + 4u * (bb[r] + (uint32_t)colg[g])), acc[g][r]);
        }
        if (CL && !GEO) {
            // compact rows: the channel sums that ride in line 0 -> tba[row][slot] (it overlays the Q panel, whose segments have been read)
            wave_lds_fence();
#pragma unroll
            for (int g = 0; g < NG; g++)
                if (tslot[g] >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++) pq[(4 * (lane >> 4) + r) * 12 + tslot[g]] = acc[g][r];
                }
        }
        // (3) the four groups' contributions of a row meet in the emission table (it overlays the W panel, which is dead now)
        if (QAB(a, 32)) return;
        wave_lds_fence();
        float* tb = pw;
#pragma unroll
        for (int vi = 0; vi < 7; vi++) tb[l16 * Q_TB + 4 * vi + gq] = o[vi];
        wave_lds_fence();
        // Emission of line 0.  Every LDS read of every pass is issued first and unconditionally (clamped indices), the atomics follow: left to
        // itself hipcc sinks each read into the branch that uses it and the passes become a chain of a dozen exposed LDS round trips
        // (the K = 0 flush took 3 800 cycles per chunk against 1 650 for the geometry-only one: profiles/r04_k_trace_bwd.txt and the A/B of EXPERIMENTS.md §10a).
        if (CL && !GEO) {
            // columns 0..15: four wave-instructions of 4 rows x 16 values, each row's line ONE request
            const int nl0 = hsr_grow_nl0(a.K);
            const int vi = lane & 15;
            float4 sa[4];
            float ta[4];
            uint32_t cid[4];
#pragma unroll
            for (int pass = 0; pass < 4; pass++) {
                const int row = (lane >> 4) + 4 * pass;
                sa[pass] = *reinterpret_cast<const float4*>(tb + row * Q_TB + (vi & 7) * 4);
                ta[pass] = pq[row * 12 + min(max(vi - 7, 0), 11)];
                cid[pass] = s_cid[wv][row];
            }
            asm volatile("" : "+v"(sa[0].x), "+v"(sa[1].x), "+v"(sa[2].x), "+v"(sa[3].x), "+v"(ta[0]), "+v"(ta[1]), "+v"(ta[2]), "+v"(ta[3]),
                         "+v"(cid[0]), "+v"(cid[1]), "+v"(cid[2]), "+v"(cid[3]));
#pragma unroll
            for (int pass = 0; pass < 4; pass++) {
                const int row = (lane >> 4) + 4 * pass;
                const float val = vi < 7 ? (sa[pass].x + sa[pass].y) + (sa[pass].z + sa[pass].w) : ta[pass];
                if (vi < 7 + nl0 && row < nrows && val != 0.f && !(a.debug_flags & 1))
                    atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(a.grow) + 4u * (cid[pass] + (uint32_t)vi)), val);
            }
        } else {
            // columns 0..6: two wave-instructions of 8 rows x 7 values, so that each row's line is ONE request
            const int vi = lane & 7;
            float4 sa[2];
            uint32_t cid[2];
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                const int row = (lane >> 3) + 8 * pass;
                sa[pass] = *reinterpret_cast<const float4*>(tb + row * Q_TB + vi * 4);
                cid[pass] = s_cid[wv][row];
            }
            asm volatile("" : "+v"(sa[0].x), "+v"(sa[1].x), "+v"(cid[0]), "+v"(cid[1]));
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                const int row = (lane >> 3) + 8 * pass;
                const float val = (sa[pass].x + sa[pass].y) + (sa[pass].z + sa[pass].w);
                if (vi < 7 && row < nrows && val != 0.f && !(a.debug_flags & 1))
                    atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(a.grow) + 4u * (cid[pass] + (uint32_t)vi)), val);
            }
        }
        wave_lds_fence();
    };

    for (int hi = hi_all; hi > 0; hi -= BATCH) {
        const int cnt = min(BATCH, hi);
        // list position of batch slot j is hi - 1 - j: "behind the last contributor" and "the median splat" as record-offset tests, per batch
        const int jf_off = (hi - last_contributor) * ENTB;
        const long long ts = TR_NOW();
        (void)ts;
        __syncthreads();
        uint32_t qmask = 0u;
        if (t < cnt) {
            const uint32_t mask = p_mask;
            qmask = quadrant_bits(mask);
            s_mask[t] = (uint16_t)mask;
            s_id[t] = id_cur;
            s_medj[t] = 0.f;
            s_ent[REC4 * t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_ent[REC4 * t + 1] = make_float4(p_r, p_g, p_b, p_d);
            s_ent[REC4 * t + 2] = make_float4((-0.5f * HSR_LOG2E) * p_co.z, p_co.w, (-0.5f * HSR_LOG2E) * p_co.y, 0.f);   // C', opacity, B' / 2
            if (SEMA) {   // the splat's features of this pass's channels (zero past K); read here, once per staged splat
                const float* f = a.semantics + (size_t)id_cur * (size_t)a.K;
#pragma unroll
                for (int q = 0; q < KC / 4; q++) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int ch = a.sem_c0 + 4 * q + i;
                        v[i] = ch < a.K ? f[ch] : 0.f;
                    }
                    s_ent[REC4 * t + 3 + q] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        __syncthreads();
        // median: the median-depth gradient of a pixel goes to ONE splat, the one at which the forward saw T cross 0.5 (ImgState::median_pos;
        // the forward records it only on a splat the pixel accepts, and this kernel evaluates alpha with the same instructions, so the
        // pixel accepts it here too).  The pixel adds it into a per-batch table indexed by batch slot; the flush of a row takes (exchanges
        // with zero) what its splat's slot holds.  The splat reaches the pixel's quadrant, so the pixel's own wave has it as a row of a
        // chunk AFTER this add (LDS operations of a wave execute in order): every contribution is emitted exactly once, whichever wave's
        // flush picks it up — the destination (column 6 of the splat's row) is the same.  Nothing of this is left in the visit loop.
        {
            const int jm = hi - 1 - median_at;
            if (jm >= 0 && jm < cnt && dpm != 0.f) atomicAdd(&s_medj[jm], dpm);
        }
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
        int nrows = 0;
        for (int c0 = 0; c0 < total; c0 += nrows) {
            nrows = min(Q_ROWS, total - c0);
            const long long tsu = TR_NOW();
            (void)tsu;
            // lane (group gq, row l16): does chunk entry l16 touch sub-block (wv, gq)?
            const int jr = l16 < nrows ? (int)s_flat[wv][c0 + l16] : BATCH;
            bool touch = l16 < nrows && ((s_mask[jr] >> (4 * wv + gq)) & 1u);
            uint64_t ball = __ballot(touch);
            if (__popcll(ball) > Q_CAP) {   // more pairs than segments (big splats): a shorter chunk, whose every pair fits
                nrows = Q_CAP / 4;
                touch = touch && l16 < nrows;
                ball = __ballot(touch);
            }
            const uint32_t eoff = (uint32_t)(l16 < nrows ? jr : BATCH) * (uint32_t)ENTB;
            if (gq == 0) {
                s_rowent[wv][l16] = eoff;
                s_cid[wv][l16] = (uint32_t)s_id[min(jr, BATCH - 1)] * (uint32_t)a.grow_stride;   // once per chunk row, not once per emitted register
            }
            // segments in ballot order; the group's visit list: every slot the dummy first, then the touched rows in list order
            const uint32_t seg = touch ? __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u)) : (uint32_t)Q_SEG_ZERO;
            s_segtab[wv][l16][gq] = (uint8_t)seg;
            ordp[l16] = DUMMY_ELEM;
            if (l16 < Q_ORD - 16) ordp[16 + l16] = DUMMY_ELEM;
            const uint32_t gmask = (uint32_t)(ball >> (16 * gq)) & 0xFFFFu;
            if (touch) {
                ordp[__popc(gmask & ((1u << l16) - 1u))] = make_uint2(eoff, seg * SEGB);
            }
            // wave-uniform iteration count: the longest of the four lists
            const int iters = __builtin_amdgcn_readfirstlane(max(max(__popc((uint32_t)ball & 0xFFFFu), __popc((uint32_t)(ball >> 16) & 0xFFFFu)),
                                                                 max(__popc((uint32_t)(ball >> 32) & 0xFFFFu), __popc((uint32_t)(ball >> 48)))));
            wave_lds_fence();
            // A wave issues one instruction every 7-10 cycles whatever the instruction, so what a visit costs is its instruction count and
            // what the chip delivers is that count times the waves in flight (profiles/r04_b_pmc_*.json.txt).  The visit is software-
            // pipelined by hand: half-iteration `it` fetches the record of visit it + 2, evaluates alpha for visit it + 1 (stage A:
            // everything that does not depend on the transmittance) and blends visit it (stage B: the short T / accum_rec chain); two
            // half-iterations per loop trip, so that the stages' registers swap roles without copies.
            struct StageA { float am, Gm, inv, nb, h; uint32_t ro; };
            auto stage_a = [&](const float4& ra, const float4& rb, const float2& rc, uint2 e) -> StageA {
                StageA o;
                const int eo = (int)e.x;
                o.ro = lane16_off + e.y;
                const float dx = ra.x - pfx, dy = ra.y - pfy;
                const float dxx = dx * dx, dxy = dx * dy, dyy = dy * dy;
                const float power2 = fmaf(rc.x, dyy, fmaf(ra.w, dxy, ra.z * dxx));
                const float G = QAB(a, 1024) ? power2 + 1.0f : __builtin_amdgcn_exp2f(power2);
                const float alpha = fminf(0.99f, rc.y * G);
                const bool active = eo >= jf_off && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
#ifdef HSR_TRACE
                {
                    const uint64_t ba = __ballot(active);
#pragma unroll
                    for (int gg = 0; gg < 4; gg++) tr_accepted += ((ba >> (16 * gg)) & 0xFFFFull) != 0ull;
                    tr_iters += 1;   // wave iterations (each is up to four group visits)
                }
#endif
                o.am = active ? alpha : 0.f;
                o.Gm = active ? G : 0.f;
                o.inv = QAB(a, 1024) ? 1.0f + o.am : __builtin_amdgcn_rcpf(1.0f - o.am);   // 1 where the pixel skips the splat: T * 1 = T
                o.nb = ntfbg * o.inv;
                if (SEMA) {
                    const float4* f = reinterpret_cast<const float4*>(entb + e.x + Q_ENTB);
                    float h0 = 0.f, h1 = 0.f;
#pragma unroll
                    for (int q = 0; q < KC / 4; q++) {
                        const float4 v = f[q];
                        h0 = fmaf(v.x, semv[SEMA ? 4 * q : 0], fmaf(v.y, semv[SEMA ? 4 * q + 1 : 0], h0));
                        h1 = fmaf(v.z, semv[SEMA ? 4 * q + 2 : 0], fmaf(v.w, semv[SEMA ? 4 * q + 3 : 0], h1));
                    }
                    o.h = h0 + h1;
                } else {
                    o.h = fmaf(rb.x, dpx0, fmaf(rb.y, dpx1, fmaf(rb.z, dpx2, fmaf(rb.w, dpd, dpo))));
                }
                return o;
            };
            auto stage_b = [&](const StageA& c) {
                const float test_T = T * c.inv;
                const float w = c.am * test_T;
                const float d = c.h - Racc;
                const float dL_dalpha = fmaf(d, test_T, c.nb);
                const float gda = c.Gm * dL_dalpha;
                if (!QAB(a, 2)) {
                    if (GEO) {
                        *reinterpret_cast<float2*>(panb + c.ro) = make_float2(gda, w * dpd);
                    } else {
                        *reinterpret_cast<float*>(panb + c.ro) = w;
                        *reinterpret_cast<float*>(panb + c.ro + Q_PANEL * 4) = gda;
                    }
                } else {
                    asm volatile("" ::"v"(w), "v"(gda));
                }
                Racc = fmaf(c.am, d, Racc);
                T = test_T;
            };
            auto load_rec = [&](uint2 e, float4& ra, float4& rb, float2& rc) {
                const float4* ent = reinterpret_cast<const float4*>(entb + e.x);
                ra = ent[0];
                rb = ent[1];
                rc = *reinterpret_cast<const float2*>(&ent[2]);
            };
            uint2 e1 = ordp[1], e2 = ordp[2];
            float4 ra1, rb1;
            float2 rc1;
            StageA cur;
            {
                const uint2 e0 = ordp[0];
                float4 ra0, rb0;
                float2 rc0;
                load_rec(e0, ra0, rb0, rc0);
                load_rec(e1, ra1, rb1, rc1);
                cur = stage_a(ra0, rb0, rc0, e0);
            }
#ifdef HSR_TRACE
            asm volatile("" ::"v"(cur.am), "v"(cur.h), "v"(ra1.x), "v"(rb1.x));   // the pipeline's first stage has landed
#endif
            TR_ADD(tr_setup, tsu);
            for (int it = 0; it < (QAB(a, 256) ? 0 : iters); it += 2) {
                // first half: visit it + 2's record and the list element after it in flight; A of visit it + 1; B of visit it
                const uint2 e3 = QAB(a, 2048) ? e1 : ordp[it + 3];
                float4 ra2, rb2;
                float2 rc2;
                if (QAB(a, 512)) { ra2 = ra1; rb2 = rb1; rc2 = rc1; }
                else load_rec(e2, ra2, rb2, rc2);
                const StageA nxt = stage_a(ra1, rb1, rc1, e1);
                stage_b(cur);
                // second half (an odd list ends on the dummy entry: weight 0 into the dummy segment)
                const uint2 e4 = QAB(a, 2048) ? e2 : ordp[it + 4];
                if (!QAB(a, 512)) load_rec(e3, ra1, rb1, rc1);
                cur = stage_a(ra2, rb2, rc2, e2);
                stage_b(nxt);
                e1 = e3; e2 = e4;
            }
            {
                const long long tf = TR_NOW();
                (void)tf;
                if (c0 == 0) HSR_SETTLE_STAGING();
                if (!QAB(a, 8)) flush(nrows);
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
            unsigned long long* o = g_hsr_trace_q + (size_t)wid * HSR_TRACE_SLOTS;
            o[0] = (unsigned long long)(clock64() - tr_t0);
            o[1] = (unsigned long long)(tr_t1 - tr_t0);
            o[2] = tr_stage; o[3] = tr_loop; o[4] = tr_flush; o[5] = tr_chunks; o[6] = tr_iters; o[7] = tr_setup;   // slot 7: chunk set-up (round 3's kernel: accepting visits)
        }
    }
#endif
}

}  // namespace

// packed mode, K <= 27, P * grow_stride < 2^30 (32-bit row addressing): the caller checks.  a.grow_layout: hsr_backward_row_layout.
int hsr_launch_render_backward_q(const RenderBwdArgs& a, hipStream_t stream)
{
    const dim3 grid(HSR_GRID_OF_TILES((a.W + HSR_TILE_X - 1) / HSR_TILE_X, (a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)), block(256);
    const int K = a.semantic ? a.K : 0;
    const bool cl = a.grow_layout == 1;
    if (K == 0) {
        if (cl) render_bwd_q_kernel<0, 208, false, true><<<grid, block, 0, stream>>>(a);
        else render_bwd_q_kernel<0, 208, false, false><<<grid, block, 0, stream>>>(a);
    } else if (K <= 11) {
        if (cl) render_bwd_q_kernel<11, 208, false, true><<<grid, block, 0, stream>>>(a);
        else render_bwd_q_kernel<11, 208, false, false><<<grid, block, 0, stream>>>(a);
    } else if (K == 16 && cl) render_bwd_q_kernel<16, 224, false, true><<<grid, block, 0, stream>>>(a);
    else if (K == 26 && !cl) render_bwd_q_kernel<26, 224, false, false><<<grid, block, 0, stream>>>(a);
    else if (cl) render_bwd_q_kernel<27, 224, false, true><<<grid, block, 0, stream>>>(a);
    else render_bwd_q_kernel<27, 224, false, false><<<grid, block, 0, stream>>>(a);
    return HSR_OK;
}

// The exact semantic -> alpha term (opt-in, hsr_set_semantic_alpha_mode(1)): AFTER the main pass has been enqueued on the same stream,
// ceil(K / 16) passes of the SEMA variant add their moments into columns 0..5 of the same rows.  a.grow / a.grow_stride: the main pass's.
int hsr_launch_render_backward_qsema(const RenderBwdArgs& a0, hipStream_t stream)
{
    RenderBwdArgs a = a0;
    const dim3 grid(HSR_GRID_OF_TILES((a.W + HSR_TILE_X - 1) / HSR_TILE_X, (a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)), block(256);
    for (int c0 = 0; c0 < a.K; c0 += 16) {
        a.sem_c0 = c0;
        render_bwd_q_kernel<16, 208, true, false, true><<<grid, block, 0, stream>>>(a);
    }
    return HSR_OK;
}

// geometry-only gradients (a.grow_stride == 16): any K
int hsr_launch_render_backward_qgeo(const RenderBwdArgs& a, hipStream_t stream)
{
    render_bwd_q_kernel<0, 208, true><<<dim3(HSR_GRID_OF_TILES((a.W + HSR_TILE_X - 1) / HSR_TILE_X, (a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)), dim3(256), 0, stream>>>(a);
    return HSR_OK;
}
