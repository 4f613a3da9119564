// hsr_render_fwd.hip — 16x16-tile forward alpha compositing for gfx950 (wave64, 4 waves per tile).
//
// Follows the per-pixel semantics of the reference's renderCUDA / renderCUDA_SEM
// (cuda_rasterizer/forward.cu:261-398, :400-538): front-to-back over the tile's depth-sorted list,
// power>0 and alpha<1/255 skipped, alpha clamped at 0.99, pixel terminated when T(1-alpha) < 1e-4,
// median depth = depth of the splat where T crosses 0.5 (default 15), NO background blend.
//
// What is different from the reference, by design for CDNA4:
//   * every per-splat quantity the blend needs — centre, conic, opacity, colour, depth and the K
//     semantic features — is staged ONCE per tile batch into LDS and then read as wave-uniform
//     broadcasts; the reference gathers colour/depth/semantics from global memory per pixel per splat
//     (forward.cu:503-508).
//   * the staging is software-pipelined: while a batch is blended out of LDS, the gathers of the next
//     batch (one splat per lane: its id two batches ahead, its 40-byte record and its 4K-byte feature
//     row one batch ahead) are already in flight into registers, so the dependent
//     point_list -> record global round trips hide behind compute instead of serialising the tile.
//   * each wave owns an 8x8 quadrant and walks a compacted per-quadrant list built at staging time from
//     the splats' alpha>=1/255 bounding boxes (hsr_tile_common.h), so splats that cannot touch the
//     quadrant cost it nothing; of the survivors, one that no lane accepts costs only the alpha test
//     (wave ballot).  A wave whose 64 pixels are all terminated stops blending but keeps staging.
//   * records are staged pre-scaled (log2(e) and the -0.5 folded into the conic) so alpha costs one
//     v_exp_f32 and a handful of FMAs: the kernel is VALU-issue bound, not HBM bound.
//   * K is a template parameter of the kernel but a run-time argument of the library: known tree
//     sizes get one fused launch, any other K is rendered in 32-channel chunks.
#include <stdlib.h>
#include <string.h>

#include "hsr_tile_common.h"

#ifdef HSR_TRACE
// Diagnostic build only (make -C hier-slam_amd/csrc trace -> libhsr_rast_trace.so, tools/trace_fwd.py): per-wave shader-cycle counts of
// the phases of render_fwd_kernel (sub-block variant).  Never in the product.
__device__ unsigned long long g_hsr_trace_fwd[16384 * 8];
extern "C" int hsr_debug_read_trace_fwd(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hsr_trace_fwd), sizeof(unsigned long long) * (size_t)n);
}
#define TRF_NOW() clock64()
#define TRF_ADD(acc, t0) (acc) += (unsigned long long)(clock64() - (t0))
#else
#define TRF_NOW() 0ll
#define TRF_ADD(acc, t0) ((void)0)
#endif

namespace {

// s[4 r + q] += x[r](quad lane q) * w for r < W, q < 4: the value travels in the FMA's own DPP operand.  Written as ONE asm block per
// group of up to 16 channels because hipcc keeps its packed FMAs and materialises every broadcast with a v_mov_b32_dpp of its own (two
// instructions per channel instead of one); the leading s_nop covers gfx9's "VALU write -> DPP read" wait states whatever the
// compiler put in front of the block (inside it nothing writes an x[]).
template <int W>
__device__ __forceinline__ void quad_fma_words(float (&s)[16], const float (&x)[4], float w)
{
    static_assert(W >= 1 && W <= 4, "1..4 words = 4..16 channels per block");
    if constexpr (W == 4) {
        asm("s_nop 1\n\t"
            "v_fmac_f32_dpp %0, %16, %20 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %16, %20 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %2, %16, %20 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %3, %16, %20 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %4, %17, %20 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %5, %17, %20 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %6, %17, %20 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %7, %17, %20 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %8, %18, %20 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %9, %18, %20 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %10, %18, %20 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %11, %18, %20 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %12, %19, %20 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %13, %19, %20 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %14, %19, %20 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %15, %19, %20 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
            : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15])
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(w));
    } else if constexpr (W == 3) {
        asm("s_nop 1\n\t"
            "v_fmac_f32_dpp %0, %12, %15 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %12, %15 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %2, %12, %15 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %3, %12, %15 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %4, %13, %15 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %5, %13, %15 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %6, %13, %15 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %7, %13, %15 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %8, %14, %15 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %9, %14, %15 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %10, %14, %15 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %11, %14, %15 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
            : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11])
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(w));
    } else if constexpr (W == 2) {
        asm("s_nop 1\n\t"
            "v_fmac_f32_dpp %0, %8, %10 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %8, %10 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %2, %8, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %3, %8, %10 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %4, %9, %10 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %5, %9, %10 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %6, %9, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %7, %9, %10 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
            : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7])
            : "v"(x[0]), "v"(x[1]), "v"(w));
    } else {
        asm("s_nop 1\n\t"
            "v_fmac_f32_dpp %0, %4, %5 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %4, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %2, %4, %5 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %3, %4, %5 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
            : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])
            : "v"(x[0]), "v"(w));
    }
}

#ifndef HSR_FWD_BATCH_48
#define HSR_FWD_BATCH_48 152     // 33..48 channels, four workgroups per CU (4 x 40 720 B of LDS)
#endif
#ifndef HSR_FWD_BATCH_WIDE
#define HSR_FWD_BATCH_WIDE 144   // 81..112 channels (PF), two workgroups per CU (the 128-channel instantiation fits 128)
#endif
template <int KC>
struct FwdCfg {
    // LDS per staged splat: 32 (x, y, A, B, C, opacity, r, g) + 4 * round4(KC + 2) (features, b, depth)
    // 33..80 channels: as many entries per batch as three (four up to 48 channels) workgroups per CU leave room for — every batch costs
    // two workgroup barriers, a list publication and a round of exposed latency (128 -> 144 entries, 152 up to 48 channels: K = 48
    // 0.278 -> 0.263 ms, K = 64 0.366 -> 0.354, K = 74 0.364 -> 0.351, 1920x1080 / 2M / K = 74 1.260 -> 1.215); the 80-channel instantiation fits 136
    static constexpr int BATCH = KC <= 32 ? 256 : (KC <= 48 ? HSR_FWD_BATCH_48 : (KC <= 64 ? 144 : (KC == 74 ? 144 : (KC <= 80 ? 136 : 64))));
};

// KC: semantic channels handled by this launch (0 = none).  BASE: also produce colour/depth/median/
// opacity/final_T/n_contrib.  MASK: non-semantic variant, writes mask = sum(alpha*T).
// ALIGNED: the launch covers whole feature rows of an even K (c0 == 0, KC == K): rows are 8-byte
// aligned and are fetched as float2.
// SUB: the 16-lane groups of a wave own 4x4 sub-blocks and walk their own lists (hsr_tile_common.h) instead of the wave's
// quadrant list.
// PF: the next batch's semantic rows are not parked in KC registers while the current batch is blended; one word per 64-byte
// line of each row is TOUCHED instead (the loads land in a handful of registers and pull the lines into this XCD's L2), and the rows
// are read for real — L2 hits — right before they are staged.  K = 74: 238 -> ~160 registers, i.e. three waves per SIMD instead of
// two, in a kernel whose blend loop is bound by instruction issue.
template <int KC, bool BASE, bool MASK, bool ALIGNED, bool SUB = false, bool PF = false>
__global__ void __launch_bounds__(256, (SUB && KC <= 26) ? (KC == 16 ? 5 : 4) : (PF ? (KC <= 48 ? 4 : (KC <= 80 ? 3 : 2)) : 1)) render_fwd_kernel(RenderFwdArgs a, int c0)
{
    // SUB: 240 splats per batch keep the 16 lists + records of K = 26 under 40 KB (four workgroups per CU); K = 16 fits five (31 KB, 96
    // registers: forward 0.147 -> 0.142 ms); K = 26 with touched rows + quad-shared rows at five: 0.177 vs 0.177 ms, not taken.  (PF at K = 26 with
    // 184-splat batches and five waves per SIMD was measured too: 96 registers with 5 spills, 0.174 vs 0.165 ms — not taken.)
    constexpr int BATCH = (SUB && KC <= 32) ? (KC > 26 ? 200 : 240) : ((PF && KC > 80) ? (KC <= 112 ? HSR_FWD_BATCH_WIDE : 128) : FwdCfg<KC>::BATCH);   // KC = 32: 200 x 176 B + lists < 40 KB; PF beyond 80 channels: two workgroups per CU, 128 x (32 + 4 KC + 8) B < 80 KB
    // per staged splat: a 32-byte record { x, y, A, B | C, opacity, r, g } (pre-scaled conic, see hsr_tile_common.h) — all the
    // alpha test needs, in two 16-byte reads at one address — and a feature row { s0 .. s(KC-1), b, depth } whose 16-byte reads
    // pair up with the packed FMAs (blue and depth ride in the row's padding at K = 26): one LDS read and one address
    // computation fewer per visit than four separate arrays
    constexpr int RW = (KC + 2 + 3) & ~3;
    // QS (sub-block lists with features): the 16 lanes of a group all visit the SAME splat, so a row read by every lane moves the
    // same 4 * RW bytes through the LDS crossbar 16 times — and at 128 bytes per cycle and CU that, not the FMAs, is what a feature
    // channel costs (measured: 2.2-2.8 CU cycles per channel and wave visit against 0.44 for the packed FMAs).  Instead each lane
    // of a QUAD reads a quarter of the row (full 16-channel groups are stored [quad lane][4]: lane q's b128 holds channels
    // q, q+4, q+8, q+12 of the group; the last, partial group stays in channel order and is read a word at a time) and the FMAs
    // take the value from the quad lane that holds it through their DPP operand (quad_perm: no extra instruction, no LDS).
    // Taken where it pays (tools/sweep.sh, 500k Gaussians): the wide per-lane kernels (PF: K = 74 0.393 -> 0.363 ms, 1920x1080 / 2M
    // 1.313 -> 1.257 ms); at K = 16 / 26 the 2 K single FMAs against K packed ones cancel the LDS saving (0.144 -> 0.158, 0.175 -> 0.176).
    constexpr bool QS = SUB && PF && KC > 0;
    constexpr int NGF = QS ? RW / 16 : 0;            // full 16-channel groups
    constexpr int REM = QS ? (RW % 16) / 4 : 0;      // words per lane of the partial group
    __shared__ float4 s_rec[BATCH * 2];
    __shared__ float4 s_row[BATCH * (RW / 4)];
    __shared__ uint8_t s_list[SUB ? 1 : 4][256];
    __shared__ uint8_t s_lcnt[4][4];
    __shared__ uint8_t s_sublist[SUB ? 16 * HSR_SUB_LSTRIDE : 4];
    __shared__ uint8_t s_subcnt[4][16];
    __shared__ int s_wdone[4];

    const int tile = HSR_TILE_OF_BLOCK(blockIdx.x, (a.W + HSR_TILE_X - 1) / HSR_TILE_X, (a.H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    if (tile >= ((a.W + HSR_TILE_X - 1) / HSR_TILE_X) * ((a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)) return;
    const int t = threadIdx.x, wv = t >> 6;
    if (a.bin.base) {   // speculative forward: the list lives where num_rendered says
        BinState bs;
        if (!hsr_bin_resolve(a.bin, *a.bin.R_dev, &bs)) {
            hsr_poison_tile(a, tile, t, BASE, c0, KC);
            return;
        }
        a.point_list = bs.vals;
        a.masks = bs.vals_unsorted;
    }
    const TileGeom tg = SUB ? tile_geom_sub(tile, a.W, a.H, t) : tile_geom(tile, a.W, a.H, t);
    const bool inside = tg.inside;
    const size_t N = (size_t)a.W * a.H;
    // the pixel's coordinates live as FLOATS through the blend loop (opaque to the compiler, which would otherwise keep the
    // integers and convert them again for every list entry); the integer pixel index is re-derived for the final stores
    float pfx = tg.pfx, pfy = tg.pfy;
    asm volatile("" : "+v"(pfx), "+v"(pfy));
    const float tile_x0 = (float)(tg.tx * HSR_TILE_X), tile_y0 = (float)(tg.ty * HSR_TILE_Y);

    unsigned long long tr_bar = 0, tr_stage = 0, tr_blend = 0, tr_iters = 0, tr_pub = 0;
    (void)tr_pub;
    const long long tr_t0 = TRF_NOW();
    long long tr_t1 = tr_t0;
    (void)tr_bar; (void)tr_stage; (void)tr_blend; (void)tr_iters; (void)tr_t0; (void)tr_t1;
    const uint2 range = a.ranges[tile];
    const int n = (int)(range.y - range.x);

    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0, Dd = 0, Mm = 0;
    uint32_t median_at = 0;   // 1 + list position of the splat at which T crossed 0.5 (ImgState::median_pos)
    float median_D = 15.0f;   // its depth (forward.cu:511-515; default 15.0, :450)
    float S[KC > 0 ? KC : 1];
#pragma unroll
    for (int c = 0; c < (KC > 0 ? KC : 1); c++) S[c] = 0.f;
    bool done = !inside;

    // ---- software-pipelined staging registers (lane t <-> splat t of a batch) ----
    int id_next = 0;           // id of splat t of batch b+1 (loaded during batch b-1)
    float2 p_xy = {0, 0};      // record of splat t of batch b+1 (loaded during batch b)
    float4 p_co = {0, 0, 0, 0};
    float p_r = 0, p_g = 0, p_b = 0, p_d = 0;
    float p_sem[(KC > 0 && !PF) ? KC : 1];
#pragma unroll
    for (int c = 0; c < ((KC > 0 && !PF) ? KC : 1); c++) p_sem[c] = 0.f;
    constexpr int NPF = PF ? (KC + 15) / 16 + 1 : 1;   // touches per row: every 16th float and the last one
    float pf_t[NPF];
#pragma unroll
    for (int c = 0; c < NPF; c++) pf_t[c] = 0.f;
    float pf_sink = 0.f;
    int id_cur = 0;            // id of the splat whose record sits in p_* (PF: its row is fetched when it is staged)

    // Staging loads are UNCONDITIONAL (clamped indices; lanes past the end of the list fetch a duplicate they never stage) and the
    // id of batch b+2 is requested BEFORE the records of batch b+1: a load inside a divergent `if` lands in a temporary that is
    // copied into the loop-carried register at the end of the branch, and that copy — like a load that overwrites the register the
    // previous loads took their addresses from — makes hipcc wait for everything in flight right there, i.e. the "pipelined" gathers
    // were waited for where they were issued (round 1 and most of round 2: s_waitcnt vmcnt(0) 30 instructions after the loads).
    auto fetch_id = [&](int start) -> int { return (int)a.point_list[range.x + min(max(start + t, 0), n - 1)]; };
    auto load_record = [&](int id_of) {
        const size_t id = (size_t)id_of;
        id_cur = id_of;
        {
            const float4* rec = a.rec + 4 * id;
            const float4 r0 = rec[0];
            p_co = rec[1];
            p_xy = make_float2(r0.x, r0.y);
            p_d = r0.z;
            if (BASE) {
                const float4 r2 = rec[2];
                p_r = r2.x; p_g = r2.y; p_b = r2.z;
            }
            if (KC > 0 && PF) {
                const float* row = a.semantics + id * (size_t)a.K + c0;
                const int last = min(KC, a.K - c0) - 1;
#pragma unroll
                for (int c = 0; c < NPF; c++) pf_t[c] = row[min(16 * c, last)];
            } else if (KC > 0) {
                if (ALIGNED) {
                    const float2* row = reinterpret_cast<const float2*>(a.semantics + id * (size_t)KC);
#pragma unroll
                    for (int q = 0; q < KC / 2; q++) {
                        const float2 v = row[q];
                        p_sem[PF ? 0 : 2 * q] = v.x;
                        p_sem[PF ? 0 : 2 * q + 1] = v.y;
                    }
                } else {
                    const float* row = a.semantics + id * (size_t)a.K + c0;
#pragma unroll
                    for (int c = 0; c < KC; c++) p_sem[PF ? 0 : c] = (c0 + c < a.K) ? row[c] : 0.f;
                }
            }
        }
    };
    if (n > 0) {
        const int id0 = fetch_id(0);
        id_next = fetch_id(BATCH);
        load_record(id0);
    }
    // Median depth = depth of the splat at which T crossed 0.5.  The blend loop only records WHERE (median_at: one select per visit
    // instead of two); the depth is read once, right after the batch in which the crossing happened, from that batch's staged row
    // (channel KC + 1) while it is still in LDS — round 2 fetched it in the epilogue from global memory (point_list, then the splat's
    // record: two dependent gathers per pixel at the end of every tile, and 19 MB of extra FETCH_SIZE per launch at the headline).
    auto resolve_median = [&](int start) {
        if (BASE && median_at > (uint32_t)start) {   // set during this batch (T crosses 0.5 once)
            const int j = (int)median_at - 1 - start;
            constexpr int c = KC + 1;
            constexpr int pos = (QS && c / 16 < NGF) ? (16 * (c / 16) + 4 * (c % 4) + (c % 16) / 4) : c;
            median_D = reinterpret_cast<const float*>(s_row)[j * RW + pos];
        }
    };


    for (int start = 0; start < n; start += BATCH) {
        const bool wave_done = __ballot(!done) == 0ull;
        if ((t & 63) == 0) s_wdone[wv] = wave_done;
        const long long tb0 = TRF_NOW();
        (void)tb0;
        __syncthreads();  // also: everyone has finished reading the previous batch
        TRF_ADD(tr_bar, tb0);
        if (start == 0) tr_t1 = TRF_NOW();   // end of the prologue: ids, first records and the first barrier
        const long long ts0 = TRF_NOW();
        (void)ts0;
        if (s_wdone[0] & s_wdone[1] & s_wdone[2] & s_wdone[3]) break;
        const int cnt = min(BATCH, n - start);
        uint32_t qmask = 0u;
        if (t < cnt) {
            const uint32_t mask16 = subblock_mask(p_xy.x, p_xy.y, p_co.x, p_co.y, p_co.z, p_co.w, tile_x0, tile_y0, !(a.debug_flags & 16));
            qmask = SUB ? mask16 : quadrant_bits(mask16);
            a.masks[range.x + start + t] = mask16;   // the backward stages the same entries: it reads the mask instead of deriving it again
            s_rec[2 * t] = make_float4(p_xy.x, p_xy.y, (-0.5f * HSR_LOG2E) * p_co.x, -HSR_LOG2E * p_co.y);
            s_rec[2 * t + 1] = make_float4((-0.5f * HSR_LOG2E) * p_co.z, p_co.w, p_r, p_g);
            float4* row = &s_row[t * (RW / 4)];
            if (PF) {
                // the touches of this batch have landed long ago (they were issued a whole batch of blending earlier): consume them
                // so that they stay real loads, then fetch the row — L2 hits — and stage it 32 floats at a time (the whole row in
                // flight at once would need KC more registers than the blend loop leaves)
#pragma unroll
                for (int c = 0; c < NPF; c++) pf_sink += pf_t[c];
                const float* grow = a.semantics + (size_t)id_cur * (size_t)a.K + c0;
                // an entry no sub-block will visit (a fifth to a third of a tile's list: the reference's 3-sigma rectangle lists
                // splats whose alpha never reaches 1/255 on this tile) needs no row — except slot 0, which a group with an EMPTY list
                // reads with weight 0 (below): an unstaged row is whatever the LDS held (the per-tile sort pads with ~0 = NaN, and
                // NaN * 0 is NaN: found by a 300-case run of tests/test_gpu_fuzz.py)
                if (qmask != 0u || t == 0)
#pragma unroll
                for (int g0 = 0; g0 < RW; g0 += 32) {
                    float rv[32];
#pragma unroll
                    for (int c = 0; c < 32; c++) {
                        const int ch = g0 + c;
                        if (ch < KC) {
                            if (ALIGNED) {
                                if ((c & 1) == 0) {
                                    const float2 v = reinterpret_cast<const float2*>(grow)[ch / 2];
                                    rv[c] = v.x;
                                    if (c + 1 < 32) rv[c + 1] = v.y;
                                }
                            } else {
                                rv[c] = (c0 + ch < a.K) ? grow[ch] : 0.f;
                            }
                        } else if (ch < RW) {
                            rv[c] = ch == KC ? p_b : (ch == KC + 1 ? p_d : 0.f);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (g0 / 4 + q < RW / 4) {
                            if (QS && g0 / 16 + q / 4 < NGF) {   // full group: [quad lane][4]
                                const int b = 16 * (q / 4) + (q & 3);
                                row[g0 / 4 + q] = make_float4(rv[b], rv[b + 4], rv[b + 8], rv[b + 12]);
                            } else {
                                row[g0 / 4 + q] = make_float4(rv[4 * q], rv[4 * q + 1], rv[4 * q + 2], rv[4 * q + 3]);
                            }
                        }
                    asm volatile("" ::: "memory");   // next group's loads stay behind this group's stores
                }
            } else {
                float rv[RW];
#pragma unroll
                for (int c = 0; c < RW; c++) rv[c] = c < KC ? p_sem[(c < KC && !PF) ? c : 0] : (c == KC ? p_b : (c == KC + 1 ? p_d : 0.f));
#pragma unroll
                for (int q = 0; q < RW / 4; q++) {
                    if (QS && q / 4 < NGF) {   // full group: [quad lane][4]
                        const int b = 16 * (q / 4) + (q & 3);
                        row[q] = make_float4(rv[b], rv[b + 4], rv[b + 8], rv[b + 12]);
                    } else {
                        row[q] = make_float4(rv[4 * q], rv[4 * q + 1], rv[4 * q + 2], rv[4 * q + 3]);
                    }
                }
            }
        }
        const long long tp0 = TRF_NOW();
        (void)tp0;
        if (SUB) publish_subblock_lists(qmask, t, s_sublist, s_subcnt);
        else publish_quadrant_lists(qmask, t, s_list, s_lcnt);
        TRF_ADD(tr_pub, tp0);
        TRF_ADD(tr_stage, ts0);
        const long long tb1 = TRF_NOW();
        (void)tb1;
        __syncthreads();
        TRF_ADD(tr_bar, tb1);
        const long long tl0 = TRF_NOW();
        (void)tl0;
        // next batch's gathers go out now and land while this batch is blended
        {
            const int id_use = id_next;            // ids of batch b+1, requested a whole batch ago
            id_next = fetch_id(start + 2 * BATCH);
            load_record(id_use);
        }
        if (wave_done) continue;

        if constexpr (SUB) {
            // four lists per wave, one per 16-lane group; the wave iterates to the longest of them
            const int lane = t & 63, sb = wv * 4 + (lane >> 4);
            const int total = flatten_sublist(sb, lane, s_sublist, s_subcnt);
            const int m = __builtin_amdgcn_readfirstlane(max(max(__builtin_amdgcn_readlane(total, 0), __builtin_amdgcn_readlane(total, 16)),
                                                             max(__builtin_amdgcn_readlane(total, 32), __builtin_amdgcn_readlane(total, 48))));
            const uint8_t* list = s_sublist + sb * HSR_SUB_LSTRIDE;
            // a group past the end of its list keeps re-reading its last entry, with weight 0 (an empty list is given the always
            // staged slot 0 as its only entry): an unconditional clamped read instead of a masked one
            if (total == 0 && (lane & 15) == 0) s_sublist[sb * HSR_SUB_LSTRIDE] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int last = max(total - 1, 0);
            if constexpr (QS) {
                constexpr int NW = 4 * NGF + REM;
                constexpr int NCH = KC + (BASE ? 2 : 0);   // channels that are accumulated: features, then blue, depth
                int j_next = (int)list[0];
                for (int k = 0; k < m; k++) {
                    const int j = j_next;
                    const bool valid = k < total;
                    j_next = (int)list[min(k + 1, last)];
                    const float4 g = s_rec[2 * j];
                    const float4 h4 = s_rec[2 * j + 1];
                    const float2 co = make_float2(h4.x, h4.y);
                    const float dx = g.x - pfx, dy = g.y - pfy;
                    const float power2 = fmaf(co.x, dy * dy, fmaf(g.w, dx * dy, g.z * (dx * dx)));
                    const float alpha = fminf(0.99f, co.y * __builtin_amdgcn_exp2f(power2));
                    bool contrib = valid && !done && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                    const float test_T = T * (1.0f - alpha);
                    if (contrib && test_T < 0.0001f) {
                        done = true;
                        contrib = false;
                    }
                    if (__ballot(contrib) == 0ull) continue;
                    const float w = contrib ? alpha * T : 0.f;
                    if (BASE) {
                        C0 = fmaf(h4.z, w, C0);
                        C1 = fmaf(h4.w, w, C1);
                        if (MASK) Mm += w;
                    }
                    const float* rowf = reinterpret_cast<const float*>(s_row) + j * RW;
                    float fq[NW];
#pragma unroll
                    for (int G = 0; G < NGF; G++) {
                        const float4 f = *reinterpret_cast<const float4*>(rowf + 16 * G + 4 * (lane & 3));
                        fq[4 * G] = f.x; fq[4 * G + 1] = f.y; fq[4 * G + 2] = f.z; fq[4 * G + 3] = f.w;
                    }
#pragma unroll
                    for (int mm = 0; mm < REM; mm++) fq[4 * NGF + mm] = rowf[16 * NGF + 4 * mm + (lane & 3)];
                    // word r (either layout) holds channels 4r .. 4r+3, channel c in quad lane c % 4; one asm block per 16 channels
                    float dummy = 0.f;
                    auto acc = [&](int c) -> float& { return c < KC ? S[c < KC ? c : 0] : ((BASE && c == KC) ? C2 : ((BASE && c == KC + 1) ? Dd : dummy)); };
#pragma unroll
                    for (int b = 0; b < (NCH + 15) / 16; b++) {
                        const int nw = min(4, (NCH - 16 * b + 3) / 4);   // words of this block: constant after unrolling
                        float sv[16], xv[4];
#pragma unroll
                        for (int i = 0; i < 16; i++) sv[i] = i < 4 * nw ? acc(16 * b + i) : 0.f;
#pragma unroll
                        for (int i = 0; i < 4; i++) xv[i] = i < nw ? fq[4 * b + i] : 0.f;
                        if (nw == 4) quad_fma_words<4>(sv, xv, w);
                        else if (nw == 3) quad_fma_words<3>(sv, xv, w);
                        else if (nw == 2) quad_fma_words<2>(sv, xv, w);
                        else quad_fma_words<1>(sv, xv, w);
#pragma unroll
                        for (int i = 0; i < 16; i++)
                            if (i < 4 * nw && 16 * b + i < NCH) acc(16 * b + i) = sv[i];
                    }
                    if (BASE) {
                        if (contrib && T > 0.5f && test_T < 0.5f) median_at = (uint32_t)(start + j + 1);   // its depth: epilogue
                    }
                    if (contrib) {
                        T = test_T;
                        last_contributor = (uint32_t)(start + j + 1);
                    }
#ifdef HSR_TRACE
                    tr_iters++;
#endif
                }
                TRF_ADD(tr_blend, tl0);
                resolve_median(start);
                continue;
            }
            int j_next = (int)list[0];
            for (int k = 0; k < m; k++) {
                const int j = j_next;
                const bool valid = k < total;
                j_next = (int)list[min(k + 1, last)];
                const float4 g = s_rec[2 * j];
                const float4 h4 = s_rec[2 * j + 1];
                const float2 co = make_float2(h4.x, h4.y);
                const float dx = g.x - pfx, dy = g.y - pfy;
                const float power2 = fmaf(co.x, dy * dy, fmaf(g.w, dx * dy, g.z * (dx * dx)));
                const float alpha = fminf(0.99f, co.y * __builtin_amdgcn_exp2f(power2));
                bool contrib = valid && !done && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
                const float test_T = T * (1.0f - alpha);
                if (contrib && test_T < 0.0001f) {
                    done = true;
                    contrib = false;
                }
                if (__ballot(contrib) == 0ull) continue;
                const float w = contrib ? alpha * T : 0.f;
                if (BASE) {
                    C0 = fmaf(h4.z, w, C0);
                    C1 = fmaf(h4.w, w, C1);
                    if (MASK) Mm += w;
                }
                {
                    const float4* row = &s_row[j * (RW / 4)];
#pragma unroll
                    for (int q = 0; q < RW / 4; q++) {
                        const float4 f = row[q];
                        const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int c = 4 * q + i;
                            if (c < KC) S[c < KC ? c : 0] = fmaf(fv[i], w, S[c < KC ? c : 0]);
                            else if (BASE && c == KC) C2 = fmaf(fv[i], w, C2);
                            else if (BASE && c == KC + 1) {
                                Dd = fmaf(fv[i], w, Dd);
                                if (contrib && T > 0.5f && test_T < 0.5f) median_at = (uint32_t)(start + j + 1);   // its depth: epilogue
                            }
                        }
                    }
                }
                if (contrib) {
                    T = test_T;
                    last_contributor = (uint32_t)(start + j + 1);
                }
#ifdef HSR_TRACE
                tr_iters++;
#endif
            }
            TRF_ADD(tr_blend, tl0);
            resolve_median(start);
            continue;
        }

        // this wave's compacted list: four segments (one per staging wave), slot order preserved
        for (int seg = 0; seg < 4; seg++) {
            const int m = s_lcnt[wv][seg];
            int j_next = s_list[wv][seg * 64];
            for (int k = 0; k < m; k++) {
                const int j = j_next;   // slot fetched one iteration ahead (slot -> record are dependent LDS round trips)
                j_next = s_list[wv][seg * 64 + min(k + 1, 63)];
                const float4 g = s_rec[2 * j];
                const float4 h4 = s_rec[2 * j + 1];
                const float2 co = make_float2(h4.x, h4.y);
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
                if (BASE) {
                    C0 = fmaf(h4.z, w, C0);
                    C1 = fmaf(h4.w, w, C1);
                    if (MASK) Mm += w;
                }
                {
                    const float4* row = &s_row[j * (RW / 4)];
#pragma unroll
                    for (int q = 0; q < RW / 4; q++) {
                        const float4 f = row[q];
                        const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int c = 4 * q + i;
                            if (c < KC) S[c < KC ? c : 0] = fmaf(fv[i], w, S[c < KC ? c : 0]);
                            else if (BASE && c == KC) C2 = fmaf(fv[i], w, C2);
                            else if (BASE && c == KC + 1) {
                                Dd = fmaf(fv[i], w, Dd);
                                if (contrib && T > 0.5f && test_T < 0.5f) median_at = (uint32_t)(start + j + 1);   // its depth: epilogue
                            }
                        }
                    }
                }
                if (contrib) {
                    T = test_T;
                    last_contributor = (uint32_t)(start + j + 1);
                }
            }
        }
        resolve_median(start);
    }

    if (PF && pf_sink == 1.2345678e-30f) T = pf_sink;   // never true for data that matters; keeps the touch loads alive
#ifdef HSR_TRACE
    const long long tr_t2 = clock64();   // end of the batch loop
#endif
    if (inside) {
        const size_t pix_id = (size_t)a.W * (size_t)(int)pfy + (size_t)(int)pfx;
        if (BASE) {
            a.final_T[pix_id] = T;
            a.n_contrib[pix_id] = last_contributor;
            a.median_pos[pix_id] = median_at;
            a.out_color[pix_id] = C0;
            a.out_color[N + pix_id] = C1;
            a.out_color[2 * N + pix_id] = C2;
            a.out_depth[pix_id] = Dd;
            a.out_median_depth[pix_id] = median_D;   // resolve_median()
            a.out_opacity[pix_id] = 1.0f - T;
            if (MASK) a.out_mask[pix_id] = Mm;
        }
        if (KC > 0) {
#pragma unroll
            for (int c = 0; c < KC; c++)
                if (c0 + c < a.K) a.out_semantic[(size_t)(c0 + c) * N + pix_id] = S[c];
        }
    }
#ifdef HSR_TRACE
    if ((t & 63) == 0 && SUB) {
        const int wid = tile * 4 + wv;
        if (wid < 16384) {
            unsigned long long* o = g_hsr_trace_fwd + (size_t)wid * 8;
            o[0] = (unsigned long long)(clock64() - tr_t0);   // total
            o[1] = (unsigned long long)(tr_t1 - tr_t0);       // prologue
            o[2] = tr_bar; o[3] = tr_stage; o[4] = tr_blend;  // inside workgroup barriers / staging work / blend loops
            o[5] = (unsigned long long)(clock64() - tr_t2);   // epilogue (output stores)
            o[6] = tr_iters;
            o[7] = tr_pub;                                    // of the staging work: publishing the sixteen sub-block lists
        }
    }
#endif
}

}  // namespace

int hsr_launch_render_forward(const RenderFwdArgs& a, hipStream_t stream)
{
    const dim3 grid(HSR_GRID_OF_TILES((a.W + HSR_TILE_X - 1) / HSR_TILE_X, (a.H + HSR_TILE_Y - 1) / HSR_TILE_Y)), block(256);
    // Default for K <= 28: the per-lane kernel on 4x4 sub-block lists (SUB).  Measured at the headline workload (500k
    // Gaussians, 1200x680, K = 26): 0.18 ms against 0.22 ms for the same kernel on quadrant lists (HSR_FWD_IMPL=valu, kept
    // in the ablate build for A/B timing) and 0.27 ms for round 1's pair-pipelined matrix-core kernel (EXPERIMENTS.md §4: the ~25 VALU
    // instructions that evaluate alpha per list entry dominate, the matrix cores only take the 15 packed FMAs behind them, and every list
    // entry has to go through the pair; removed in round 3 — it predates the saved sub-block masks and median positions the backward reads).
#ifdef HSR_ABLATE
    static const char* impl = getenv("HSR_FWD_IMPL");
    static const bool force_valu = impl && !strcmp(impl, "valu");   // quadrant lists, per-lane accumulators for every K: ablate build only
#else
    constexpr bool force_valu = false;   // the product renders on sub-block lists; HSR_FWD_IMPL selects nothing here
#endif
#ifdef HSR_ABLATE
    // round 3's sub-block forward with the channel sums on the fp32 matrix cores (experiments/hsr_render_fwd_mma.hip): parity-green and
    // slower at every width — fp32 MFMA and plain VALU FMA both run ~34 MAC per cycle and SIMD (EXPERIMENTS.md §9b)
    static const bool use_mma = impl && !strcmp(impl, "mma");
    if (use_mma && hsr_launch_render_forward_mma(a, stream)) return HSR_OK;
#endif
    if (!a.semantic) {
#ifdef HSR_ABLATE
        if (force_valu) {
            render_fwd_kernel<0, true, true, false, false><<<grid, block, 0, stream>>>(a, 0);
            return HSR_OK;
        }
#endif
        render_fwd_kernel<0, true, true, false, true><<<grid, block, 0, stream>>>(a, 0);
        return HSR_OK;
    }
    // Wide trees.  Round 1 accumulated 29 <= K <= 124 on the matrix cores (experiments/hsr_render_fwd_wide.hip; HSR_FWD_IMPL=wide in the
    // ablate build still selects it).  Since round 2 the per-lane kernel on sub-block lists takes every K: with the rows of the next batch
    // touched into L2 instead of parked in registers (PF) it runs at three or four waves per SIMD up to 80 channels, and with quad-shared
    // rows it also wins beyond (tools/fwd_pf_max.sh, 500k Gaussians, matrix-core -> per-lane: K = 90 0.644 -> 0.557 ms, K = 102
    // 0.712 -> 0.617, K = 124 0.732 -> 0.695; profiles/r02_fwd_generic_k.log for the narrower ones).
    // (K = 74 and K = 102 — the reference's large ScanNet tree and its flat Replica label set — have instantiations of their exact width
    // below: rows fetched as aligned float2, no padding channels: K = 74 0.371 vs 0.430 ms through the 80-channel kernel)
#ifdef HSR_ABLATE
    static const bool prefer_wide = impl && !strcmp(impl, "wide");
    if (prefer_wide && hsr_launch_render_forward_wide(a, stream)) return HSR_OK;
#endif
    if (!force_valu && a.K >= 27 && a.K <= 128 && a.K != 74 && a.K != 102) {
        if (a.K <= 32) render_fwd_kernel<32, true, false, false, true, true><<<grid, block, 0, stream>>>(a, 0);
        else if (a.K <= 48) render_fwd_kernel<48, true, false, false, true, true><<<grid, block, 0, stream>>>(a, 0);
        else if (a.K <= 64) render_fwd_kernel<64, true, false, false, true, true><<<grid, block, 0, stream>>>(a, 0);
        else if (a.K <= 80) render_fwd_kernel<80, true, false, false, true, true><<<grid, block, 0, stream>>>(a, 0);
        else if (a.K <= 96) render_fwd_kernel<96, true, false, false, true, true><<<grid, block, 0, stream>>>(a, 0);
        else if (a.K <= 112) render_fwd_kernel<112, true, false, false, true, true><<<grid, block, 0, stream>>>(a, 0);
        else render_fwd_kernel<128, true, false, false, true, true><<<grid, block, 0, stream>>>(a, 0);
        return HSR_OK;
    }
    if (!force_valu) {
        switch (a.K) {
        case 0: render_fwd_kernel<0, true, false, false, true><<<grid, block, 0, stream>>>(a, 0); return HSR_OK;
        case 16: render_fwd_kernel<16, true, false, true, true><<<grid, block, 0, stream>>>(a, 0); return HSR_OK;   // ScanNet tree
        case 26: render_fwd_kernel<26, true, false, true, true><<<grid, block, 0, stream>>>(a, 0); return HSR_OK;   // Replica tree
        case 74: {   // ScanNet large tree
#ifdef HSR_ABLATE
            static const bool no_pf = getenv("HSR_FWD_PF") && !strcmp(getenv("HSR_FWD_PF"), "0");   // A/B selector, ablate build only: rows parked in registers
            if (no_pf) {
                render_fwd_kernel<74, true, false, true, true><<<grid, block, 0, stream>>>(a, 0);
                return HSR_OK;
            }
#endif
            render_fwd_kernel<74, true, false, true, true, true><<<grid, block, 0, stream>>>(a, 0);
            return HSR_OK;
        }
        case 102:   // Replica flat label set
            render_fwd_kernel<102, true, false, true, true, true><<<grid, block, 0, stream>>>(a, 0);
            return HSR_OK;
        default:   // K <= 26 or K > 128: 32-channel chunks; the first chunk also produces the base outputs
            render_fwd_kernel<32, true, false, false, true><<<grid, block, 0, stream>>>(a, 0);
            for (int c0 = 32; c0 < a.K; c0 += 32) render_fwd_kernel<32, false, false, false, true><<<grid, block, 0, stream>>>(a, c0);
            return HSR_OK;
        }
    }
#ifdef HSR_ABLATE
    // HSR_FWD_IMPL=valu: quadrant lists, per-lane accumulators
    switch (a.K) {
    case 0: render_fwd_kernel<0, true, false, false><<<grid, block, 0, stream>>>(a, 0); break;
    case 16: render_fwd_kernel<16, true, false, true><<<grid, block, 0, stream>>>(a, 0); break;   // ScanNet tree
    case 26: render_fwd_kernel<26, true, false, true><<<grid, block, 0, stream>>>(a, 0); break;   // Replica tree
    case 74: render_fwd_kernel<74, true, false, true><<<grid, block, 0, stream>>>(a, 0); break;   // ScanNet large tree
    case 102: render_fwd_kernel<102, true, false, true><<<grid, block, 0, stream>>>(a, 0); break; // Replica flat
    default:
        // any other K: 32-channel chunks; the first chunk also produces the base outputs
        render_fwd_kernel<32, true, false, false><<<grid, block, 0, stream>>>(a, 0);
        for (int c0 = 32; c0 < a.K; c0 += 32) render_fwd_kernel<32, false, false, false><<<grid, block, 0, stream>>>(a, c0);
        break;
    }
#endif
    return HSR_OK;
}
