// hsr_wave_reduce.h — wave64 "transposing" multi-value reduction for gfx950 (used by the backward tile kernels).
#pragma once
#include <hip/hip_runtime.h>

// ---- cross-lane helpers (gfx950) ----
typedef unsigned uint2v __attribute__((ext_vector_type(2)));

// lanes 0-31 <- x.lo + x.hi ; lanes 32-63 <- y.lo + y.hi
__device__ __forceinline__ float pair32(float x, float y)
{
    const uint2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows (16 lanes) with bit4 = 0 <- x.row(2i) + x.row(2i+1) ; bit4 = 1 <- y.row(2i) + y.row(2i+1)
__device__ __forceinline__ float pair16(float x, float y)
{
    const uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
// lanes with `bit` clear keep x (+ partner's x), lanes with it set keep y (+ partner's y)
template <int CTRL>
__device__ __forceinline__ float pair_dpp(float x, float y, bool bit)
{
    const float keep = bit ? y : x;
    const float send = bit ? x : y;
    return keep + dpp_mov<CTRL>(send);
}

constexpr int DPP_ROW_ROR8 = 0x128;
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_QUAD_XOR2 = 0x4E;  // quad_perm [2,3,0,1]
constexpr int DPP_QUAD_XOR1 = 0xB1;  // quad_perm [1,0,3,2]

// Sums each of the N per-lane values over the 64 lanes of the wave; lane l returns the total of
// v[reduce_slot(l)] (don't-care where that index is >= N).  Stage order is chosen by instruction cost on gfx950: the four
// in-row stages (quad_perm xor 1, xor 2, row_half_mirror, row_ror:8) are full-rate DPP adds and run
// while there are many registers; the two cross-row stages (v_permlane16_swap, v_permlane32_swap —
// slower, with hazard wait states) run last on the 3 and 2 registers that are left.
template <int M>
__device__ __forceinline__ float elem_or_zero(const float (&x)[M], int i)
{
    return i < M ? x[i < M ? i : 0] : 0.f;
}

// Two DPP pair-sums whose "which operand do I keep" select is done by the DPP bank mask instead of v_cndmask:
//   r = x + perm(x) in every lane, then r = y + perm(y) written only in the banks (4-lane groups of a row) where the
//   select bit is set.  row_half_mirror pairs lanes across bit 2 (banks 1,3 = 0xA), row_ror:8 across bit 3 (banks 2,3 =
//   0xC).  2 VALU instructions per pair instead of 3 — the backward is VALU-issue bound.  Inline asm because the
//   compiler does not emit bank-masked DPP adds; the leading s_nop covers the VALU-write -> DPP-read hazard (2 wait
//   states on gfx9) for operands produced just before the block (hipcc pads nothing inside asm).
#define HSR_BANKED_PAIR2(NAME, CTRL, MASK)                                                                   \
    __device__ __forceinline__ void NAME(float x0, float y0, float x1, float y1, float& r0, float& r1)        \
    {                                                                                                          \
        asm volatile("s_nop 1\n\t"                                                                             \
                     "v_add_f32_dpp %0, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
                     "v_add_f32_dpp %1, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
                     "v_add_f32_dpp %0, %3, %3 " CTRL " row_mask:0xf bank_mask:" MASK "\n\t"                   \
                     "v_add_f32_dpp %1, %5, %5 " CTRL " row_mask:0xf bank_mask:" MASK                          \
                     : "=&v"(r0), "=&v"(r1)                                                                    \
                     : "v"(x0), "v"(y0), "v"(x1), "v"(y1));                                                    \
    }
HSR_BANKED_PAIR2(pair2_half_mirror, "row_half_mirror", "0xa")
HSR_BANKED_PAIR2(pair2_ror8, "row_ror:8", "0xc")

#ifndef HSR_REDUCE_SWAP_FIRST
#define HSR_REDUCE_SWAP_FIRST 0
#endif

template <int N>
__device__ __forceinline__ float wave_reduce_transpose(const float (&v)[N], int lane)
{
    static_assert(N >= 1 && N <= 64, "at most one value per lane");
    constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2, N5 = (N4 + 1) / 2;
    static_assert((N5 + 1) / 2 == 1, "six stages reduce to one register");
    float a[N1], b[N2], c[N3], d[N4], e[N5];
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#if HSR_REDUCE_SWAP_FIRST
    // cross-row stages first: a v_permlane*_swap pair-sum is 2 instructions (swap + add) against 3 for a DPP
    // pair-sum (2 selects + add), so the stages that touch the most registers use the swaps
#pragma unroll
    for (int i = 0; i < N1; i++) a[i] = pair32(v[2 * i], elem_or_zero(v, 2 * i + 1));
#pragma unroll
    for (int i = 0; i < N2; i++) b[i] = pair16(a[2 * i], elem_or_zero(a, 2 * i + 1));
#pragma unroll
    for (int i = 0; i < N3; i++) c[i] = pair_dpp<DPP_ROW_ROR8>(b[2 * i], elem_or_zero(b, 2 * i + 1), b3);
    // row_half_mirror pairs l with 7-l (flips bits 0..2): it has to come before the xor-2 / xor-1 stages
#pragma unroll
    for (int i = 0; i < N4; i++) d[i] = pair_dpp<DPP_ROW_HALF_MIRROR>(c[2 * i], elem_or_zero(c, 2 * i + 1), b2);
#pragma unroll
    for (int i = 0; i < N5; i++) e[i] = pair_dpp<DPP_QUAD_XOR2>(d[2 * i], elem_or_zero(d, 2 * i + 1), b1);
    return pair_dpp<DPP_QUAD_XOR1>(e[0], elem_or_zero(e, 1), b0);
#else
    // Stage order by cost: the two pairings whose select bit is a DPP BANK bit (row_half_mirror <-> bit 2,
    // row_ror:8 <-> bit 3) need no v_cndmask (2 instructions per pair) and run first, on the most registers;
    // quad_perm xor 1 / xor 2 (3 instructions per pair) follow; the permlane swaps (slow) see 2 + 1 pairs.
    // Validity: row_half_mirror pairs l with 7-l (flips bits 0..2), so it precedes every stage whose select bit
    // is one of those; all later pairings (xor 8, 1, 2, 16, 32) join lanes that agree on the earlier select bits.
#pragma unroll
    for (int i = 0; i + 1 < N1; i += 2)
        pair2_half_mirror(v[2 * i], elem_or_zero(v, 2 * i + 1), v[2 * i + 2], elem_or_zero(v, 2 * i + 3), a[i], a[i + 1]);
    if (N1 & 1) a[N1 - 1] = pair_dpp<DPP_ROW_HALF_MIRROR>(v[2 * (N1 - 1)], elem_or_zero(v, 2 * (N1 - 1) + 1), b2);
#pragma unroll
    for (int i = 0; i + 1 < N2; i += 2)
        pair2_ror8(a[2 * i], elem_or_zero(a, 2 * i + 1), a[2 * i + 2], elem_or_zero(a, 2 * i + 3), b[i], b[i + 1]);
    if (N2 & 1) b[N2 - 1] = pair_dpp<DPP_ROW_ROR8>(a[2 * (N2 - 1)], elem_or_zero(a, 2 * (N2 - 1) + 1), b3);
#pragma unroll
    for (int i = 0; i < N3; i++) c[i] = pair_dpp<DPP_QUAD_XOR1>(b[2 * i], elem_or_zero(b, 2 * i + 1), b0);
#pragma unroll
    for (int i = 0; i < N4; i++) d[i] = pair_dpp<DPP_QUAD_XOR2>(c[2 * i], elem_or_zero(c, 2 * i + 1), b1);
#pragma unroll
    for (int i = 0; i < N5; i++) e[i] = pair16(d[2 * i], elem_or_zero(d, 2 * i + 1));
    return pair32(e[0], elem_or_zero(e, 1));
#endif
}

// which value a lane holds after wave_reduce_transpose (select bits in stage order)
__device__ __forceinline__ int reduce_slot(int l)
{
#if HSR_REDUCE_SWAP_FIRST
    // stage order b5, b4, b3, b2, b1, b0: bit reversal
    return ((l & 1) << 5) | ((l & 2) << 3) | ((l & 4) << 1) | ((l & 8) >> 1) | ((l & 16) >> 3) | ((l & 32) >> 5);
#else
    // stage order b2, b3, b0, b1, b4, b5
    return ((l >> 2) & 3) | ((l & 3) << 2) | (l & 0x30);
#endif
}

