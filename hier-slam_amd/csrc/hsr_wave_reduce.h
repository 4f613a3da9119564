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

template <int N>
__device__ __forceinline__ float wave_reduce_transpose(const float (&v)[N], int lane)
{
    static_assert(N >= 1 && N <= 64, "at most one value per lane");
    constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2, N5 = (N4 + 1) / 2;
    static_assert((N5 + 1) / 2 == 1, "six stages reduce to one register");
    float a[N1], b[N2], c[N3], d[N4], e[N5];
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
    // row_half_mirror pairs l with 7-l (flips bits 0..2), so it must come first: each later pairing
    // (xor 1, xor 2, xor 8, xor 16, xor 32) then joins lanes that agree on every earlier select bit
    for (int i = 0; i < N1; i++) a[i] = pair_dpp<DPP_ROW_HALF_MIRROR>(v[2 * i], elem_or_zero(v, 2 * i + 1), b2);
#pragma unroll
    for (int i = 0; i < N2; i++) b[i] = pair_dpp<DPP_QUAD_XOR1>(a[2 * i], elem_or_zero(a, 2 * i + 1), b0);
#pragma unroll
    for (int i = 0; i < N3; i++) c[i] = pair_dpp<DPP_QUAD_XOR2>(b[2 * i], elem_or_zero(b, 2 * i + 1), b1);
#pragma unroll
    for (int i = 0; i < N4; i++) d[i] = pair_dpp<DPP_ROW_ROR8>(c[2 * i], elem_or_zero(c, 2 * i + 1), b3);
#pragma unroll
    for (int i = 0; i < N5; i++) e[i] = pair16(d[2 * i], elem_or_zero(d, 2 * i + 1));
    return pair32(e[0], elem_or_zero(e, 1));
}

// which value a lane holds after wave_reduce_transpose: select bits in stage order b2, b0, b1, b3, b4, b5
__device__ __forceinline__ int reduce_slot(int l)
{
    return ((l >> 2) & 1) | ((l & 1) << 1) | (((l >> 1) & 1) << 2) | (l & 0x38);
}

