// hsr_sort.hip — stable LSD radix sort of (u64 key, u32 value) pairs for gfx950.
//
// Replaces cub::DeviceRadixSort::SortPairs as the reference calls it (rasterizer_impl.cu:307-312,
// :570-575): keys are (tile << 32) | float_bits(depth), sorted on bits [0, 32 + ceil-log2(#tiles)),
// STABLE, so instances with equal (tile, depth) keep emission order (ascending Gaussian index).
// The result is therefore uniquely determined — bit-exact by construction, not by tolerance.
//
// Structure per 8-bit pass (HBM-bound integer work; no MFMA):
//   1. hist:    each block counts the digits of its 4096-item tile in LDS -> hist[digit][block]
//   2. rowscan: one wave per digit scans that digit's per-block counts; the 256-entry scan across
//               digits is redone by every scatter block in LDS
//   3. scatter: each WAVE owns a contiguous 1024-item chunk of the tile, processed as 16 rounds of
//               64 consecutive items.  Equal digits inside a round are found with 8 wave ballots
//               (64-bit masks, v_cmp + s_and), the rank among them is a popcount of the lower-lane
//               mask, and per-wave running digit counters live in LDS.  Order (wave, round, lane) =
//               input order, so the scatter is stable.
#include "hsr_common.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096

__global__ void __launch_bounds__(SORT_THREADS) sort_hist_kernel(const uint64_t* __restrict__ keys, int n, int shift,
                                                                 int nblocks, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; i++) {
        const int j = base + i * SORT_THREADS + threadIdx.x;
        if (j < n) atomicAdd(&h[(uint32_t)(keys[j] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// One wave per digit: exclusive scan of that digit's per-block counts in place (row d of the
// digit-major table) and the digit's total -> totals[d].  The scan ACROSS digits (256 entries) is
// redone by every scatter block in LDS, which removes a serial whole-table scan from the pass.
__global__ void __launch_bounds__(64) sort_rowscan_kernel(uint32_t* __restrict__ hist, int nblocks,
                                                           uint32_t* __restrict__ totals)
{
    const int lane = threadIdx.x;
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 64) {
        const int j = base + lane;
        const uint32_t v = j < nblocks ? row[j] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (j < nblocks) row[j] = carry + inc - v;
        carry += __shfl(inc, 63);
    }
    if (lane == 0) totals[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(SORT_THREADS) sort_scatter_kernel(const uint64_t* __restrict__ kin,
                                                                    const uint32_t* __restrict__ vin,
                                                                    uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                                    int n, int shift, int nblocks,
                                                                    const uint32_t* __restrict__ hist_scanned,
                                                                    const uint32_t* __restrict__ totals)
{
    __shared__ uint32_t wcnt[4][256];  // per-wave running digit counts, then global bases
    __shared__ uint32_t dig_base[256]; // exclusive scan of the digit totals
    __shared__ uint32_t wtot[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; i++) wcnt[i][threadIdx.x] = 0;
    __syncthreads();

    const int base = blockIdx.x * SORT_TILE + w * (64 * SORT_ITEMS);
    uint64_t key[SORT_ITEMS];
    uint32_t val[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t* mycnt = wcnt[w];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const int j = base + r * 64 + lane;
        const bool valid = j < n;
        key[r] = valid ? kin[j] : 0ull;
        val[r] = valid ? vin[j] : 0u;
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const int j = base + r * 64 + lane;
        const bool valid = j < n;
        const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t before = __popcll(m & lt_mask);
        const uint32_t prior = valid ? mycnt[d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) mycnt[d] = prior + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
        rank[r] = prior + before;
    }
    {
        // exclusive scan over the 256 digit totals (thread t <-> digit t)
        const uint32_t tv = totals[threadIdx.x];
        uint32_t inc = tv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();  // also: all waves finished their ranking rounds
        uint32_t woff = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) woff += i < w ? wtot[i] : 0u;
        dig_base[threadIdx.x] = woff + inc - tv;
    }
    {
        // thread t owns digit t: wave-exclusive offsets + this block's global base for the digit
        const uint32_t c0 = wcnt[0][threadIdx.x], c1 = wcnt[1][threadIdx.x], c2 = wcnt[2][threadIdx.x];
        const uint32_t g = dig_base[threadIdx.x] + hist_scanned[(size_t)threadIdx.x * nblocks + blockIdx.x];
        wcnt[0][threadIdx.x] = g;
        wcnt[1][threadIdx.x] = g + c0;
        wcnt[2][threadIdx.x] = g + c0 + c1;
        wcnt[3][threadIdx.x] = g + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const int j = base + r * 64 + lane;
        if (j < n) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            const uint32_t pos = wcnt[w][d] + rank[r];
            kout[pos] = key[r];
            vout[pos] = val[r];
        }
    }
}

}  // namespace

uint32_t hsr_sort_hist_entries(int R)
{
    const int nblocks = (R + SORT_TILE - 1) / SORT_TILE;
    return 256u * (uint32_t)(nblocks > 0 ? nblocks : 1) + 256u;  // per-block counts + 256 digit totals
}

// Sorts the R pairs on key bits [0, end_bit).  The input must already be in the buffer pair that
// makes the LAST pass land in (b.keys, b.vals): (keys_unsorted, vals_unsorted) when the pass count
// is odd, (keys, vals) when it is even — see hsr_sort_input_is_unsorted_buffer().
int hsr_launch_sort_pairs(BinState& b, int R, int end_bit, hipStream_t stream)
{
    if (R <= 0) return HSR_OK;
    const int passes = (end_bit + 7) / 8;
    const int nblocks = (R + SORT_TILE - 1) / SORT_TILE;
    uint32_t* totals = b.hist + (size_t)256 * nblocks;
    uint64_t* ka = (passes & 1) ? b.keys_unsorted : b.keys;
    uint32_t* va = (passes & 1) ? b.vals_unsorted : b.vals;
    uint64_t* kb = (passes & 1) ? b.keys : b.keys_unsorted;
    uint32_t* vb = (passes & 1) ? b.vals : b.vals_unsorted;
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        sort_hist_kernel<<<nblocks, SORT_THREADS, 0, stream>>>(ka, R, shift, nblocks, b.hist);
        sort_rowscan_kernel<<<256, 64, 0, stream>>>(b.hist, nblocks, totals);
        sort_scatter_kernel<<<nblocks, SORT_THREADS, 0, stream>>>(ka, va, kb, vb, R, shift, nblocks, b.hist, totals);
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    return HSR_OK;
}
