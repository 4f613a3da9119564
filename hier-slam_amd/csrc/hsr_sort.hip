// hsr_sort.hip — stable LSD radix sort of (u64 key, u32 value) pairs for gfx950.
//
// Replaces cub::DeviceRadixSort::SortPairs as the reference calls it (rasterizer_impl.cu:307-312,
// :570-575): keys are (tile << 32) | float_bits(depth), sorted on bits [0, 32 + ceil-log2(#tiles)),
// STABLE, so instances with equal (tile, depth) keep emission order (ascending Gaussian index).
// The result is therefore uniquely determined — bit-exact by construction, not by tolerance.
//
// Two phases (hsr_launch_sort_pairs): the radix passes below run over the tile bits only (2 passes at 1200x680
// instead of 6 over all 44 key bits); the depth order inside each tile comes from tile_sort_kernel in LDS.
// Structure per pass (HBM-bound integer work; no MFMA):
//   1. hist:    each block counts the digits of its 4096-item tile in LDS -> hist[digit][block]
//   2. rowscan: one wave per digit scans that digit's per-block counts; the 256-entry scan across
//               digits is redone by every scatter block in LDS
//   3. scatter: each WAVE owns a contiguous 1024-item chunk of the tile, processed as 16 rounds of
//               64 consecutive items.  Equal digits inside a round are found with 8 wave ballots
//               (64-bit masks, v_cmp + s_and), the rank among them is a popcount of the lower-lane
//               mask, and per-wave running digit counters live in LDS.  Order (wave, round, lane) =
//               input order, so the scatter is stable.
#include <stdlib.h>
#include <string.h>

#include "hsr_common.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096
static_assert(SORT_TILE == HSR_SORT_TILE, "hsr_common.h sizes the histogram scratch");

__global__ void __launch_bounds__(SORT_THREADS) sort_hist_kernel(const uint64_t* __restrict__ keys, int n, int shift,
                                                                 uint32_t mask, int nblocks, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; i++) {
        const int j = base + i * SORT_THREADS + threadIdx.x;
        if (j < n) atomicAdd(&h[(uint32_t)(keys[j] >> shift) & mask], 1u);
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
                                                                    int n, int shift, uint32_t mask, int nblocks,
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
        const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
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
            const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
            const uint32_t pos = wcnt[w][d] + rank[r];
            kout[pos] = key[r];
            vout[pos] = val[r];
        }
    }
}

}  // namespace

uint32_t hsr_sort_hist_entries(int R) { return hsr_sort_hist_entries_inline((uint32_t)(R > 0 ? R : 0)); }


// ---- phase 2: per-tile sort by (depth bits, Gaussian index) ----
// After the tile passes every tile's entries are contiguous and still in emission order (ascending Gaussian
// index).  A stable sort on depth inside the tile then equals the reference's stable 64-bit sort; since
// (depth, index) pairs are unique inside a tile, sorting the composite key (depth << 32) | index with ANY
// network gives exactly that order.  One workgroup per tile:
//   n <= TS_MAX : bitonic network on composite keys in LDS (padding = +inf keys);
//   n  > TS_MAX : block-local stable LSD radix on the 32 depth bits — preceded by the Gaussian-index bytes when the
//                 segment does not arrive in emission order (direct binning) — ping-ponging inside the tile's own
//                 segment of the two global buffer pairs (segments of different tiles are disjoint).
constexpr int TS_MAX = 2048;
constexpr int TW_MAX = 1024;   // tiles of at most this many entries: one wave each, elements in registers (tile_sort_wave_kernel)

__device__ __forceinline__ void ts_block_radix(uint64_t* ka, uint32_t* va, uint64_t* kb, uint32_t* vb, int r0, int n,
                                               uint32_t* hist /*[256]*/, uint32_t (*wcnt)[256], int gid_passes)
{
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    // LSD over the composite (depth bits, Gaussian index): index bytes first, then the four depth bytes
    const int npass = gid_passes + 4;
    for (int pass = 0; pass < npass; pass++) {
        const bool on_gid = pass < gid_passes;
        const int shift = 8 * (on_gid ? pass : pass - gid_passes);
        hist[t] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) wcnt[i][t] = 0;
        __syncthreads();
        for (int i = t; i < n; i += 256)
            atomicAdd(&hist[((on_gid ? va[r0 + i] : (uint32_t)ka[r0 + i]) >> shift) & 255u], 1u);
        __syncthreads();
        // exclusive scan of the 256 counts (thread t <-> digit t)
        {
            const uint32_t v = hist[t];
            uint32_t inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t x = __shfl_up(inc, o);
                if (lane >= o) inc += x;
            }
            __shared__ uint32_t wt[4];
            if (lane == 63) wt[w] = inc;
            __syncthreads();
            uint32_t off = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) off += i < w ? wt[i] : 0u;
            hist[t] = off + inc - v;  // running base of digit t
        }
        __syncthreads();
        for (int c = 0; c < n; c += 256) {
            const int i = c + t;
            const bool valid = i < n;
            const uint64_t k = valid ? ka[r0 + i] : 0ull;
            const uint32_t v = valid ? va[r0 + i] : 0u;
            const uint32_t d = ((on_gid ? v : (uint32_t)k) >> shift) & 255u;
            uint64_t m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const uint32_t before = __popcll(m & lt);
            if (valid && before == 0) wcnt[w][d] = (uint32_t)__popcll(m);
            __syncthreads();
            if (valid) {
                uint32_t pos = hist[d] + before;
#pragma unroll
                for (int i2 = 0; i2 < 4; i2++) pos += i2 < w ? wcnt[i2][d] : 0u;
                kb[r0 + pos] = k;
                vb[r0 + pos] = v;
            }
            __syncthreads();
            hist[t] += wcnt[0][t] + wcnt[1][t] + wcnt[2][t] + wcnt[3][t];
#pragma unroll
            for (int i2 = 0; i2 < 4; i2++) wcnt[i2][t] = 0;
            __syncthreads();
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
        __threadfence_block();
    }
    if (npass & 1) {  // an odd number of passes ends in the alternate pair: bring the segment home
        __syncthreads();
        for (int i = t; i < n; i += 256) {
            kb[r0 + i] = ka[r0 + i];
            vb[r0 + i] = va[r0 + i];
        }
    }
}

// ---- per-tile sort, TWO waves per tile (round 3) ----
// At the headline workload every one of the 3 225 tiles holds 263..448 entries: tile_sort_wave_kernel gives each to ONE wave with 8
// elements per lane, and the launch — 3.15 waves per SIMD, all resident, every wave running the same 45-step network at a third of the
// SIMD's issue rate with a dependent LDS round trip in 21 of the steps — lasts as long as one wave does (27.8 us; waves average 19 us of
// life, profiles/r02_h_final.json).  Two waves per tile halve the elements per lane (E = 4 for 257..512 entries: 128 lanes x 4), so every
// step is half as long and twice as many waves hide each other's LDS round trips.  Thread = tid128 of the pair, element e = E * tid128 + r;
// a step with stride j >= E exchanges with thread tid128 ^ (j / E): inside the wave for j / E < 64 (wave-level fence), and across the two
// waves for exactly ONE step of the whole network (k = N, j = N / 2), bracketed by two workgroup barriers (partner's stores visible;
// partner's loads done before the next step overwrites the slots).  Two pairs = two tiles per 256-thread workgroup; every pair
// passes exactly those two barriers whatever its tile holds.  16 KB of exchange buffers: all 1 613 workgroups of the headline resident.  Tiles above TP_MAX entries are left to the whole workgroup afterwards (block_sort_tile), as in
// tile_sort_wave_kernel.
constexpr int TP_MAX = 1024;   // E = 8
constexpr int TQ_MAX = 2048;   // four waves, E = 8: tiles of 1025..2048 entries (block_sort_tile)

// NT threads (2 or 4 waves) sort N = NT * E composites held E per thread; buf: E / 2 rows of NT x 16 bytes.
template <int E, int NT>
__device__ __forceinline__ void net_bitonic(uint64_t (&x)[E], int tid, ulonglong2* buf)
{
    constexpr int N = NT * E;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {
#pragma unroll
                for (int r = 0; r < E; r++) {
                    if ((r & j) == 0) {
                        const bool up = k < E ? ((r & k) == 0) : (((E * tid) & k) == 0);
                        const uint64_t a = x[r], b = x[r + j];
                        const bool sw = (a > b) == up;
                        x[r] = sw ? b : a;
                        x[r + j] = sw ? a : b;
                    }
                }
            } else {
                const int m = j / E;                   // partner thread = tid ^ m
                const bool cross = m >= 64;            // partner in another wave: one step of the network at two waves, three at four
#pragma unroll
                for (int q = 0; q < E / 2; q++) buf[q * NT + tid] = make_ulonglong2(x[2 * q], x[2 * q + 1]);
                if (cross) {
                    __syncthreads();
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                const bool keep_min = ((tid & m) == 0) == (((E * tid) & k) == 0);
#pragma unroll
                for (int q = 0; q < E / 2; q++) {
                    const ulonglong2 y = buf[q * NT + (tid ^ m)];
                    const uint64_t a0 = x[2 * q], a1 = x[2 * q + 1];
                    x[2 * q] = keep_min ? (a0 < y.x ? a0 : y.x) : (a0 > y.x ? a0 : y.x);
                    x[2 * q + 1] = keep_min ? (a1 < y.y ? a1 : y.y) : (a1 > y.y ? a1 : y.y);
                }
                // the next exchange's stores stay behind these loads
                if (cross) {
                    __syncthreads();
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
        }
    }
}

// One tile, the whole workgroup (256 threads).  comp / hist / wcnt: the workgroup's LDS scratch.
__device__ __forceinline__ void block_sort_tile(int tile, const uint2* __restrict__ ranges, uint64_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals, uint64_t* __restrict__ keys_alt,
                                                uint32_t* __restrict__ vals_alt, int gid_passes, int composite_in, uint64_t* comp,
                                                uint32_t* hist, uint32_t (*wcnt)[256])
{
    const uint2 rg = ranges[tile];
    const int r0 = (int)rg.x, n = (int)(rg.y - rg.x);
    const int t = threadIdx.x;
    // composite_in (direct binning): keys[] holds (depth bits << 32) | index per instance, the tile is given
    const uint64_t my_tile_hi = (uint64_t)tile << 32;
    if (n <= 0) return;
    if (n == 1 || n > TS_MAX) {
        if (composite_in) {
            for (int i = t; i < n; i += 256) {
                const uint64_t c = keys[r0 + i];
                keys[r0 + i] = my_tile_hi | (c >> 32);
                vals[r0 + i] = (uint32_t)c;
            }
            __syncthreads();
        }
        if (n == 1) return;
    }
    if (n > TS_MAX) {
        ts_block_radix(keys, vals, keys_alt, vals_alt, r0, n, hist, wcnt, gid_passes);  // ends in (keys, vals)
        return;
    }
    if (composite_in && n > TP_MAX) {
        // 1025..2048 composites (every tile of the 2M-Gaussian workload at 1200x680): eight per thread in registers, the four waves
        // exchange through LDS in 3 of the 66 steps — instead of the LDS network below (16 bytes read and written per compare-exchange,
        // a workgroup barrier at every stride >= 128: 0.141 ms of that workload's 1.86 ms step)
        static_assert(TQ_MAX == TS_MAX && TS_MAX * 8 >= 4 * 256 * 16, "exchange buffer = the LDS network's key array");
        uint64_t x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int e = 8 * t + r;
            x[r] = e < n ? keys[r0 + e] : ~0ull;
        }
        net_bitonic<8, 256>(x, t, reinterpret_cast<ulonglong2*>(comp));
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int e = 8 * t + r;
            if (e < n) {
                keys[r0 + e] = my_tile_hi | (x[r] >> 32);
                vals[r0 + e] = (uint32_t)x[r];
            }
        }
        return;
    }
    int N = 64;
    while (N < n) N <<= 1;
    const uint64_t tile_hi = composite_in ? my_tile_hi : (keys[r0] & 0xFFFFFFFF00000000ull);
    for (int i = t; i < N; i += 256)
        comp[i] = i < n ? (composite_in ? keys[r0 + i] : (((keys[r0 + i] & 0xFFFFFFFFull) << 32) | (uint64_t)vals[r0 + i])) : ~0ull;
    __syncthreads();
    // Bitonic network.  Thread t does compare-exchange i = t (+256 m) of a step; for strides j <= 64 the 64 exchanges of a wave
    // stay inside the wave's own 128 elements, and LDS operations of one wave complete in order, so consecutive steps with
    // j <= 64 need no workgroup barrier: 7 barriers instead of 45 for a 512-entry tile.
    bool cross = false;   // the previous step exchanged across waves
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 128 || cross) __syncthreads();
            cross = j >= 128;
            for (int i = t; i < N / 2; i += 256) {
                // i-th compare-exchange of this step: partner indices lo < hi differ in bit j
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const uint64_t a = comp[lo], b = comp[hi];
                if ((a > b) == up) {
                    comp[lo] = b;
                    comp[hi] = a;
                }
            }
        }
    }
    __syncthreads();
    for (int i = t; i < n; i += 256) {
        const uint64_t c = comp[i];
        keys[r0 + i] = tile_hi | (c >> 32);
        vals[r0 + i] = (uint32_t)c;
    }
}

__global__ void __launch_bounds__(256) tile_sort_kernel(const uint2* __restrict__ ranges, uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, uint64_t* __restrict__ keys_alt,
                                                        uint32_t* __restrict__ vals_alt, int gid_passes, int composite_in, BinDevRef ref)
{
    if (ref.base) {   // speculative forward: the arrays live where num_rendered says
        BinState bs;
        if (!hsr_bin_resolve(ref, *ref.R_dev, &bs)) return;
        keys = bs.keys; vals = bs.vals; keys_alt = bs.keys_unsorted; vals_alt = bs.vals_unsorted;
    }
    __shared__ uint64_t comp[TS_MAX];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wcnt[4][256];
    if (composite_in == 2) {   // direct binning, tiles of at most TW_MAX entries belong to tile_sort_wave_kernel
        const uint2 rg = ranges[blockIdx.x];
        if (rg.y - rg.x <= (uint32_t)TW_MAX) return;
        composite_in = 1;
    }
    block_sort_tile((int)blockIdx.x, ranges, keys, vals, keys_alt, vals_alt, gid_passes, composite_in, comp, hist, wcnt);
}

// ---- per-tile sort, one WAVE per tile, for tiles of at most 1024 entries (direct binning composites) ----
// The block-wide network above is LDS-bound: every compare-exchange is two 8-byte reads and up to two writes with 2- to
// 8-way bank conflicts at small strides, 45 steps for a 512-entry tile.  Here a lane keeps E = 2, 4 or 8 CONSECUTIVE
// elements in registers (N = 64 E): the steps with stride j < E — more than half of them — are register-only, and a step
// with j >= E exchanges whole E-element blocks with lane ^ (j / E) through a conflict-free LDS buffer (16-byte accesses,
// lanes contiguous) and keeps the smaller or the larger element of every pair.  One wave, so no barriers at all; four
// tiles per 256-thread workgroup.  Tiles above TW_MAX entries are left to tile_sort_kernel (which skips the others).
// E = 16 (513..1024 entries, 32 key registers) came late in round 2: on the anisotropic and the 1920x1080 / 2M workloads most tiles
// hold 500-1000 entries and the block-wide LDS network took 65 / 131 us per frame for them (tools/ktrace_cfg.sh).

template <int E>
__device__ __forceinline__ void wave_bitonic(uint64_t (&x)[E], int lane, ulonglong2* buf)
{
    constexpr int N = 64 * E;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {
                // both partners in this lane's registers
#pragma unroll
                for (int r = 0; r < E; r++) {
                    if ((r & j) == 0) {
                        const bool up = k < E ? ((r & k) == 0) : (((E * lane) & k) == 0);
                        const uint64_t a = x[r], b = x[r + j];
                        const bool sw = (a > b) == up;
                        x[r] = sw ? b : a;
                        x[r + j] = sw ? a : b;
                    }
                }
            } else {
                const int m = j / E;   // partner lane = lane ^ m holds the partner of every one of my elements
#pragma unroll
                for (int q = 0; q < E / 2; q++) buf[q * 64 + lane] = make_ulonglong2(x[2 * q], x[2 * q + 1]);
                // the reads below are of OTHER lanes' stores: per thread the addresses provably differ, so without a
                // wavefront-scope release/acquire pair the compiler is free to hoist them above the stores
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool keep_min = ((lane & m) == 0) == (((E * lane) & k) == 0);
#pragma unroll
                for (int q = 0; q < E / 2; q++) {
                    const ulonglong2 y = buf[q * 64 + (lane ^ m)];
                    const uint64_t a0 = x[2 * q], a1 = x[2 * q + 1];
                    x[2 * q] = keep_min ? (a0 < y.x ? a0 : y.x) : (a0 > y.x ? a0 : y.x);
                    x[2 * q + 1] = keep_min ? (a1 < y.y ? a1 : y.y) : (a1 > y.y ? a1 : y.y);
                }
                // ... and the next exchange's stores must stay behind these loads
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
}

template <int E>
__device__ __forceinline__ void wave_sort_tile(int r0, int n, int lane, uint64_t tile_hi, uint64_t* __restrict__ keys,
                                               uint32_t* __restrict__ vals, ulonglong2* buf)
{
    uint64_t x[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = E * lane + r;
        x[r] = e < n ? keys[r0 + e] : ~0ull;
    }
    wave_bitonic<E>(x, lane, buf);
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = E * lane + r;
        if (e < n) {
            keys[r0 + e] = tile_hi | (x[r] >> 32);
            vals[r0 + e] = (uint32_t)x[r];
        }
    }
}

// Four tiles per workgroup: each wave sorts its tile if it has at most TW_MAX entries; the larger ones of the four are then
// sorted one after the other by the whole workgroup (block_sort_tile) in the same launch.
__global__ void __launch_bounds__(256) tile_sort_wave_kernel(int T, const uint2* __restrict__ ranges, uint64_t* __restrict__ keys,
                                                             uint32_t* __restrict__ vals, uint64_t* __restrict__ keys_alt,
                                                             uint32_t* __restrict__ vals_alt, int gid_passes, int big_too, BinDevRef ref)
{
    // wave phase: per wave E/2 <= 8 rows of 64 x 16 bytes; workgroup phase: comp[TS_MAX], hist[256], wcnt[4][256]
    constexpr int RAW_WG = TS_MAX * 8 + 256 * 4 + 4 * 256 * 4, RAW_WV = 4 * 8 * 64 * 16;
    constexpr int RAW = RAW_WG > RAW_WV ? RAW_WG : RAW_WV;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[RAW];
    __shared__ int s_big[4];
    if (ref.base) {   // speculative forward: the arrays live where num_rendered says
        BinState bs;
        if (!hsr_bin_resolve(ref, *ref.R_dev, &bs)) return;
        keys = bs.keys; vals = bs.vals; keys_alt = bs.keys_unsorted; vals_alt = bs.vals_unsorted;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wv;
    int n = 0, r0 = 0;
    if (tile < T) {
        const uint2 rg = ranges[tile];
        r0 = (int)rg.x;
        n = (int)(rg.y - rg.x);
    }
    if (lane == 0) s_big[wv] = n > TW_MAX;
    if (n > 0 && n <= TW_MAX) {
        ulonglong2* buf = reinterpret_cast<ulonglong2*>(s_raw) + wv * (8 * 64);
        const uint64_t tile_hi = (uint64_t)tile << 32;
        if (n <= 128) wave_sort_tile<2>(r0, n, lane, tile_hi, keys, vals, buf);
        else if (n <= 256) wave_sort_tile<4>(r0, n, lane, tile_hi, keys, vals, buf);
        else if (n <= 512) wave_sort_tile<8>(r0, n, lane, tile_hi, keys, vals, buf);
        else wave_sort_tile<16>(r0, n, lane, tile_hi, keys, vals, buf);
    }
    if (!big_too) return;   // the larger tiles have a launch of their own (one workgroup per tile)
    __syncthreads();
    uint64_t* comp = reinterpret_cast<uint64_t*>(s_raw);
    uint32_t* hist = reinterpret_cast<uint32_t*>(s_raw + TS_MAX * 8);
    uint32_t (*wcnt)[256] = reinterpret_cast<uint32_t (*)[256]>(s_raw + TS_MAX * 8 + 256 * 4);
    for (int w = 0; w < 4; w++) {
        if (!s_big[w]) continue;
        block_sort_tile(blockIdx.x * 4 + w, ranges, keys, vals, keys_alt, vals_alt, gid_passes, 1, comp, hist, wcnt);
        __syncthreads();
    }
}

template <int E>
__device__ __forceinline__ void pair_sort_tile(int r0, int n, int tid, uint64_t tile_hi, uint64_t* __restrict__ keys,
                                               uint32_t* __restrict__ vals, ulonglong2* buf)
{
    uint64_t x[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = E * tid + r;
        x[r] = e < n ? keys[r0 + e] : ~0ull;
    }
    net_bitonic<E, 128>(x, tid, buf);
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = E * tid + r;
        if (e < n) {
            keys[r0 + e] = tile_hi | (x[r] >> 32);
            vals[r0 + e] = (uint32_t)x[r];
        }
    }
}

__global__ void __launch_bounds__(256, 7) tile_sort_pair_kernel(int T, const uint2* __restrict__ ranges, uint64_t* __restrict__ keys,
                                                             uint32_t* __restrict__ vals, uint64_t* __restrict__ keys_alt,
                                                             uint32_t* __restrict__ vals_alt, int gid_passes, int big_too, BinDevRef ref)
{
    // pair phase: per pair E/2 <= 4 rows of 128 x 16 bytes; workgroup phase: comp[TS_MAX], hist[256], wcnt[4][256]
    constexpr int RAW_WG = TS_MAX * 8 + 256 * 4 + 4 * 256 * 4, RAW_PAIR = 2 * 4 * 128 * 16;
    constexpr int RAW = RAW_WG > RAW_PAIR ? RAW_WG : RAW_PAIR;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[RAW];
    __shared__ int s_big[2];
    if (ref.base) {   // speculative forward: the arrays live where num_rendered says
        BinState bs;
        if (!hsr_bin_resolve(ref, *ref.R_dev, &bs)) return;
        keys = bs.keys; vals = bs.vals; keys_alt = bs.keys_unsorted; vals_alt = bs.vals_unsorted;
    }
    const int pair = threadIdx.x >> 7, tid = threadIdx.x & 127;
    const int tile = blockIdx.x * 2 + pair;
    int n = 0, r0 = 0;
    if (tile < T) {
        const uint2 rg = ranges[tile];
        r0 = (int)rg.x;
        n = (int)(rg.y - rg.x);
    }
    if (tid == 0) s_big[pair] = n > TP_MAX;
    {
        ulonglong2* buf = reinterpret_cast<ulonglong2*>(s_raw) + pair * (4 * 128);
        const uint64_t tile_hi = (uint64_t)tile << 32;
        // every pair passes exactly two workgroup barriers here (inside the network, or the bare ones of a pair with nothing to sort)
        if (n > 0 && n <= 256) pair_sort_tile<2>(r0, n, tid, tile_hi, keys, vals, buf);
        else if (n > 256 && n <= 512) pair_sort_tile<4>(r0, n, tid, tile_hi, keys, vals, buf);
        else if (n > 512 && n <= TP_MAX) pair_sort_tile<8>(r0, n, tid, tile_hi, keys, vals, buf);
        else { __syncthreads(); __syncthreads(); }
    }
    if (!big_too) return;   // the larger tiles have a launch of their own (one workgroup per tile)
    __syncthreads();
    uint64_t* comp = reinterpret_cast<uint64_t*>(s_raw);
    uint32_t* hist = reinterpret_cast<uint32_t*>(s_raw + TS_MAX * 8);
    uint32_t (*wcnt)[256] = reinterpret_cast<uint32_t (*)[256]>(s_raw + TS_MAX * 8 + 256 * 4);
    for (int w = 0; w < 2; w++) {
        if (!s_big[w]) continue;
        block_sort_tile(blockIdx.x * 2 + w, ranges, keys, vals, keys_alt, vals_alt, gid_passes, 1, comp, hist, wcnt);
        __syncthreads();
    }
}

// Sorts the R pairs on key bits [0, end_bit) — the contract of the reference's cub::DeviceRadixSort::SortPairs
// call (rasterizer_impl.cu:307-312) — in two phases: stable LSD passes over the TILE bits only (bits 32..end_bit,
// at most 8 per pass), then one per-tile sort by depth (tile_sort_kernel).  ranges[] is produced between the two
// (tile boundaries do not depend on the depth order).  Input must be in the buffer pair from which the tile passes
// end in (b.keys, b.vals): hsr_sort_emit_into_sorted_buffers() tells the caller which.
int hsr_sort_tile_passes(int end_bit)
{
    const int tile_bits = end_bit - 32;
    return tile_bits <= 0 ? 0 : (tile_bits + 7) / 8;
}
bool hsr_sort_emit_into_sorted_buffers(int end_bit) { return (hsr_sort_tile_passes(end_bit) & 1) == 0; }

int hsr_launch_sort_pairs(BinState& b, int R, int end_bit, int T, uint2* ranges, hipStream_t stream)
{
    // ranges[] was zeroed by the key-emission kernel (hsr_launch_duplicate)
    if (R <= 0) return HSR_OK;
    const int passes = hsr_sort_tile_passes(end_bit);
    const int tile_bits = end_bit - 32;
    const int nblocks = (R + SORT_TILE - 1) / SORT_TILE;
    uint32_t* totals = b.hist + (size_t)256 * nblocks;
    uint64_t* ka = (passes & 1) ? b.keys_unsorted : b.keys;
    uint32_t* va = (passes & 1) ? b.vals_unsorted : b.vals;
    uint64_t* kb = (passes & 1) ? b.keys : b.keys_unsorted;
    uint32_t* vb = (passes & 1) ? b.vals : b.vals_unsorted;
    const int bits_per_pass = passes ? (tile_bits + passes - 1) / passes : 0;
    for (int p = 0; p < passes; p++) {
        const int shift = 32 + bits_per_pass * p;
        const int bits = min(bits_per_pass, end_bit - shift);
        const uint32_t mask = (1u << bits) - 1u;
        sort_hist_kernel<<<nblocks, SORT_THREADS, 0, stream>>>(ka, R, shift, mask, nblocks, b.hist);
        sort_rowscan_kernel<<<256, 64, 0, stream>>>(b.hist, nblocks, totals);
        sort_scatter_kernel<<<nblocks, SORT_THREADS, 0, stream>>>(ka, va, kb, vb, R, shift, mask, nblocks, b.hist, totals);
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    // (b.keys, b.vals) now hold the instances grouped by tile, in emission order inside each tile
    hsr_launch_tile_ranges_only(R, b.keys, ranges, stream);
    tile_sort_kernel<<<T, 256, 0, stream>>>(ranges, b.keys, b.vals, b.keys_unsorted, b.vals_unsorted, 0, 0, BinDevRef{nullptr, nullptr, 0});
    return HSR_OK;
}

// Per-tile sort alone, for segments in ARBITRARY order holding the 8-byte composites written by direct tile binning
// (hsr_launch_bin_tiles): tiles above TS_MAX entries radix-sort the Gaussian-index bytes before the depth bytes.
int hsr_launch_tile_sort(BinState& b, int T, int P, const uint2* ranges, hipStream_t stream, const BinDevRef* ref, int avg_per_tile_hint)
{
    int bits = 0;
    while (bits < 32 && (1ull << bits) < (unsigned long long)(P > 1 ? P : 1)) bits++;
    const BinDevRef r = ref ? *ref : BinDevRef{nullptr, nullptr, 0};
    static const bool block_only = getenv("HSR_SORT_IMPL") && !strcmp(getenv("HSR_SORT_IMPL"), "block");
    if (block_only) {
        tile_sort_kernel<<<T, 256, 0, stream>>>(ranges, b.keys, b.vals, b.keys_unsorted, b.vals_unsorted, (bits + 7) / 8, 1, r);
        return HSR_OK;
    }
    // tiles of <= 1024 entries: one wave each, elements in registers; the larger ones by whole workgroups — in the same launch
    // (four tiles per workgroup, one after the other) while they are the exception, in a launch of their own (one workgroup
    // per tile) when the previous frame averaged more than 800 entries per tile
    const bool many_big = avg_per_tile_hint > 800;
    // round 3: two waves per tile (tile_sort_pair_kernel); HSR_SORT_IMPL=wave keeps one wave per tile (parity-tested selector)
    static const bool one_wave = getenv("HSR_SORT_IMPL") && !strcmp(getenv("HSR_SORT_IMPL"), "wave");
    if (!one_wave)
        tile_sort_pair_kernel<<<(T + 1) / 2, 256, 0, stream>>>(T, ranges, b.keys, b.vals, b.keys_unsorted, b.vals_unsorted, (bits + 7) / 8,
                                                               many_big ? 0 : 1, r);
    else
    tile_sort_wave_kernel<<<(T + 3) / 4, 256, 0, stream>>>(T, ranges, b.keys, b.vals, b.keys_unsorted, b.vals_unsorted, (bits + 7) / 8,
                                                           many_big ? 0 : 1, r);
    if (many_big)
        tile_sort_kernel<<<T, 256, 0, stream>>>(ranges, b.keys, b.vals, b.keys_unsorted, b.vals_unsorted, (bits + 7) / 8, 2, r);
    return HSR_OK;
}
