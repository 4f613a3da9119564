// hsr_preprocess.hip — per-Gaussian forward stages for gfx950:
//   mark_visible, preprocess (cull / project / covariance / radius / tile rect / SH), the tile-count
//   scan, (tile|depth) key emission, and tile-range identification.
//
// Compiled with -ffp-contract=off: every value that decides an INTEGER output (radius, tile rect,
// tiles_touched, depth key bits) is evaluated in fp32 IEEE order with no FMA contraction, in the
// operation order of the reference (forward.cu:74-256, auxiliary.h:41-164; GLM mat3 products sum
// left to right, glm/detail/type_mat3x3.inl:486-520), so these outputs are bit-identical to the CPU
// oracle.  Division and sqrt are correctly rounded (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).
#include "hsr_common.h"

namespace {

__device__ __constant__ float SH_C0 = 0.28209479177387814f;
__device__ __constant__ float SH_C1 = 0.4886025119029199f;
__device__ __constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

// column-major 3x3 like glm::mat3: c[col][row]
struct M3 {
    float c[3][3];
};
__device__ __forceinline__ M3 m3mul(const M3& a, const M3& b)
{
    M3 r;
#pragma unroll
    for (int cc = 0; cc < 3; cc++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
            r.c[cc][rr] = a.c[0][rr] * b.c[cc][0] + a.c[1][rr] * b.c[cc][1] + a.c[2][rr] * b.c[cc][2];
    return r;
}
__device__ __forceinline__ M3 m3t(const M3& a)
{
    M3 r;
#pragma unroll
    for (int cc = 0; cc < 3; cc++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) r.c[cc][rr] = a.c[rr][cc];
    return r;
}

// tile rect of a splat (reference getRect, auxiliary.h:46-56): float divide by 16 then truncation
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, uint32_t& x0, uint32_t& y0,
                                          uint32_t& x1, uint32_t& y1)
{
    const float r = (float)radius;
    int a;
    a = (int)((px - r) / 16.0f); a = a < 0 ? 0 : a; x0 = a < gx ? a : gx;
    a = (int)((py - r) / 16.0f); a = a < 0 ? 0 : a; y0 = a < gy ? a : gy;
    // (p + r + BLOCK - 1) evaluates left to right in fp32: ((p + r) + 16) - 1, not p + r + 15
    a = (int)((((px + r) + 16.0f) - 1.0f) / 16.0f); a = a < 0 ? 0 : a; x1 = a < gx ? a : gx;
    a = (int)((((py + r) + 16.0f) - 1.0f) / 16.0f); a = a < 0 ? 0 : a; y1 = a < gy ? a : gy;
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ pts,
                                                           const float* __restrict__ view, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float x = pts[3 * idx], y = pts[3 * idx + 1], z = pts[3 * idx + 2];
    const float vz = view[2] * x + view[6] * y + view[10] * z + view[14];
    present[idx] = !(vz <= 0.2f);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// SH -> RGB (reference computeColorFromSH, forward.cu:20-71)
__device__ void sh_to_rgb(int idx, int deg, int max_coeffs, float px, float py, float pz, const float* campos,
                          const float* __restrict__ shs, uint8_t* __restrict__ clamped, float* __restrict__ rgb)
{
    float dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define SH(i) sh[(i) * 3 + c]
        float r = SH_C0 * SH(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                    SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        r += 0.5f;
        clamped[3 * idx + c] = (r < 0);
        rgb[3 * idx + c] = r > 0.0f ? r : 0.0f;
    }
}

// One Gaussian per lane; also leaves the per-block sum of tiles_touched in block_sums[blockIdx.x]
// so that the scan needs no extra pass over P.
__global__ void __launch_bounds__(256) preprocess_kernel(PreprocessArgs a, GeomState g, int* __restrict__ radii)
{
    __shared__ uint32_t wsum[4];
    __shared__ float4 s_rec[3][256];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    uint32_t touched = 0;
    float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0, rec2 = rec0;
    if (idx < a.P) {
        int my_radius_i = 0;
        do {
            const float px = a.means3D[3 * idx], py = a.means3D[3 * idx + 1], pz = a.means3D[3 * idx + 2];
            const float* vm = a.viewmatrix;
            const float* pm = a.projmatrix;
            // in_frustum (auxiliary.h:139-164): only the view-space depth decides
            const float tvx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
            const float tvy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
            const float tvz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
            if (tvz <= 0.2f) {
                // The reference traps the device here (auxiliary.h:156-160).  A trap is a queue exception on ROCm — it takes
                // the process (and the host's wait for num_rendered) down with it — so the violation is flagged instead: the
                // point is culled like any other, counters[1] tells the host, and the call returns an error.
                if (a.prefiltered) g.counters[1] = 1u;
                break;
            }
            const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
            const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
            const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
            const float p_w = 1.0f / (hw + 0.0000001f);
            const float projx = hx * p_w, projy = hy * p_w;

            float cov3D[6];
            if (a.cov3D_precomp) {
#pragma unroll
                for (int i = 0; i < 6; i++) cov3D[i] = a.cov3D_precomp[(size_t)idx * 6 + i];
            } else {
                // computeCov3D (forward.cu:118-152)
                const float mod = a.scale_modifier;
                M3 S = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
                S.c[0][0] = mod * a.scales[3 * idx];
                S.c[1][1] = mod * a.scales[3 * idx + 1];
                S.c[2][2] = mod * a.scales[3 * idx + 2];
                const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
                const float r = q.x, x = q.y, y = q.z, z = q.w;
                M3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                         {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                         {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
                const M3 Mm = m3mul(S, R);
                const M3 Sg = m3mul(m3t(Mm), Mm);
                cov3D[0] = Sg.c[0][0]; cov3D[1] = Sg.c[0][1]; cov3D[2] = Sg.c[0][2];
                cov3D[3] = Sg.c[1][1]; cov3D[4] = Sg.c[1][2]; cov3D[5] = Sg.c[2][2];
#pragma unroll
                for (int i = 0; i < 6; i++) g.cov3D[(size_t)idx * 6 + i] = cov3D[i];
            }
            // computeCov2D (forward.cu:74-113)
            const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
            const float txtz = tvx / tvz, tytz = tvy / tvz;
            const float tx = fminf(limx, fmaxf(-limx, txtz)) * tvz;
            const float ty = fminf(limy, fmaxf(-limy, tytz)) * tvz;
            const M3 J = {{{a.focal_x / tvz, 0.0f, -(a.focal_x * tx) / (tvz * tvz)},
                           {0.0f, a.focal_y / tvz, -(a.focal_y * ty) / (tvz * tvz)},
                           {0, 0, 0}}};
            const M3 Wm = {{{vm[0], vm[4], vm[8]}, {vm[1], vm[5], vm[9]}, {vm[2], vm[6], vm[10]}}};
            const M3 T = m3mul(Wm, J);
            const M3 Vrk = {{{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}}};
            const M3 cv = m3mul(m3mul(m3t(T), m3t(Vrk)), T);
            const float cx = cv.c[0][0] + 0.3f, cy = cv.c[0][1], cz = cv.c[1][1] + 0.3f;
            const float det = (cx * cz - cy * cy);
            if (det == 0.0f) break;
            const float det_inv = 1.f / det;
            const float conx = cz * det_inv, cony = -cy * det_inv, conz = cx * det_inv;
            const float mid = 0.5f * (cx + cz);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            // ndc2Pix in double (auxiliary.h:41-44)
            const float pix = (float)((((double)projx + 1.0) * (double)a.W - 1.0) * 0.5);
            const float piy = (float)((((double)projy + 1.0) * (double)a.H - 1.0) * 0.5);
            uint32_t x0, y0, x1, y1;
            tile_rect(pix, piy, (int)my_radius, a.tiles_x, a.tiles_y, x0, y0, x1, y1);
            if ((x1 - x0) * (y1 - y0) == 0) break;
            if (a.colors_precomp == nullptr) sh_to_rgb(idx, a.D, a.M, px, py, pz, a.cam_pos, a.shs, g.clamped, g.rgb);
            g.depths[idx] = tvz;
            my_radius_i = (int)my_radius;
            g.means2D[idx] = make_float2(pix, piy);
            const float op = a.opacities[idx];
            g.conic_opacity[idx] = make_float4(conx, cony, conz, op);
            {
                // the record the tile kernels gather per (tile, Gaussian): one 64-byte line instead of a line in each of
                // means2D / conic_opacity / depths / colours; staged in LDS and written as one contiguous stream below
                const float* col = a.colors_precomp ? a.colors_precomp + 3 * (size_t)idx : g.rgb + 3 * (size_t)idx;
                rec0 = make_float4(pix, piy, tvz, 0.f);
                rec1 = make_float4(conx, cony, conz, op);
                rec2 = make_float4(col[0], col[1], col[2], 0.f);
            }
            touched = (y1 - y0) * (x1 - x0);
        } while (0);
        radii[idx] = my_radius_i;
        g.tiles_touched[idx] = touched;
    }
    const uint32_t ws = wave_sum_u32(touched);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ws;
    s_rec[0][threadIdx.x] = rec0;
    s_rec[1][threadIdx.x] = rec1;
    s_rec[2][threadIdx.x] = rec2;
    __syncthreads();
    if (threadIdx.x == 0) g.block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    // the block's 256 records (4 x 16 bytes each, the fourth unused) as one contiguous, fully coalesced stream
    float4* out = g.rec + (size_t)blockIdx.x * 1024;
    const int nrec = min(256, a.P - blockIdx.x * 256);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int f = threadIdx.x + 256 * k, gi = f >> 2, c = f & 3;
        if (gi < nrec) out[f] = c < 3 ? s_rec[c][gi] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// block-wide exclusive scan of one value per thread (1024 threads = 16 waves)
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* smem /*[NT/64 + 1]*/, uint32_t& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    if (w == 0) {
        uint32_t s = lane < NT / 64 ? smem[lane] : 0;
        uint32_t si = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(si, o);
            if (lane >= o) si += t;
        }
        if (lane < NT / 64) smem[lane] = si - s;  // exclusive wave offsets
        if (lane == 63) smem[NT / 64] = si;       // total (lanes >= NT/64 carry the full sum)
    }
    __syncthreads();
    total = smem[NT / 64];
    const uint32_t r = smem[w] + inc - v;
    __syncthreads();
    return r;
}

// exclusive scan of the per-block sums in place (single block); total -> counters[0]
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(int nblk, uint32_t* __restrict__ block_sums,
                                                               uint32_t* __restrict__ counters)
{
    __shared__ uint32_t smem[1024 / 64 + 1];
    const int per = (nblk + 1023) / 1024;
    const int beg = threadIdx.x * per;
    uint32_t local = 0;
    for (int i = 0; i < per; i++) {
        const int j = beg + i;
        if (j < nblk) local += block_sums[j];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan<1024>(local, smem, total);
    for (int i = 0; i < per; i++) {
        const int j = beg + i;
        if (j < nblk) {
            const uint32_t v = block_sums[j];
            block_sums[j] = run;
            run += v;
        }
    }
    if (threadIdx.x == 0) counters[0] = total;
}

// Finishes the scan (point_offsets = inclusive sum, as cub::DeviceScan::InclusiveSum at
// rasterizer_impl.cu:281) and emits one (tile|depth, id) pair per touched tile, row-major over the
// rect exactly like the reference's duplicateWithKeys (rasterizer_impl.cu:70-111).
__global__ void __launch_bounds__(256) duplicate_kernel(int P, const int* __restrict__ radii, int tiles_x, int tiles_y,
                                                        GeomState g, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        uint2* __restrict__ ranges)
{
    __shared__ uint32_t smem[256 / 64 + 1];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    // the reference's cudaMemset of the tile ranges (rasterizer_impl.cu:314) rides along here: ranges are next
    // written by tile_ranges_kernel, several launches later
    for (int i = idx; i < tiles_x * tiles_y; i += gridDim.x * 256) ranges[i] = make_uint2(0u, 0u);
    const uint32_t touched = idx < P ? g.tiles_touched[idx] : 0;
    uint32_t total;
    uint32_t off = block_exclusive_scan<256>(touched, smem, total) + g.block_sums[blockIdx.x];
    if (idx >= P) return;
    g.point_offsets[idx] = off + touched;
    const int radius = radii[idx];
    if (radius > 0) {
        const float2 xy = g.means2D[idx];
        uint32_t x0, y0, x1, y1;
        tile_rect(xy.x, xy.y, radius, tiles_x, tiles_y, x0, y0, x1, y1);
        const uint64_t dbits = (uint64_t)__float_as_uint(g.depths[idx]);
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) {
                keys[off] = ((uint64_t)(y * (uint32_t)tiles_x + x) << 32) | dbits;
                vals[off] = (uint32_t)idx;
                off++;
            }
    }
}

// reference identifyTileRanges (rasterizer_impl.cu:116-138)
__global__ void __launch_bounds__(256) tile_ranges_kernel(int L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L) return;
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    if (idx == 0)
        ranges[cur].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (cur != prev) {
            ranges[prev].y = idx;
            ranges[cur].x = idx;
        }
    }
    if (idx == L - 1) ranges[cur].y = L;
}

// ---------------------------------------------------------------------------------------------------------------
// Direct tile binning (T <= BIN_MAX_TILES): instead of emitting the instances in Gaussian order and radix-sorting them
// on the tile bits (2 passes of histogram / scan / scatter over all R pairs at 1200x680), count per tile first and emit
// every instance straight into its tile's segment:
//   bin_hist   : workgroup b owns a contiguous range of Gaussians, counts their tiles in LDS (ds_add), writes row b of
//                table[nblk][T]; also finishes the per-Gaussian inclusive scan (point_offsets), which no longer feeds
//                the emission but is part of the state the reference keeps (rasterizer_impl.cu:281);
//   bin_scan   : thread = tile: exclusive prefix over the nblk rows in place, tile totals;
//   bin_emit   : same ownership as bin_hist; every workgroup scans the tile totals itself (= tile segment starts = the reference's
//                tile ranges, written by workgroup 0); LDS cursors (segment start + row prefix), ds_add_rtn gives each instance
//                its slot (it stores the 8-byte sort composite (depth bits, index) there; the tile is implied by the
//                segment).  The order INSIDE a tile segment is whatever the LDS atomics produce — irrelevant, because
//                tile_sort_kernel then orders each segment by the unique composite (depth bits, Gaussian index), which
//                is exactly the order a stable sort of the emission order on (tile, depth) gives.
// Bit-identical sorted keys / values / ranges, 3 small launches instead of 8, and the 12 B x R unsorted copy is gone.
constexpr int BIN_MAX_TILES = 8192;   // LDS: one u32 counter per tile

constexpr int BIN_THREADS = 1024;     // 16 waves per workgroup: each workgroup touches all T counters twice (zero / read
                                      // out), so few large workgroups with many waves beat many small ones

__global__ void __launch_bounds__(BIN_THREADS) bin_hist_kernel(int P, int per_block, const int* __restrict__ radii, int tiles_x,
                                                               int tiles_y, GeomState g, uint32_t* __restrict__ table,
                                                               uint32_t* host_counter, uint32_t host_seq)
{
    extern __shared__ uint32_t s_cnt[];   // [T]
    __shared__ uint32_t s_wsum[BIN_THREADS / 64];
    __shared__ uint32_t s_red[BIN_THREADS / 64];
    const int T = tiles_x * tiles_y;
    for (int i = threadIdx.x; i < T; i += BIN_THREADS) s_cnt[i] = 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g0 = blockIdx.x * per_block;   // per_block is a multiple of 1024: chunks coincide with 4 preprocess blocks
    // first chunk's Gaussians requested before the prefix over the block sums, every later chunk one iteration ahead (see bin_emit_kernel)
    uint32_t n_touched; int n_radius; float2 n_xy;
    {
        const int i0 = min(g0 + (int)threadIdx.x, P - 1);
        n_touched = g.tiles_touched[i0]; n_radius = radii[i0]; n_xy = g.means2D[i0];
    }
    // The scan over the preprocess blocks' tile-count sums rides along here (no separate single-workgroup launch): this
    // workgroup sums the RAW per-block sums in front of its first Gaussian itself — at most P/256 L2-resident words — and the
    // last workgroup, whose running sum ends at the total, publishes num_rendered.
    const int nblk_pre = (P + 255) >> 8, pre0 = g0 >> 8;
    uint32_t part = 0u;
    for (int j = threadIdx.x; j < pre0 && j < nblk_pre; j += BIN_THREADS) part += g.block_sums[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) s_red[w] = part;
    __syncthreads();
    uint32_t chunk_base = 0u;   // sum of the tile counts of every Gaussian in front of the current chunk
#pragma unroll
    for (int k = 0; k < BIN_THREADS / 64; k++) chunk_base += s_red[k];
    for (int c = 0; c < per_block && g0 + c < P; c += BIN_THREADS) {
        const int idx = g0 + c + threadIdx.x;
        const uint32_t touched = idx < P ? n_touched : 0u;
        const int radius_cur = n_radius;
        const float2 xy_cur = n_xy;
        {
            const int in = min(idx + BIN_THREADS, P - 1);
            n_touched = g.tiles_touched[in]; n_radius = radii[in]; n_xy = g.means2D[in];
        }
        // inclusive scan inside the wave, then the sums of the earlier waves of the same 256-Gaussian preprocess block
        uint32_t inc = touched;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        __syncthreads();
        if (lane == 63) s_wsum[w] = inc;
        __syncthreads();
        // the four preprocess blocks of this chunk: their raw sums give the offsets of blocks 1..3 and the next chunk's base
        const int pb0 = (g0 + c) >> 8;
        uint32_t bs[BIN_THREADS / 256];
#pragma unroll
        for (int q = 0; q < BIN_THREADS / 256; q++) bs[q] = pb0 + q < nblk_pre ? g.block_sums[pb0 + q] : 0u;
        uint32_t off = chunk_base;
#pragma unroll
        for (int q = 0; q < BIN_THREADS / 256; q++) {
            off += q < (w >> 2) ? bs[q] : 0u;
            chunk_base += bs[q];
        }
        for (int k = w & ~3; k < w; k++) off += s_wsum[k];
        if (idx < P) {
            g.point_offsets[idx] = off + inc;
            const int radius = radius_cur;
            if (radius > 0) {
                const float2 xy = xy_cur;
                uint32_t x0, y0, x1, y1;
                tile_rect(xy.x, xy.y, radius, tiles_x, tiles_y, x0, y0, x1, y1);
                for (uint32_t y = y0; y < y1; y++)
                    for (uint32_t x = x0; x < x1; x++) atomicAdd(&s_cnt[y * (uint32_t)tiles_x + x], 1u);
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        g.counters[0] = chunk_base;   // num_rendered
        if (host_counter) {   // straight into host-mapped memory, value first, then the call's sequence number: the host polls it
            __hip_atomic_store(&host_counter[0], chunk_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_counter[2], g.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // prefilter flag
            __hip_atomic_store(&host_counter[1], host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * T;
    for (int i = threadIdx.x; i < T; i += BIN_THREADS) row[i] = s_cnt[i];
}
constexpr int BIN_SCAN_TILES = 16;   // tiles per workgroup of bin_scan_kernel: 64 row groups each -> T/16 workgroups fill the chip
__global__ void __launch_bounds__(1024) bin_scan_kernel(int T, int nblk, uint32_t* __restrict__ table, uint32_t* __restrict__ totals)
{
    constexpr int GROUPS = 1024 / BIN_SCAN_TILES;
    __shared__ uint32_t s_sum[GROUPS][BIN_SCAN_TILES + 1];
    const int tl = threadIdx.x % BIN_SCAN_TILES, grp = threadIdx.x / BIN_SCAN_TILES;
    const int i = blockIdx.x * BIN_SCAN_TILES + tl;
    const int rows = (nblk + GROUPS - 1) / GROUPS;
    const int b0 = grp * rows, b1 = min(nblk, b0 + rows);
    uint32_t sum = 0;
    if (i < T)
        for (int b = b0; b < b1; b++) sum += table[(size_t)b * T + i];
    s_sum[grp][tl] = sum;
    __syncthreads();
    uint32_t run = 0, tot = 0;
#pragma unroll 8
    for (int k = 0; k < GROUPS; k++) {
        const uint32_t v = s_sum[k][tl];
        run += k < grp ? v : 0u;
        tot += v;
    }
    if (i >= T) return;
    if (grp == 0) totals[i] = tot;
    for (int b = b0; b < b1; b++) {
        const uint32_t c = table[(size_t)b * T + i];
        table[(size_t)b * T + i] = run;
        run += c;
    }
}
__global__ void __launch_bounds__(BIN_THREADS) bin_emit_kernel(int P, int per_block, const int* __restrict__ radii, int tiles_x, int tiles_y,
                                                       GeomState g, const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals,
                                                       uint2* __restrict__ ranges, uint64_t* __restrict__ comp, BinDevRef ref)
{
    if (ref.base) {   // speculative forward: the array lives where num_rendered says
        BinState bs;
        if (!hsr_bin_resolve(ref, *ref.R_dev, &bs)) return;
        comp = bs.keys;
    }
    extern __shared__ uint32_t s_cur[];   // [T]
    __shared__ uint32_t s_scan[BIN_THREADS / 64 + 1];
    const int T = tiles_x * tiles_y;
    const uint32_t* row = table + (size_t)blockIdx.x * T;
    // The Gaussians of this workgroup's first chunk are requested NOW, before the scan of the tile totals below (three workgroup
    // barriers and two rounds of L2 loads that these loads do not depend on), and every later chunk one iteration ahead: the kernel is
    // a chain of dependent round trips (67 % of its wave cycles were waits), not bandwidth.  Unconditional, clamped loads.
    const int g0 = blockIdx.x * per_block;
    const int g1 = min(P, g0 + per_block);
    int n_radius; float2 n_xy; float n_depth;
    {
        const int i0 = min(g0 + (int)threadIdx.x, P - 1);
        n_radius = radii[i0]; n_xy = g.means2D[i0]; n_depth = g.depths[i0];
    }
    // segment starts = exclusive scan of the tile totals: every workgroup scans the (L2-resident) totals itself — T <= 8192 words,
    // at most 8 per thread — instead of reading them from a single-workgroup launch of their own; workgroup 0 also writes the
    // reference's tile ranges (identifyTileRanges, rasterizer_impl.cu:116-138; untouched tiles keep {0,0} as after its memset, :314)
    {
        const int per = (T + BIN_THREADS - 1) / BIN_THREADS;
        const int beg = threadIdx.x * per;
        uint32_t cnt[BIN_MAX_TILES / BIN_THREADS];
        uint32_t local = 0;
#pragma unroll
        for (int i = 0; i < BIN_MAX_TILES / BIN_THREADS; i++) {
            cnt[i] = (i < per && beg + i < T) ? totals[beg + i] : 0u;
            local += cnt[i];
        }
        uint32_t total;
        uint32_t run = block_exclusive_scan<BIN_THREADS>(local, s_scan, total);
#pragma unroll
        for (int i = 0; i < BIN_MAX_TILES / BIN_THREADS; i++) {
            const int j = beg + i;
            if (i < per && j < T) {
                s_cur[j] = run + row[j];
                if (blockIdx.x == 0) ranges[j] = cnt[i] ? make_uint2(run, run + cnt[i]) : make_uint2(0u, 0u);
                run += cnt[i];
            }
        }
    }
    __syncthreads();
    for (int base = g0; base < g1; base += BIN_THREADS) {
        const int idx = base + threadIdx.x;
        const int radius = n_radius;
        const float2 xy = n_xy;
        const float depth = n_depth;
        {
            const int in = min(idx + BIN_THREADS, P - 1);
            n_radius = radii[in]; n_xy = g.means2D[in]; n_depth = g.depths[in];
        }
        if (idx >= g1 || radius <= 0) continue;
        uint32_t x0, y0, x1, y1;
        tile_rect(xy.x, xy.y, radius, tiles_x, tiles_y, x0, y0, x1, y1);
        // the tile is implied by the segment: one 8-byte store of the sort composite (depth bits, index) per instance;
        // tile_sort_kernel turns it into the reference's (tile | depth) key and the index value
        const uint64_t c = ((uint64_t)__float_as_uint(depth) << 32) | (uint64_t)(uint32_t)idx;
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) comp[atomicAdd(&s_cur[y * (uint32_t)tiles_x + x], 1u)] = c;
    }
}

}  // namespace

int hsr_launch_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present,
                            hipStream_t stream)
{
    (void)proj;
    if (P <= 0) return HSR_OK;
    mark_visible_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, means3D, view, present);
    return HSR_OK;
}

int hsr_launch_preprocess(const PreprocessArgs& a, GeomState& g, hipStream_t stream)
{
    preprocess_kernel<<<(a.P + 255) / 256, 256, 0, stream>>>(a, g, a.radii);
    return HSR_OK;
}

int hsr_launch_scan_block_sums(int P, GeomState& g, hipStream_t stream)
{
    scan_block_sums_kernel<<<1, 1024, 0, stream>>>((P + 255) / 256, g.block_sums, g.counters);
    return HSR_OK;
}

int hsr_launch_duplicate(int P, const int* radii, int tiles_x, int tiles_y, GeomState& g, BinState& b, uint2* ranges,
                         hipStream_t stream)
{
    duplicate_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, radii, tiles_x, tiles_y, g, b.keys_unsorted, b.vals_unsorted, ranges);
    return HSR_OK;
}

// Direct tile binning (see bin_hist_kernel), in two halves so that the host's wait for num_rendered (it sizes the binning
// buffer, reference rasterizer_impl.cu:285) overlaps the count phase instead of idling the GPU:
//   hsr_launch_bin_count : needs neither num_rendered nor the binning buffer — count table in `scratch` (the caller lends
//                          the image state's final_T array, which nothing touches before the render kernel), fills ranges;
//   hsr_launch_bin_emit  : after the binning buffer exists, emits into its `keys` array.
// hsr_bin_plan() decides the partition (or declines: too many tiles for LDS counters / scratch too small).
bool hsr_bin_plan(int P, int T, size_t scratch_words, HsrBinPlan* plan)
{
    if (T > BIN_MAX_TILES || T <= 0 || P <= 0) return false;
    if (scratch_words < (size_t)T * (2 + 8)) return false;
    int nblk = (P + 2047) / 2048;
    if (P <= (1 << 18)) nblk = (P + 1023) / 1024;   // small maps: 2 048 per workgroup would leave most CUs without one (100k: 49 workgroups); A/B at 100k: 0.0262 -> 0.0240 ms, none from 300k up
    if (nblk > 512) nblk = 512;   // every workgroup zeroes, writes and re-reads a T-word row: beyond two rounds of the chip, larger chunks win
    const size_t max_blk = (scratch_words - 2 * (size_t)T) / (size_t)T;
    if ((size_t)nblk > max_blk) nblk = (int)max_blk;
    if (nblk > 2048) nblk = 2048;
    plan->per_block = ((P + nblk - 1) / nblk + BIN_THREADS - 1) & ~(BIN_THREADS - 1);   // multiple of 1024
    plan->nblk = (P + plan->per_block - 1) / plan->per_block;
    return true;
}

int hsr_launch_bin_count(const HsrBinPlan& plan, int P, const int* radii, int tiles_x, int tiles_y, GeomState& g, uint32_t* scratch,
                         uint2* ranges, hipStream_t stream, uint32_t* host_counter, uint32_t host_seq)
{
    const int T = tiles_x * tiles_y;
    uint32_t* table = scratch;
    uint32_t* totals = table + (size_t)plan.nblk * T;
    (void)ranges;   // written by bin_emit_kernel's workgroup 0, from the same totals
    bin_hist_kernel<<<plan.nblk, BIN_THREADS, (size_t)T * sizeof(uint32_t), stream>>>(P, plan.per_block, radii, tiles_x, tiles_y, g, table,
                                                                                     host_counter, host_seq);
    bin_scan_kernel<<<(T + BIN_SCAN_TILES - 1) / BIN_SCAN_TILES, 1024, 0, stream>>>(T, plan.nblk, table, totals);
    return HSR_OK;
}

int hsr_launch_bin_emit(const HsrBinPlan& plan, int P, const int* radii, int tiles_x, int tiles_y, GeomState& g, const uint32_t* scratch,
                        uint2* ranges, uint64_t* comp, hipStream_t stream, const BinDevRef* ref)
{
    const int T = tiles_x * tiles_y;
    const uint32_t* table = scratch;
    const uint32_t* totals = table + (size_t)plan.nblk * T;
    bin_emit_kernel<<<plan.nblk, BIN_THREADS, (size_t)T * sizeof(uint32_t), stream>>>(P, plan.per_block, radii, tiles_x, tiles_y, g, table,
                                                                                      totals, ranges, comp,
                                                                                      ref ? *ref : BinDevRef{nullptr, nullptr, 0});
    return HSR_OK;
}

int hsr_launch_tile_ranges_only(int R, const uint64_t* keys, uint2* ranges, hipStream_t stream)
{
    if (R > 0) tile_ranges_kernel<<<(R + 255) / 256, 256, 0, stream>>>(R, keys, ranges);
    return HSR_OK;
}

int hsr_launch_tile_ranges(int R, int T, const uint64_t* keys, uint2* ranges, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)T, stream);
    if (e != hipSuccess) {
        hsr_set_error("hipMemsetAsync(ranges) failed: %s", hipGetErrorString(e));
        return HSR_ERR_HIP;
    }
    if (R > 0) tile_ranges_kernel<<<(R + 255) / 256, 256, 0, stream>>>(R, keys, ranges);
    return HSR_OK;
}
