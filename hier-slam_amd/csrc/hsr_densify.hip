// hsr_densify.hip — silhouette densification of one mapped frame on device (gfx950), SURVEY.md §8(f) rank 3.
// See include/hsr_densify.h for the reference lines (scripts/hierslam.py:1264-1305, :144-194, :1157).
//   median : 4 x (select_hist_kernel -> select_pick_kernel): 8-bit radix select, most significant byte first, of the rank
//            (N-1)/2 element of depth_error's bit patterns (torch.median = lower median);
//   mask   : mask_count_kernel — the non-presence mask + one count per 256-pixel block;
//   scan   : scan_counts_kernel — one workgroup, exclusive scan of the block counts, total -> out_count;
//   emit   : emit_points_kernel — order-preserving compaction (block offset + wave ballot rank), back-projection, colours,
//            log-scales.  Compiled with -ffp-contract=off: the emitted means are what the rasterizer will bin next.
#include "hsr_common.h"
#include "../../include/hsr_densify.h"

namespace {

constexpr int DB = 256;

__device__ __forceinline__ float depth_error(float gt, float rd) { return fabsf(gt - rd) * (gt > 0.f ? 1.f : 0.f); }

struct SelectState { unsigned prefix, mask, rank; };   // device scratch: bits fixed so far, their mask, remaining rank

__global__ __launch_bounds__(DB) void select_init_kernel(SelectState* st, int N, unsigned* hist)
{
    if (threadIdx.x == 0) { st->prefix = 0u; st->mask = 0u; st->rank = (unsigned)((N - 1) / 2); }
    hist[threadIdx.x] = 0u;
}

// histogram of byte `shift` among the elements whose already-fixed bits equal the prefix
__global__ __launch_bounds__(DB) void select_hist_kernel(const float* __restrict__ gt, const float* __restrict__ rd, int N, int shift,
                                                         const SelectState* __restrict__ st, unsigned* __restrict__ hist)
{
    __shared__ unsigned s_h[256];
    s_h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned prefix = st->prefix, mask = st->mask;
    for (int i = blockIdx.x * DB * 8 + threadIdx.x, it = 0; it < 8; it++, i += DB) {
        if (i >= N) break;
        const unsigned bits = __float_as_uint(depth_error(gt[i], rd[i]));
        if ((bits & mask) == prefix) atomicAdd(&s_h[(bits >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);   // integer atomics: order-independent
}

__global__ __launch_bounds__(DB) void select_pick_kernel(SelectState* st, unsigned* hist, int shift, float* out_median)
{
    __shared__ unsigned s_h[256];
    s_h[threadIdx.x] = hist[threadIdx.x];
    hist[threadIdx.x] = 0u;   // ready for the next pass
    __syncthreads();
    if (threadIdx.x != 0) return;
    unsigned rank = st->rank, d = 0;
    for (; d < 255u; d++) {
        if (rank < s_h[d]) break;
        rank -= s_h[d];
    }
    st->prefix |= d << shift;
    st->mask |= 255u << shift;
    st->rank = rank;
    if (shift == 0 && out_median) *out_median = __uint_as_float(st->prefix);
}

__device__ __forceinline__ bool non_presence(float sil, float rd, float gt, float sil_thres, float thr)
{
    const float derr = depth_error(gt, rd);
    const bool by_depth = (rd > gt) && (derr > thr);
    return ((sil < sil_thres) || by_depth) && (gt > 0.f);
}

__global__ __launch_bounds__(DB) void mask_count_kernel(const float* __restrict__ sil, const float* __restrict__ rd,
                                                        const float* __restrict__ gt, int N, float sil_thres, float depth_factor,
                                                        const SelectState* __restrict__ st, unsigned* __restrict__ counts,
                                                        uint8_t* __restrict__ out_mask)
{
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * DB + threadIdx.x;
    const float thr = depth_factor * __uint_as_float(st->prefix);   // 50 * depth_error.median()
    const bool m = i < N && non_presence(sil[i], rd[i], gt[i], sil_thres, thr);
    if (out_mask && i < N) out_mask[i] = m ? 1 : 0;
    const unsigned long long b = __ballot(m);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(1024) void scan_counts_kernel(int nblk, unsigned* __restrict__ counts, int* __restrict__ out_count)
{
    __shared__ unsigned s_w[17];
    const int per = (nblk + 1023) / 1024, beg = threadIdx.x * per;
    unsigned local = 0;
    for (int k = 0; k < per; k++)
        if (beg + k < nblk) local += counts[beg + k];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int k = 0; k < 16; k++) { const unsigned v = s_w[k]; s_w[k] = run; run += v; }
        s_w[16] = run;
    }
    __syncthreads();
    unsigned run = s_w[w] + inc - local;
    for (int k = 0; k < per; k++)
        if (beg + k < nblk) { const unsigned v = counts[beg + k]; counts[beg + k] = run; run += v; }
    if (threadIdx.x == 0) *out_count = (int)s_w[16];
}

struct Frame { float fx, fy, cx, cy; };

__global__ __launch_bounds__(DB) void emit_points_kernel(const float* __restrict__ sil, const float* __restrict__ rd,
                                                         const float* __restrict__ gt, const float* __restrict__ color, int W, int N,
                                                         Frame f, const float* __restrict__ c2w, float sil_thres, float depth_factor,
                                                         const SelectState* __restrict__ st, const unsigned* __restrict__ offsets,
                                                         int capacity, float* __restrict__ out_means, float* __restrict__ out_rgb,
                                                         float* __restrict__ out_log_scales, float* __restrict__ out_msd)
{
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * DB + threadIdx.x;
    const float thr = depth_factor * __uint_as_float(st->prefix);
    const float z = i < N ? gt[i] : 0.f;
    const bool m = i < N && non_presence(sil[i], rd[i], z, sil_thres, thr);
    const unsigned long long b = __ballot(m);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = (unsigned)__popcll(b);
    __syncthreads();
    if (!m) return;
    unsigned pos = offsets[blockIdx.x] + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
    for (int k = 0; k < w; k++) pos += s_w[k];
    if (pos >= (unsigned)capacity) return;
    // get_pointcloud (scripts/hierslam.py:153-168): xx = (x - CX)/FX, pts_cam = (xx*z, yy*z, z), pts = (c2w @ [pts_cam, 1])[:3]
    const int py = i / W, px = i - py * W;
    const float xx = ((float)px - f.cx) / f.fx, yy = ((float)py - f.cy) / f.fy;
    const float pc0 = xx * z, pc1 = yy * z, pc2 = z;
#pragma unroll
    for (int r = 0; r < 3; r++)
        out_means[3 * pos + r] = ((c2w[4 * r] * pc0 + c2w[4 * r + 1] * pc1) + c2w[4 * r + 2] * pc2) + c2w[4 * r + 3] * 1.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) out_rgb[3 * pos + c] = color[(size_t)c * N + i];
    const float sg = z / ((f.fx + f.fy) / 2.0f);   // :176-177
    const float msd = sg * sg;
    if (out_msd) out_msd[pos] = msd;
    out_log_scales[pos] = logf(sqrtf(msd));        // :1157
}

size_t dalign(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- prune + concat of the map (utils/slam_external.py:121-188) as ONE order-preserving row compaction --------------------
// keep mask of prune_gaussians (:175-180): to_remove = sigmoid(logit_opacity) < threshold  |  max_c exp(log_scale_c) > big
__global__ __launch_bounds__(DB) void prune_mask_kernel(int P, int S, const float* __restrict__ logit, const float* __restrict__ lscale,
                                                        float thr, float big, uint8_t* __restrict__ keep, unsigned* __restrict__ counts)
{
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * DB + threadIdx.x;
    bool k = false;
    if (i < P) {
        const float sg = 1.0f / (1.0f + expf(-logit[i]));          // torch.sigmoid
        bool rem = sg < thr;
        if (big > 0.f && lscale) {
            float m = expf(lscale[(size_t)i * S]);
            for (int c = 1; c < S; c++) m = fmaxf(m, expf(lscale[(size_t)i * S + c]));
            rem = rem || (m > big);
        }
        k = !rem;
        keep[i] = k ? 1 : 0;
    }
    const unsigned long long b = __ballot(k);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(DB) void count_keep_kernel(int P, const uint8_t* __restrict__ keep, unsigned* __restrict__ counts)
{
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * DB + threadIdx.x;
    const unsigned long long b = __ballot(i < P && keep[i] != 0);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void add_int_kernel(int* v, int base, int add, int overwrite) { *v = (overwrite ? base : *v) + add; }
struct RowTables {
    int n;
    hsr_row_table t[HSR_MAX_ROW_TABLES];
};
// rows [0, P): kept rows move to their rank (block offset + ballot rank: source order is preserved, which is what
// `tensor[to_keep]` yields); rows [P, P + n_append): the appended rows (or zeros: fresh Adam moments) land behind them
__global__ __launch_bounds__(DB) void compact_append_kernel(int P, int n_append, const uint8_t* __restrict__ keep,
                                                            const unsigned* __restrict__ offsets, const int* __restrict__ kept_total,
                                                            RowTables tb)
{
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * DB + threadIdx.x;
    const int nblk_keep = (P + DB - 1) / DB;
    if ((int)blockIdx.x < nblk_keep) {
        const bool k = i < P && (!keep || keep[i] != 0);
        const unsigned long long b = __ballot(k);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) s_w[w] = (unsigned)__popcll(b);
        __syncthreads();
        if (!k) return;
        size_t pos = (keep ? offsets[blockIdx.x] : (unsigned)(blockIdx.x * DB)) + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
        for (int q = 0; q < w; q++) pos += s_w[q];
        for (int t = 0; t < tb.n; t++) {
            const int C = tb.t[t].cols;
            const float* src = tb.t[t].src + (size_t)i * C;
            float* dst = tb.t[t].dst + pos * C;
            for (int c = 0; c < C; c++) dst[c] = src[c];
        }
        return;
    }
    const int a = i - nblk_keep * DB;
    if (a >= n_append) return;
    const size_t pos = (size_t)(keep ? *kept_total : P) + a;
    for (int t = 0; t < tb.n; t++) {
        const int C = tb.t[t].cols;
        float* dst = tb.t[t].dst + pos * C;
        const float* app = tb.t[t].append;
        for (int c = 0; c < C; c++) dst[c] = app ? app[(size_t)a * C + c] : 0.f;
    }
}

}  // namespace

extern "C" size_t hsr_compact_scratch_bytes(int P)
{
    const size_t nblk = ((size_t)(P > 0 ? P : 1) + DB - 1) / DB;
    return dalign(nblk * sizeof(unsigned)) + 512;
}

extern "C" int hsr_prune_mask(int P, int S, const float* logit_opacities, const float* log_scales, float removal_opacity_threshold,
                              float big_world_threshold, uint8_t* out_keep, int* out_kept, char* scratch, size_t scratch_bytes,
                              void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || !out_kept || (P > 0 && (!logit_opacities || !out_keep)) || (big_world_threshold > 0.f && (S < 1 || !log_scales))) {
        hsr_set_error("prune_mask: invalid arguments (P=%d S=%d)", P, S);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!scratch || scratch_bytes < hsr_compact_scratch_bytes(P)) {
        hsr_set_error("prune_mask: scratch too small: %zu bytes needed", hsr_compact_scratch_bytes(P));
        return HSR_ERR_BUFFER_TOO_SMALL;
    }
    unsigned* counts = reinterpret_cast<unsigned*>(scratch);
    const int nblk = (P + DB - 1) / DB;
    if (P > 0)
        prune_mask_kernel<<<nblk, DB, 0, stream>>>(P, S, logit_opacities, log_scales, removal_opacity_threshold, big_world_threshold,
                                                   out_keep, counts);
    scan_counts_kernel<<<1, 1024, 0, stream>>>(nblk, counts, out_kept);   // counts -> exclusive offsets (what compact needs), total
    HSR_HIP_CHECK(hipGetLastError());
    return HSR_OK;
}

extern "C" int hsr_compact_append_rows(int P, const uint8_t* keep, int keep_is_scanned, int n_tables, const hsr_row_table* tables,
                                       int n_append, int* out_rows, char* scratch, size_t scratch_bytes, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || n_append < 0 || n_tables < 0 || n_tables > HSR_MAX_ROW_TABLES || (n_tables > 0 && !tables) || !out_rows) {
        hsr_set_error("compact_append_rows: invalid arguments (P=%d n_append=%d n_tables=%d, at most %d tables)", P, n_append, n_tables,
                      HSR_MAX_ROW_TABLES);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    RowTables tb;
    tb.n = n_tables;
    for (int t = 0; t < n_tables; t++) {
        tb.t[t] = tables[t];
        if (tb.t[t].cols < 1 || !tb.t[t].dst || (P > 0 && !tb.t[t].src)) {
            hsr_set_error("compact_append_rows: table %d needs cols >= 1, dst and (for P > 0) src", t);
            return HSR_ERR_INVALID_ARGUMENT;
        }
    }
    if (!scratch || scratch_bytes < hsr_compact_scratch_bytes(P)) {
        hsr_set_error("compact_append_rows: scratch too small: %zu bytes needed", hsr_compact_scratch_bytes(P));
        return HSR_ERR_BUFFER_TOO_SMALL;
    }
    unsigned* counts = reinterpret_cast<unsigned*>(scratch);
    const int nblk = (P + DB - 1) / DB;
    if (keep && !keep_is_scanned) {   // a caller-made mask: count and scan it here
        if (P > 0) count_keep_kernel<<<nblk, DB, 0, stream>>>(P, keep, counts);
        scan_counts_kernel<<<1, 1024, 0, stream>>>(nblk, counts, out_rows);
    }
    const int ablk = (n_append + DB - 1) / DB;
    if (nblk + ablk > 0 && n_tables > 0)
        compact_append_kernel<<<nblk + ablk, DB, 0, stream>>>(P, n_append, keep, counts, out_rows, tb);
    add_int_kernel<<<1, 1, 0, stream>>>(out_rows, keep ? 0 : P, n_append, keep ? 0 : 1);   // out_rows = kept + appended
    HSR_HIP_CHECK(hipGetLastError());
    return HSR_OK;
}

extern "C" size_t hsr_densify_scratch_bytes(int H, int W)
{
    if (H < 1 || W < 1) return 4096;
    const size_t nblk = ((size_t)H * W + DB - 1) / DB;
    return 256 + dalign(256 * sizeof(unsigned)) + dalign(nblk * sizeof(unsigned)) + 1024;
}

extern "C" int hsr_densify_frame(int H, int W, const float* silhouette, const float* render_depth, const float* gt_depth,
                                 const float* color, float fx, float fy, float cx, float cy, const float* c2w, float sil_thres,
                                 float depth_factor, int capacity, int* out_count, float* out_means3D, float* out_rgb,
                                 float* out_log_scales, float* out_mean_sq_dist, uint8_t* out_mask, float* out_median, char* scratch,
                                 size_t scratch_bytes, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (H < 1 || W < 1 || (size_t)H * W > 0x7fffffffu || !silhouette || !render_depth || !gt_depth || !color || !c2w || !out_count) {
        hsr_set_error("densify_frame: invalid sizes H=%d W=%d or NULL silhouette/render_depth/gt_depth/color/c2w/out_count", H, W);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (capacity < 0 || (capacity > 0 && (!out_means3D || !out_rgb || !out_log_scales))) {
        hsr_set_error("densify_frame: capacity=%d needs out_means3D, out_rgb and out_log_scales", capacity);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!scratch || scratch_bytes < hsr_densify_scratch_bytes(H, W)) {
        hsr_set_error("densify_frame: scratch too small: %zu bytes needed", hsr_densify_scratch_bytes(H, W));
        return HSR_ERR_BUFFER_TOO_SMALL;
    }
    const int N = H * W;
    const int nblk = (N + DB - 1) / DB;
    SelectState* st = reinterpret_cast<SelectState*>(scratch);
    unsigned* hist = reinterpret_cast<unsigned*>(scratch + 256);
    unsigned* counts = reinterpret_cast<unsigned*>(scratch + 256 + dalign(256 * sizeof(unsigned)));
    select_init_kernel<<<1, DB, 0, stream>>>(st, N, hist);
    const int hblk = (N + DB * 8 - 1) / (DB * 8);
    for (int shift = 24; shift >= 0; shift -= 8) {
        select_hist_kernel<<<hblk, DB, 0, stream>>>(gt_depth, render_depth, N, shift, st, hist);
        select_pick_kernel<<<1, DB, 0, stream>>>(st, hist, shift, out_median);
    }
    mask_count_kernel<<<nblk, DB, 0, stream>>>(silhouette, render_depth, gt_depth, N, sil_thres, depth_factor, st, counts, out_mask);
    scan_counts_kernel<<<1, 1024, 0, stream>>>(nblk, counts, out_count);
    if (capacity > 0) {
        Frame f{fx, fy, cx, cy};
        emit_points_kernel<<<nblk, DB, 0, stream>>>(silhouette, render_depth, gt_depth, color, W, N, f, c2w, sil_thres, depth_factor, st,
                                                    counts, capacity, out_means3D, out_rgb, out_log_scales, out_mean_sq_dist);
    }
    HSR_HIP_CHECK(hipGetLastError());
    return HSR_OK;
}
