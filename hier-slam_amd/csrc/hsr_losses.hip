// hsr_losses.hip — fused loss heads on the rendered maps (gfx950), SURVEY.md §8(f) rank 2.
//
// What it replaces in the reference: the torch eager chains of get_loss_semantic_mlp (scripts/hierslam.py:921-1016) —
// boolean-mask gathers + abs + sum/mean for depth and colour, calc_ssim's five grouped conv2d calls and their autograd
// (utils/slam_external.py:66-97), and per tree level a permute + view + CrossEntropyLoss (log_softmax, nll_loss and
// their backward) over a channel slice of the K logit planes (scripts/hierslam.py:91-111, :963-974).
// Here every head is ONE streaming pass over the planar maps that yields the value and the gradient together:
//   * L1: value = block partial sums -> fixed-order finish; gradient = sign * scale written in the same pass (the
//     selection count of a masked mean comes from a byte-wide pre-pass over the mask);
//   * SSIM: 32x32-pixel tiles with a 5-pixel halo of both images in LDS, the reference's Gaussian applied as two
//     11-tap passes with sliding register windows; the forward pass also writes the three partial-derivative maps, the
//     backward pass is the adjoint correlation of those;
//   * tree cross-entropy: one thread per pixel walks the levels, channels are read with stride H*W (coalesced across
//     the wave), online log-sum-exp, then softmax - onehot straight into the planar gradient — no [H*W, n] permute.
// All HBM-bound; algorithmic bytes per pixel: L1 4C*3 (+1 mask), SSIM 4C*(2 + 3) fwd + 4C*(3 + 2 + 1) bwd, CE 4K*3 + 8L.
#include "hsr_common.h"
#include "../../include/hsr_losses.h"
#include <cmath>

namespace {

constexpr int LB = 256;          // threads per block of the streaming kernels
constexpr int L1_ITEMS = 8;      // pixels per thread (L1)
constexpr int SS_T = 32;         // SSIM tile edge (256 threads x 2x2 outputs)
constexpr int SS_R = 5;          // window radius (11x11)
constexpr int SS_E = SS_T + 2 * SS_R;

__device__ __forceinline__ float block_sum(float v, float* s_red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wv] = v;
    __syncthreads();
    return ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

// ---------------------------------------------------------------- finish: fixed-order sum of per-block partials
// out[k] = scale_k * sum_b partials[b * stride + k];  scale: 1, or *inv (device), see `mode`
__global__ __launch_bounds__(LB) void finish_kernel(const float* __restrict__ partials, int nblocks, int stride, int nout,
                                                    const float* __restrict__ inv, float host_scale, float* __restrict__ out)
{
    __shared__ double s_acc[LB];
    for (int k = 0; k < nout; k++) {
        double acc = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += LB) acc += (double)partials[(size_t)b * stride + k];
        s_acc[threadIdx.x] = acc;
        __syncthreads();
        for (int o = LB / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) s_acc[threadIdx.x] += s_acc[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            double v = s_acc[0] * (double)host_scale;
            if (inv) v *= (double)inv[k];
            out[k] = (float)v;
        }
        __syncthreads();
    }
}

// counts -> reciprocal:  inv[k] = 1 / sum_b partials[b*stride + k]   (NaN-free: 0 count gives +inf, which makes the mean
// NaN exactly like torch's mean over an empty selection and the gradient 0 * inf = NaN only where selected — nowhere)
__global__ __launch_bounds__(LB) void count_finish_kernel(const unsigned* __restrict__ partials, int nblocks, int stride, int nout,
                                                          float* __restrict__ inv)
{
    __shared__ unsigned long long s_acc[LB];
    for (int k = 0; k < nout; k++) {
        unsigned long long acc = 0;
        for (int b = threadIdx.x; b < nblocks; b += LB) acc += partials[(size_t)b * stride + k];
        s_acc[threadIdx.x] = acc;
        __syncthreads();
        for (int o = LB / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) s_acc[threadIdx.x] += s_acc[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) inv[k] = 1.0f / (float)s_acc[0];
        __syncthreads();
    }
}

// ---------------------------------------------------------------- L1
__global__ __launch_bounds__(LB) void mask_count_kernel(const uint8_t* __restrict__ mask, int N, unsigned* __restrict__ partials)
{
    __shared__ unsigned s_red[4];
    unsigned c = 0;
    for (int i = blockIdx.x * LB * L1_ITEMS + threadIdx.x, it = 0; it < L1_ITEMS; it++, i += LB)
        if (i < N) c += mask[i] != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// grid: (blocks over N, C).  scale = host_scale * (inv ? inv[0] : 1)
__global__ __launch_bounds__(LB) void l1_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                const uint8_t* __restrict__ mask, int N, const float* __restrict__ inv,
                                                float host_scale, float* __restrict__ grad, float* __restrict__ partials)
{
    __shared__ float s_red[4];
    const size_t plane = (size_t)blockIdx.y * N;
    const float scale = host_scale * (inv ? inv[0] : 1.0f);
    float acc = 0.f;
    for (int i = blockIdx.x * LB * L1_ITEMS + threadIdx.x, it = 0; it < L1_ITEMS; it++, i += LB) {
        if (i >= N) break;
        const bool sel = mask ? mask[i] != 0 : true;
        const float d = pred[plane + i] - gt[plane + i];
        // unselected pixels contribute nothing, whatever they hold (the reference's mask exists to exclude NaN depths)
        acc += sel ? fabsf(d) : 0.f;
        if (grad) grad[plane + i] = sel ? (d > 0.f ? scale : (d < 0.f ? -scale : 0.f)) : 0.f;
    }
    const float tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// ---------------------------------------------------------------- SSIM
// The reference's window is the outer product of a normalised 11-tap Gaussian with itself (utils/slam_external.py:60-62),
// so the 121-tap correlation is evaluated as a horizontal then a vertical 11-tap pass (the two differ from the 2-D form by
// one fp32 rounding per weight).  Tile = 32x32 outputs, 42x42 inputs in LDS; both passes slide a 14-value register window
// over 4 consecutive outputs, so a pixel costs ~27 LDS reads and ~130 FMAs instead of 242 and 605.
struct Gauss { float g[11]; };

template <int NQ>
__device__ __forceinline__ void blur4(const float (&v)[NQ][14], const Gauss& gw, float (&out)[NQ][4])
{
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) acc = fmaf(gw.g[k], v[q][o + k], acc);
            out[q][o] = acc;
        }
}

// grid (tiles_x, tiles_y, C); block 256
__global__ __launch_bounds__(LB) void ssim_forward_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                                                          Gauss gw, float* __restrict__ d_mu1, float* __restrict__ d_s11,
                                                          float* __restrict__ d_s12, float* __restrict__ partials)
{
    __shared__ float s_x[SS_E][SS_E + 1], s_y[SS_E][SS_E + 1];
    __shared__ float s_h[5][SS_E][SS_T + 1];   // horizontally blurred x, y, xx, yy, xy
    __shared__ float s_red[4];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * SS_T - SS_R, y0 = blockIdx.y * SS_T - SS_R;
    for (int i = threadIdx.x; i < SS_E * SS_E; i += LB) {
        const int ly = i / SS_E, lx = i - ly * SS_E;
        const int gx = x0 + lx, gy = y0 + ly;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;   // zero padding (conv2d padding = 5)
        const size_t o = plane + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
        const float a = img1[o], b = img2[o];
        s_x[ly][lx] = in ? a : 0.f;
        s_y[ly][lx] = in ? b : 0.f;
    }
    __syncthreads();
    // pass 1: rows 0..41, 8 groups of 4 columns each
    for (int item = threadIdx.x; item < SS_E * (SS_T / 4); item += LB) {
        const int r = item / (SS_T / 4), c0 = (item % (SS_T / 4)) * 4;
        float v[5][14];
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const float a = s_x[r][c0 + k], b = s_y[r][c0 + k];
            v[0][k] = a; v[1][k] = b; v[2][k] = a * a; v[3][k] = b * b; v[4][k] = a * b;
        }
        float o4[5][4];
        blur4<5>(v, gw, o4);
#pragma unroll
        for (int q = 0; q < 5; q++)
#pragma unroll
            for (int o = 0; o < 4; o++) s_h[q][r][c0 + o] = o4[q][o];
    }
    __syncthreads();
    // pass 2: thread = (column, group of 4 rows)
    const int col = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 4;
    float v[5][14];
#pragma unroll
    for (int q = 0; q < 5; q++)
#pragma unroll
        for (int k = 0; k < 14; k++) v[q][k] = s_h[q][r0 + k][col];
    float m[5][4];
    blur4<5>(v, gw, m);
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    float acc = 0.f;
    const int px = blockIdx.x * SS_T + col;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int py = blockIdx.y * SS_T + r0 + o;
        const float m1 = m[0][o], m2 = m[1][o];
        const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
        const float sig1 = m[2][o] - mu1_sq, sig2 = m[3][o] - mu2_sq, sig12 = m[4][o] - mu12;
        const float A1 = 2.f * mu12 + c1, A2 = 2.f * sig12 + c2, B1 = mu1_sq + mu2_sq + c1, B2 = sig1 + sig2 + c2;
        const float inv12 = 1.0f / (B1 * B2);
        const float smap = (A1 * A2) * inv12;
        const bool live = px < W && py < H;
        if (live && d_mu1) {
            const size_t off = plane + (size_t)py * W + px;
            d_mu1[off] = (2.f * m2 * A2 - 2.f * m2 * A1) * inv12 - smap * (2.f * m1 / B1 - 2.f * m1 / B2);
            d_s11[off] = -smap / B2;
            d_s12[off] = 2.f * A1 * inv12;
        }
        acc += live ? smap : 0.f;
    }
    const float tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
}

// adjoint of the zero-padded correlation with a symmetric window = the same correlation of the three partial-derivative
// maps (which are zero outside the image)
__global__ __launch_bounds__(LB) void ssim_backward_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                                                           Gauss gw, const float* __restrict__ d_mu1, const float* __restrict__ d_s11,
                                                           const float* __restrict__ d_s12, float inv_n, float* __restrict__ grad)
{
    __shared__ float s_in[3][SS_E][SS_E + 1];
    __shared__ float s_h[3][SS_E][SS_T + 1];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * SS_T - SS_R, y0 = blockIdx.y * SS_T - SS_R;
    for (int i = threadIdx.x; i < SS_E * SS_E; i += LB) {
        const int ly = i / SS_E, lx = i - ly * SS_E;
        const int gx = x0 + lx, gy = y0 + ly;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const size_t o = plane + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
        const float m = d_mu1[o], a = d_s11[o], b = d_s12[o];
        s_in[0][ly][lx] = in ? m : 0.f;
        s_in[1][ly][lx] = in ? a : 0.f;
        s_in[2][ly][lx] = in ? b : 0.f;
    }
    __syncthreads();
    for (int item = threadIdx.x; item < SS_E * (SS_T / 4); item += LB) {
        const int r = item / (SS_T / 4), c0 = (item % (SS_T / 4)) * 4;
        float v[3][14];
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int k = 0; k < 14; k++) v[q][k] = s_in[q][r][c0 + k];
        float o4[3][4];
        blur4<3>(v, gw, o4);
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int o = 0; o < 4; o++) s_h[q][r][c0 + o] = o4[q][o];
    }
    __syncthreads();
    const int col = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 4;
    float v[3][14];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int k = 0; k < 14; k++) v[q][k] = s_h[q][r0 + k][col];
    float gsum[3][4];
    blur4<3>(v, gw, gsum);
    const int px = blockIdx.x * SS_T + col;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int py = blockIdx.y * SS_T + r0 + o;
        if (px < W && py < H) {
            const size_t off = plane + (size_t)py * W + px;
            grad[off] = (gsum[0][o] + 2.f * img1[off] * gsum[1][o] + img2[off] * gsum[2][o]) * inv_n;
        }
    }
}

// ---------------------------------------------------------------- tree cross-entropy
struct Levels {
    int n;
    int begin[HSR_LOSS_MAX_LEVELS], size[HSR_LOSS_MAX_LEVELS];
    float weight[HSR_LOSS_MAX_LEVELS];
};

__global__ __launch_bounds__(LB) void ce_count_kernel(const int64_t* __restrict__ labels, int N, int num_levels, int ignore_index,
                                                      unsigned* __restrict__ partials)
{
    __shared__ unsigned s_red[4];
    const int i = blockIdx.x * LB + threadIdx.x;
    for (int l = 0; l < num_levels; l++) {
        unsigned c = (i < N && labels[(size_t)l * N + i] != (int64_t)ignore_index) ? 1u : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) partials[(size_t)blockIdx.x * HSR_LOSS_MAX_LEVELS + l] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    }
}

__global__ __launch_bounds__(LB) void tree_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int N, int K,
                                                     Levels lv, int ignore_index, const float* __restrict__ inv_count,
                                                     float* __restrict__ grad, float* __restrict__ partials)
{
    __shared__ float s_red[4];
    const int i = blockIdx.x * LB + threadIdx.x;
    const bool live = i < N;
    const size_t p = live ? (size_t)i : 0;
    int covered_end = 0;
    for (int l = 0; l < lv.n; l++) {
        const int b = lv.begin[l], n = lv.size[l];
        const int64_t lab64 = labels[(size_t)l * N + p];
        const bool valid = live && lab64 != (int64_t)ignore_index;
        const int lab = (int)lab64;
        // online log-sum-exp over the level's channels (stride N: consecutive lanes read consecutive pixels)
        float m = -INFINITY, s = 0.f, picked = 0.f;
        for (int c = 0; c < n; c++) {
            const float z = logits[(size_t)(b + c) * N + p];
            const float nm = fmaxf(m, z);
            s = s * expf(m - nm) + expf(z - nm);
            m = nm;
            picked = c == lab ? z : picked;
        }
        const float lse = m + logf(s);
        const float loss = valid ? lse - picked : 0.f;
        if (grad && live) {
            const float sc = valid ? lv.weight[l] * inv_count[l] : 0.f;
            const float inv_s = 1.0f / s;
            for (int c = 0; c < n; c++) {
                const float z = logits[(size_t)(b + c) * N + p];   // second read comes from L2
                const float sm = expf(z - m) * inv_s;
                grad[(size_t)(b + c) * N + p] = (sm - (c == lab ? 1.f : 0.f)) * sc;
            }
        }
        const float tot = block_sum(loss, s_red);
        if (threadIdx.x == 0) partials[(size_t)blockIdx.x * HSR_LOSS_MAX_LEVELS + l] = tot;
        covered_end = b + n;
    }
    if (grad && live)
        for (int c = covered_end; c < K; c++) grad[(size_t)c * N + p] = 0.f;   // channels behind the last level
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int check_scratch(const char* what, char* scratch, size_t have, size_t need)
{
    if (!scratch || have < need) {
        hsr_set_error("%s: scratch too small: %zu bytes needed, %zu given", what, need, have);
        return HSR_ERR_BUFFER_TOO_SMALL;
    }
    return HSR_OK;
}

}  // namespace

extern "C" size_t hsr_loss_scratch_bytes(int channels, int H, int W)
{
    if (channels < 1 || H < 1 || W < 1) return 1024;
    const size_t N = (size_t)H * W;
    const size_t tiles = (size_t)((W + SS_T - 1) / SS_T) * ((H + SS_T - 1) / SS_T);
    const size_t blocks = (N + LB - 1) / LB + 1;
    // SSIM: three partial-derivative maps + one partial per tile;  CE: MAX_LEVELS partials per block, twice;  L1: small
    size_t need = align256(3 * (size_t)channels * N * sizeof(float)) + align256((size_t)channels * tiles * sizeof(float));
    const size_t ce = 2 * align256(blocks * HSR_LOSS_MAX_LEVELS * sizeof(float)) + 256;
    const size_t l1 = 2 * align256((size_t)channels * blocks * sizeof(float)) + 256;
    if (ce > need) need = ce;
    if (l1 > need) need = l1;
    return need + 1024;
}

extern "C" int hsr_loss_l1(int C, int H, int W, const float* pred, const float* gt, const uint8_t* mask, int reduction, float* out_loss,
                           float* out_grad, char* scratch, size_t scratch_bytes, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (C < 1 || H < 1 || W < 1 || (size_t)H * W > 0x7fffffffu || !pred || !gt || !out_loss) {
        hsr_set_error("loss_l1: invalid sizes C=%d H=%d W=%d or NULL pred/gt/out_loss", C, H, W);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (reduction != HSR_LOSS_SUM && reduction != HSR_LOSS_MEAN) {
        hsr_set_error("loss_l1: reduction must be HSR_LOSS_SUM or HSR_LOSS_MEAN");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    const int N = H * W;
    const int nb = (N + LB * L1_ITEMS - 1) / (LB * L1_ITEMS);
    const size_t part_bytes = align256((size_t)C * nb * sizeof(float));
    int rc = check_scratch("loss_l1", scratch, scratch_bytes, 2 * part_bytes + 256);
    if (rc != HSR_OK) return rc;
    float* partials = reinterpret_cast<float*>(scratch);
    unsigned* cparts = reinterpret_cast<unsigned*>(scratch + part_bytes);
    float* inv = reinterpret_cast<float*>(scratch + 2 * part_bytes);
    const float* inv_arg = nullptr;
    float host_scale = 1.0f;
    if (reduction == HSR_LOSS_MEAN) {
        if (mask) {
            mask_count_kernel<<<nb, LB, 0, stream>>>(mask, N, cparts);
            count_finish_kernel<<<1, LB, 0, stream>>>(cparts, nb, 1, 1, inv);
            inv_arg = inv;
            host_scale = 1.0f / (float)C;   // the selection is tiled over the C planes
        } else {
            host_scale = (float)(1.0 / ((double)C * N));
        }
    }
    l1_kernel<<<dim3(nb, C), LB, 0, stream>>>(pred, gt, mask, N, inv_arg, host_scale, out_grad, partials);
    finish_kernel<<<1, LB, 0, stream>>>(partials, C * nb, 1, 1, inv_arg, host_scale, out_loss);
    HSR_HIP_CHECK(hipGetLastError());
    return HSR_OK;
}

extern "C" int hsr_loss_ssim(int C, int H, int W, const float* img1, const float* img2, float* out_ssim, float* out_grad, char* scratch,
                             size_t scratch_bytes, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (C < 1 || H < 1 || W < 1 || !img1 || !img2 || !out_ssim) {
        hsr_set_error("loss_ssim: invalid sizes C=%d H=%d W=%d or NULL img1/img2/out_ssim", C, H, W);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    const size_t N = (size_t)H * W;
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, C);
    const size_t tiles = (size_t)grid.x * grid.y * C;
    const size_t map_bytes = align256((size_t)C * N * sizeof(float) * 3);
    int rc = check_scratch("loss_ssim", scratch, scratch_bytes, (out_grad ? map_bytes : 0) + align256(tiles * sizeof(float)));
    if (rc != HSR_OK) return rc;
    // the reference's 1-D window: gaussian(11, 1.5) as float32, normalised in float32 (utils/slam_external.py:54-56)
    Gauss win;
    {
        float sum = 0.f;
        for (int x = 0; x < 11; x++) {
            win.g[x] = (float)std::exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5));
            sum += win.g[x];
        }
        for (int x = 0; x < 11; x++) win.g[x] = win.g[x] / sum;
    }
    float *d_mu1 = nullptr, *d_s11 = nullptr, *d_s12 = nullptr;
    char* cur = scratch;
    if (out_grad) {
        d_mu1 = reinterpret_cast<float*>(cur);
        d_s11 = d_mu1 + (size_t)C * N;
        d_s12 = d_s11 + (size_t)C * N;
        cur += map_bytes;
    }
    float* partials = reinterpret_cast<float*>(cur);
    const float inv_n = (float)(1.0 / ((double)C * (double)N));
    ssim_forward_kernel<<<grid, LB, 0, stream>>>(img1, img2, H, W, win, d_mu1, d_s11, d_s12, partials);
    finish_kernel<<<1, LB, 0, stream>>>(partials, (int)tiles, 1, 1, nullptr, inv_n, out_ssim);
    if (out_grad) ssim_backward_kernel<<<grid, LB, 0, stream>>>(img1, img2, H, W, win, d_mu1, d_s11, d_s12, inv_n, out_grad);
    HSR_HIP_CHECK(hipGetLastError());
    return HSR_OK;
}

extern "C" int hsr_loss_tree_ce(int K, int H, int W, int num_levels, const int* level_sizes, const float* level_weight,
                                const float* logits, const int64_t* labels, int ignore_index, float* out_level_loss, float* out_grad,
                                char* scratch, size_t scratch_bytes, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (K < 1 || H < 1 || W < 1 || (size_t)H * W > 0x7fffffffu || !logits || !labels || !out_level_loss || !level_sizes) {
        hsr_set_error("loss_tree_ce: invalid sizes K=%d H=%d W=%d or NULL logits/labels/level_sizes/out_level_loss", K, H, W);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (num_levels < 1 || num_levels > HSR_LOSS_MAX_LEVELS) {
        hsr_set_error("loss_tree_ce: num_levels=%d outside [1, %d]", num_levels, HSR_LOSS_MAX_LEVELS);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    Levels lv;
    lv.n = num_levels;
    int begin = 0;
    for (int l = 0; l < num_levels; l++) {
        if (level_sizes[l] < 1) {
            hsr_set_error("loss_tree_ce: level %d has %d classes", l, level_sizes[l]);
            return HSR_ERR_INVALID_ARGUMENT;
        }
        lv.begin[l] = begin;
        lv.size[l] = level_sizes[l];
        lv.weight[l] = level_weight ? level_weight[l] : 1.0f;
        begin += level_sizes[l];
    }
    if (begin > K) {
        hsr_set_error("loss_tree_ce: levels cover %d channels but the map has K=%d", begin, K);  // the reference would slice short
        return HSR_ERR_INVALID_ARGUMENT;
    }
    const int N = H * W;
    const int nb = (N + LB - 1) / LB;
    const size_t part_bytes = align256((size_t)nb * HSR_LOSS_MAX_LEVELS * sizeof(float));
    int rc = check_scratch("loss_tree_ce", scratch, scratch_bytes, 2 * part_bytes + 256);
    if (rc != HSR_OK) return rc;
    float* partials = reinterpret_cast<float*>(scratch);
    unsigned* cparts = reinterpret_cast<unsigned*>(scratch + part_bytes);
    float* inv = reinterpret_cast<float*>(scratch + 2 * part_bytes);
    ce_count_kernel<<<nb, LB, 0, stream>>>(labels, N, num_levels, ignore_index, cparts);
    count_finish_kernel<<<1, LB, 0, stream>>>(cparts, nb, HSR_LOSS_MAX_LEVELS, num_levels, inv);
    tree_ce_kernel<<<nb, LB, 0, stream>>>(logits, labels, N, K, lv, ignore_index, inv, out_grad, partials);
    finish_kernel<<<1, LB, 0, stream>>>(partials, nb, HSR_LOSS_MAX_LEVELS, num_levels, inv, 1.0f, out_level_loss);
    HSR_HIP_CHECK(hipGetLastError());
    return HSR_OK;
}
