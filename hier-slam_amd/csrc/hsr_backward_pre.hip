// hsr_backward_pre.hip — per-Gaussian chain rule back to the rasterizer's inputs, gfx950.
//
// One fused kernel for what the reference runs as two (computeCov2DCUDA, backward.cu:144-274, then
// preprocessCUDA, backward.cu:346-412): dL_dconic -> dL_dcov2D -> dL_dcov3D and the covariance part
// of dL_dmean3D; dL_dmean2D -> dL_dmean3D through the projection; dL_ddepth -> dL_dmean3D through the
// view matrix; SH backward; cov3D -> scale / (un-normalised) quaternion.  Fusing keeps dL_dcov3D and
// the three dL_dmean3D contributions in registers (one write instead of a write + two
// read-modify-writes) and writes zeros for culled Gaussians, so callers need no zero-fill.
// Memory-bound streaming: ~130 B read + ~90 B written per visible Gaussian.
#include "hsr_tile_common.h"

namespace {

__device__ __constant__ float B_SH_C0 = 0.28209479177387814f;
__device__ __constant__ float B_SH_C1 = 0.4886025119029199f;
__device__ __constant__ float B_SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                            -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float B_SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                            0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                            -0.5900435899266435f};

struct M3 {
    float c[3][3];  // c[col][row], glm convention
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

// SH backward (reference computeColorFromSH, backward.cu:20-139); returns the dL_dmean contribution
__device__ void sh_backward(int idx, int deg, int max_coeffs, float mx, float my, float mz, const float* campos,
                            const float* __restrict__ shs, const uint8_t* __restrict__ clamped,
                            const float* __restrict__ dL_dcolor, float* __restrict__ dL_dshs, float& gmx, float& gmy, float& gmz)
{
    const float ox = mx - campos[0], oy = my - campos[1], oz = mz - campos[2];
    const float len = sqrtf(ox * ox + oy * oy + oz * oz);
    const float x = ox / len, y = oy / len, z = oz / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float* dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
    float dRGB[3];
#pragma unroll
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0.f : 1.f);
    float ddx[3] = {0, 0, 0}, ddy[3] = {0, 0, 0}, ddz[3] = {0, 0, 0};
#define SH(i) sh[(i) * 3 + c]
#define DSH(i, v)                                           \
    {                                                       \
        const float vv = (v);                               \
        for (int c = 0; c < 3; c++) dsh[(i) * 3 + c] = vv * dRGB[c]; \
    }
    DSH(0, B_SH_C0);
    if (deg > 0) {
        DSH(1, -B_SH_C1 * y); DSH(2, B_SH_C1 * z); DSH(3, -B_SH_C1 * x);
        for (int c = 0; c < 3; c++) { ddx[c] = -B_SH_C1 * SH(3); ddy[c] = -B_SH_C1 * SH(1); ddz[c] = B_SH_C1 * SH(2); }
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            DSH(4, B_SH_C2[0] * xy); DSH(5, B_SH_C2[1] * yz); DSH(6, B_SH_C2[2] * (2.f * zz - xx - yy));
            DSH(7, B_SH_C2[3] * xz); DSH(8, B_SH_C2[4] * (xx - yy));
            for (int c = 0; c < 3; c++) {
                ddx[c] += B_SH_C2[0] * y * SH(4) + B_SH_C2[2] * 2.f * -x * SH(6) + B_SH_C2[3] * z * SH(7) + B_SH_C2[4] * 2.f * x * SH(8);
                ddy[c] += B_SH_C2[0] * x * SH(4) + B_SH_C2[1] * z * SH(5) + B_SH_C2[2] * 2.f * -y * SH(6) + B_SH_C2[4] * 2.f * -y * SH(8);
                ddz[c] += B_SH_C2[1] * y * SH(5) + B_SH_C2[2] * 2.f * 2.f * z * SH(6) + B_SH_C2[3] * x * SH(7);
            }
            if (deg > 2) {
                DSH(9, B_SH_C3[0] * y * (3.f * xx - yy)); DSH(10, B_SH_C3[1] * xy * z);
                DSH(11, B_SH_C3[2] * y * (4.f * zz - xx - yy)); DSH(12, B_SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                DSH(13, B_SH_C3[4] * x * (4.f * zz - xx - yy)); DSH(14, B_SH_C3[5] * z * (xx - yy));
                DSH(15, B_SH_C3[6] * x * (xx - 3.f * yy));
                for (int c = 0; c < 3; c++) {
                    ddx[c] += (B_SH_C3[0] * SH(9) * 3.f * 2.f * xy + B_SH_C3[1] * SH(10) * yz + B_SH_C3[2] * SH(11) * -2.f * xy +
                               B_SH_C3[3] * SH(12) * -3.f * 2.f * xz + B_SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                               B_SH_C3[5] * SH(14) * 2.f * xz + B_SH_C3[6] * SH(15) * 3.f * (xx - yy));
                    ddy[c] += (B_SH_C3[0] * SH(9) * 3.f * (xx - yy) + B_SH_C3[1] * SH(10) * xz +
                               B_SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + B_SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                               B_SH_C3[4] * SH(13) * -2.f * xy + B_SH_C3[5] * SH(14) * -2.f * yz + B_SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                    ddz[c] += (B_SH_C3[1] * SH(10) * xy + B_SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                               B_SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + B_SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                               B_SH_C3[5] * SH(14) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    const float dLx = ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2];
    const float dLy = ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2];
    const float dLz = ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2];
    // dnormvdv (auxiliary.h:107-118)
    const float sum2 = ox * ox + oy * oy + oz * oz;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    gmx += ((+sum2 - ox * ox) * dLx - oy * ox * dLy - oz * ox * dLz) * invsum32;
    gmy += (-ox * oy * dLx + (sum2 - oy * oy) * dLy - oz * oy * dLz) * invsum32;
    gmz += (-ox * oz * dLx - oy * oz * dLy + (sum2 - oz * oz) * dLz) * invsum32;
}

// Zero-fill of the packed gradient rows, VISIBLE Gaussians only (radii > 0): the tile kernel adds into no other row and the unpack
// below reads no other row, so the fill — like the unpack — costs bytes in proportion to what the camera sees, not to the size of
// the map (the reference zero-fills P x (K + 28) floats per backward whatever is visible, rasterize_points.cu:378-388).  Block =
// 256 consecutive rows = one contiguous slab of 256 * stride floats, written as float4 (stride is a multiple of 16 floats).
__global__ void __launch_bounds__(256) zero_visible_rows_kernel(int P, const int* __restrict__ radii, float* __restrict__ grow, int stride)
{
    __shared__ uint8_t s_vis[256];
    const int g0 = blockIdx.x * 256;
    const int ng = min(256, P - g0);
    s_vis[threadIdx.x] = (int)threadIdx.x < ng && radii[g0 + threadIdx.x] > 0;
    __syncthreads();
    const int q = stride >> 2;                          // float4 per row
    float4* dst = reinterpret_cast<float4*>(grow + (size_t)g0 * stride);
    const int total = ng * q;
    const int drow = 256 / q, dcol = 256 - drow * q;    // e += 256: row += drow, column += dcol (then one carry)
    int row = (int)threadIdx.x / q, col = (int)threadIdx.x - row * q;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = threadIdx.x; e < total; e += 256) {
        if (s_vis[row]) dst[e] = z;
        row += drow; col += dcol;
        if (col >= q) { col -= q; row++; }
    }
}

// KC < 0: legacy mode (sums already accumulated atomically in dL_dmean2D / dL_dconic / dL_ddepth).
// KC >= 0 (ablate build only): rows mode — this thread first sums the rows of its Gaussian's instances (emission order = ascending
// tile id inside its rect, a FIXED order: gradients are bit-reproducible), writes the six per-Gaussian sums
// the tile kernel used to add atomically, and continues with them in registers.
template <int KC>
__global__ void __launch_bounds__(256) preprocess_backward_kernel(PreBwdArgs a)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    // packed mode: rows of culled Gaussians (radii <= 0) were neither zero-filled nor added into (zero_visible_rows_kernel): they
    // are not read either — their gradients are the zeros written below
    __shared__ uint8_t s_vis[256];
    if (KC < 0 && a.grow) {
        s_vis[threadIdx.x] = idx < a.P && a.radii[idx] > 0;
        __syncthreads();
    }
    if (KC < 0 && a.grow && a.K > 0 && a.out_semantics) {
        // packed mode: the block unpacks the semantic columns of its 256 rows cooperatively — consecutive
        // lanes read consecutive floats of a row and write one contiguous [256, K] slab of dL_dsemantics
        // (row, column) of element e advance incrementally — no division in the loop — and four loads are in flight per lane
        const int g0 = blockIdx.x * 256;
        const int ng = min(256, a.P - g0);
        const int K = a.K, total = ng * K;
        const int dgi = 256 / K, dc = 256 - dgi * K;          // e += 256: row += dgi, column += dc (then one carry)
        int gi = (int)threadIdx.x / K, c = (int)threadIdx.x - gi * K;
        const float* src = a.grow + (size_t)g0 * a.grow_stride;   // + the column of channel c: hsr_grow_col
        float* dst = a.out_semantics + (size_t)g0 * K;
        int e = threadIdx.x;
        for (; e + 768 < total; e += 1024) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                // unconditional load (a culled Gaussian's lanes re-read the slab's first word, value discarded): a load under a
                // divergent condition would be waited for where it is issued instead of four being in flight
                const bool vs = s_vis[gi];
                const float x = src[vs ? (size_t)gi * a.grow_stride + hsr_grow_col(a.grow_layout, K, c) : (size_t)0];
                v[u] = vs ? x : 0.f;
                gi += dgi; c += dc;
                if (c >= K) { c -= K; gi++; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) dst[e + 256 * u] = v[u];
        }
        for (; e < total; e += 256) {
            const bool vs = s_vis[gi];
            const float x = src[vs ? (size_t)gi * a.grow_stride + hsr_grow_col(a.grow_layout, K, c) : (size_t)0];
            dst[e] = vs ? x : 0.f;
            gi += dgi; c += dc;
            if (c >= K) { c -= K; gi++; }
        }
    }
    if (idx >= a.P) return;
    float g_m2x = 0, g_m2y = 0, g_cx = 0, g_cy = 0, g_cw = 0, g_depth = 0;
    if (KC >= 0) {
        constexpr int NCHP = 16 * ((KC + 5 + 15) / 16);
        constexpr int ROW = 8 + NCHP;
        float racc[ROW];
#pragma unroll
        for (int c = 0; c < ROW; c++) racc[c] = 0.f;
        const uint32_t beg = idx == 0 ? 0u : a.point_offsets[idx - 1], end = a.point_offsets[idx];
        for (uint32_t u = beg; u < end; u++) {
            const float4* r = reinterpret_cast<const float4*>(a.rows + (size_t)a.inv[u] * ROW);
#pragma unroll
            for (int q = 0; q < ROW / 4; q++) {
                const float4 v = r[q];
                racc[4 * q] += v.x; racc[4 * q + 1] += v.y; racc[4 * q + 2] += v.z; racc[4 * q + 3] += v.w;
            }
        }
        constexpr int KCC = KC < 0 ? 0 : KC;
        g_m2x = racc[0]; g_m2y = racc[1]; g_cx = racc[2]; g_cy = racc[3]; g_cw = racc[4];
        g_depth = racc[6] + racc[8 + KCC + 3];
        a.out_mean2D[3 * idx] = g_m2x; a.out_mean2D[3 * idx + 1] = g_m2y; a.out_mean2D[3 * idx + 2] = 0.f;
        if (a.out_conic) reinterpret_cast<float4*>(a.out_conic)[idx] = make_float4(g_cx, g_cy, 0.f, g_cw);
        a.out_opacity[idx] = racc[5] + racc[8 + KCC + 4];
        a.out_color[3 * idx] = racc[8 + KCC]; a.out_color[3 * idx + 1] = racc[8 + KCC + 1]; a.out_color[3 * idx + 2] = racc[8 + KCC + 2];
        if (a.out_depth) a.out_depth[idx] = g_depth;
#pragma unroll
        for (int c = 0; c < KCC; c++)
            if (c < a.K) a.out_semantics[(size_t)idx * a.K + c] = racc[8 + c];
    } else if (a.grow) {
        // packed mode: unpack this Gaussian's atomically accumulated row into the reference's arrays
        // a culled Gaussian's lane re-reads the first row of the block's slab (unconditional loads, values discarded)
        const bool vis = s_vis[threadIdx.x];
        const size_t row = vis ? (size_t)idx : (size_t)blockIdx.x * 256;
        const float4* r = reinterpret_cast<const float4*>(a.grow + row * a.grow_stride);
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 l0 = r[0], l1 = r[1];
        const float4 r0 = vis ? l0 : zero4, r1 = vis ? l1 : zero4;
        float d_r = 0.f, d_g = 0.f, d_b = 0.f, d_dep = 0.f, d_op = 0.f;
        if (!a.geo) {
            // r, g, b, depth, opacity (direct): channel columns K .. K + 4, five consecutive row columns under either layout
            const float* dr = a.grow + row * a.grow_stride + hsr_grow_col(a.grow_layout, a.K, a.K);
            const float x0 = dr[0], x1 = dr[1], x2 = dr[2], x3 = dr[3], x4 = dr[4];
            if (vis) { d_r = x0; d_g = x1; d_b = x2; d_dep = x3; d_op = x4; }
        }
        g_m2x = r0.x; g_m2y = r0.y; g_cx = r0.z; g_cy = r0.w; g_cw = r1.x;
        g_depth = r1.z + d_dep;
        a.out_mean2D[3 * idx] = g_m2x; a.out_mean2D[3 * idx + 1] = g_m2y; a.out_mean2D[3 * idx + 2] = 0.f;
        if (a.out_conic) reinterpret_cast<float4*>(a.out_conic)[idx] = make_float4(g_cx, g_cy, 0.f, g_cw);
        if (a.out_opacity) a.out_opacity[idx] = r1.y + d_op;
        if (a.out_color) { a.out_color[3 * idx] = d_r; a.out_color[3 * idx + 1] = d_g; a.out_color[3 * idx + 2] = d_b; }
        if (a.out_depth) a.out_depth[idx] = g_depth;
    } else {
        g_m2x = a.dL_dmean2D[3 * idx]; g_m2y = a.dL_dmean2D[3 * idx + 1];
        g_cx = a.dL_dconic[4 * idx]; g_cy = a.dL_dconic[4 * idx + 1]; g_cw = a.dL_dconic[4 * idx + 3];
        g_depth = a.dL_ddepth[idx];
    }
    float gmx = 0, gmy = 0, gmz = 0;
    float dcov[6] = {0, 0, 0, 0, 0, 0};
    float dsc[3] = {0, 0, 0};
    float dq[4] = {0, 0, 0, 0};
    const bool visible = a.radii[idx] > 0;
    if (visible) {
        const float mx = a.means3D[3 * idx], my = a.means3D[3 * idx + 1], mz = a.means3D[3 * idx + 2];
        const float* vm = a.viewmatrix;
        const float* proj = a.projmatrix;
        // ---- conic -> cov2D -> cov3D, mean (backward.cu:144-274) ----
        {
            const float* cov3D = a.cov3Ds + 6 * (size_t)idx;
            const float dcx = g_cx, dcy = g_cy, dcz = g_cw;
            float tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
            float ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
            const float tz_ = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
            const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
            const float txtz = tx / tz_, tytz = ty / tz_;
            tx = fminf(limx, fmaxf(-limx, txtz)) * tz_;
            ty = fminf(limy, fmaxf(-limy, tytz)) * tz_;
            const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
            const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
            const float h_x = a.focal_x, h_y = a.focal_y;
            const M3 J = {{{h_x / tz_, 0.0f, -(h_x * tx) / (tz_ * tz_)}, {0.0f, h_y / tz_, -(h_y * ty) / (tz_ * tz_)}, {0, 0, 0}}};
            const M3 Wm = {{{vm[0], vm[4], vm[8]}, {vm[1], vm[5], vm[9]}, {vm[2], vm[6], vm[10]}}};
            const M3 Vrk = {{{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}}};
            const M3 T = m3mul(Wm, J);
            const M3 c2 = m3mul(m3mul(m3t(T), m3t(Vrk)), T);
            const float ca = c2.c[0][0] + 0.3f, cb = c2.c[0][1], cc = c2.c[1][1] + 0.3f;
            const float denom = ca * cc - cb * cb;
            float dL_da = 0, dL_db = 0, dL_dc = 0;
            const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(i, j) T.c[i][j]
#define VV(i, j) Vrk.c[i][j]
            if (denom2inv != 0) {
                dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
                dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
                dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
                dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
                dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
                dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
                dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
                dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
                dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
            }
            const float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                                  (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
            const float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                                  (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
            const float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                                  (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
            const float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                                  (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
            const float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                                  (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
            const float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                                  (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
#undef TT
#undef VV
            const float dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
            const float dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
            const float dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
            const float dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
            const float tz = 1.f / tz_, tz2 = tz * tz, tz3 = tz2 * tz;
            const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
            const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
            const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * tx) * tz3 * dL_dJ02 + (2 * h_y * ty) * tz3 * dL_dJ12;
            // transformVec4x3Transpose (auxiliary.h:89-97)
            gmx = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
            gmy = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
            gmz = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
        }
        // ---- mean2D and depth -> mean3D (backward.cu:372-403) ----
        {
            const float hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
            const float m_w = 1.0f / (hw + 0.0000001f);
            const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
            const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
            const float d2x = g_m2x, d2y = g_m2y;
            gmx += (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
            gmy += (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
            gmz += (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;
            const float mul3 = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
            const float dd = g_depth;
            gmx += (vm[2] - vm[3] * mul3) * dd;
            gmy += (vm[6] - vm[7] * mul3) * dd;
            gmz += (vm[10] - vm[11] * mul3) * dd;
        }
        if (a.shs) sh_backward(idx, a.D, a.M, mx, my, mz, a.campos, a.shs, a.clamped, a.dL_dcolor, a.dL_dsh, gmx, gmy, gmz);
        // ---- cov3D -> scale, rotation (backward.cu:278-341) ----
        if (a.scales) {
            const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            const M3 Rm = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                            {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                            {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
            const float sx = a.scale_modifier * a.scales[3 * idx], sy = a.scale_modifier * a.scales[3 * idx + 1],
                        sz = a.scale_modifier * a.scales[3 * idx + 2];
            M3 S = {{{sx, 0, 0}, {0, sy, 0}, {0, 0, sz}}};
            const M3 Mm = m3mul(S, Rm);
            const M3 dSig = {{{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                              {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                              {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}}};
            M3 M2;
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) M2.c[cc][rr] = 2.0f * Mm.c[cc][rr];
            const M3 dL_dM = m3mul(M2, dSig);
            const M3 Rt = m3t(Rm);
            M3 dMt = m3t(dL_dM);
            dsc[0] = Rt.c[0][0] * dMt.c[0][0] + Rt.c[0][1] * dMt.c[0][1] + Rt.c[0][2] * dMt.c[0][2];
            dsc[1] = Rt.c[1][0] * dMt.c[1][0] + Rt.c[1][1] * dMt.c[1][1] + Rt.c[1][2] * dMt.c[1][2];
            dsc[2] = Rt.c[2][0] * dMt.c[2][0] + Rt.c[2][1] * dMt.c[2][1] + Rt.c[2][2] * dMt.c[2][2];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                dMt.c[0][k] *= sx;
                dMt.c[1][k] *= sy;
                dMt.c[2][k] *= sz;
            }
#define DM(i, j) dMt.c[i][j]
            dq[0] = 2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1));
            dq[1] = 2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1));
            dq[2] = 2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0));
            dq[3] = 2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0));
#undef DM
        }
    }
    a.dL_dmean3D[3 * idx] = gmx;
    a.dL_dmean3D[3 * idx + 1] = gmy;
    a.dL_dmean3D[3 * idx + 2] = gmz;
    if (a.dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    }
    if (a.dL_dscale) {
        a.dL_dscale[3 * idx] = dsc[0];
        a.dL_dscale[3 * idx + 1] = dsc[1];
        a.dL_dscale[3 * idx + 2] = dsc[2];
    }
    if (a.dL_drot) reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(dq[0], dq[1], dq[2], dq[3]);
}

}  // namespace

int hsr_launch_zero_visible_rows(int P, const int* radii, float* grow, int stride, hipStream_t stream)
{
    if (P <= 0) return HSR_OK;
    zero_visible_rows_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, radii, grow, stride);
    return HSR_OK;
}

int hsr_launch_preprocess_backward(const PreBwdArgs& a, hipStream_t stream)
{
    if (a.P <= 0) return HSR_OK;
    const dim3 grid((a.P + 255) / 256), block(256);
    switch (a.rows_kc) {
    case 0: preprocess_backward_kernel<-1><<<grid, block, 0, stream>>>(a); break;
#ifdef HSR_ABLATE   // per-instance rows experiment (experiments/hsr_render_bwd_rows.hip)
    case 11: preprocess_backward_kernel<11><<<grid, block, 0, stream>>>(a); break;
    case 16: preprocess_backward_kernel<16><<<grid, block, 0, stream>>>(a); break;
    case 26: preprocess_backward_kernel<26><<<grid, block, 0, stream>>>(a); break;
    case 27: preprocess_backward_kernel<27><<<grid, block, 0, stream>>>(a); break;
#endif
    default: hsr_set_error("unsupported rows_kc %d", a.rows_kc); return HSR_ERR_INVALID_ARGUMENT;
    }
    return HSR_OK;
}
