// hsr_frame_prep.hip — fused rasterizer-input preparation for one frame (gfx950), SURVEY.md §8(f) rank 1.
//
// What it replaces in the reference (all per-iteration torch eager chains, ~12 kernels forward and ~25 backward):
//   transform_to_frame (utils/slam_helpers.py:278-330), build_rotation (utils/slam_external.py:25-42), quat_mult
//   (utils/slam_helpers.py:21-28), transformed_params2rendervar / _semantic / depthplussilhouette
//   (utils/slam_helpers.py:124-139, :195-219, :260-275, :222-239), and what torch.autograd derives for them.
//
// HBM-streaming work: 36-44 B read and 44-72 B written per Gaussian forward, about twice that backward; one thread
// per Gaussian, AoS rows read with the widest aligned access their stride allows.  The only cross-Gaussian step is the
// pose-gradient reduction (sum g, sum g x^T, quaternion-product terms: 16 sums): wave butterfly -> per-block partial
// in scratch -> one-block finish in double, fixed order, so the result is reproducible bit for bit.
// Compiled with -ffp-contract=off: the forward then evaluates exactly the expression tree of the oracle
// (oracle/frame_prep_oracle.py), which matters because out_means3D feeds the rasterizer's integer decisions.
#include "hsr_common.h"
#include "../../include/hsr_frame_prep.h"

namespace {

constexpr int PREP_BLOCK = 256;
constexpr int PREP_BWD_ITEMS = 4;   // Gaussians per thread in the backward kernel: 4x fewer partial rows for the finish
constexpr int PREP_FINISH_BLOCK = 1024;
constexpr int PREP_SUMS = 16;  // 0..2 sum g | 3..11 M[i][j] = sum g_i x_j | 12..15 quat_mult adjoint wrt the camera quaternion

struct Pose {
    float qhat[4], nhat;  // parameter column and its norm
    float q[4];           // F.normalize(qhat)
    float n2, qq[4];      // build_rotation's own normalisation of q
    float R[9], t[3];
};

__device__ __forceinline__ Pose load_pose(const float* cam_unnorm_rots, const float* cam_trans, int num_frames, int time_idx)
{
    Pose p;
#pragma unroll
    for (int c = 0; c < 4; c++) p.qhat[c] = cam_unnorm_rots[c * num_frames + time_idx];
#pragma unroll
    for (int c = 0; c < 3; c++) p.t[c] = cam_trans[c * num_frames + time_idx];
    p.nhat = sqrtf(((p.qhat[0] * p.qhat[0] + p.qhat[1] * p.qhat[1]) + p.qhat[2] * p.qhat[2]) + p.qhat[3] * p.qhat[3]);
    const float dn = fmaxf(p.nhat, 1e-12f);
#pragma unroll
    for (int c = 0; c < 4; c++) p.q[c] = p.qhat[c] / dn;
    p.n2 = sqrtf(((p.q[0] * p.q[0] + p.q[1] * p.q[1]) + p.q[2] * p.q[2]) + p.q[3] * p.q[3]);
#pragma unroll
    for (int c = 0; c < 4; c++) p.qq[c] = p.q[c] / p.n2;
    const float r = p.qq[0], x = p.qq[1], y = p.qq[2], z = p.qq[3];
    p.R[0] = 1.0f - 2.0f * (y * y + z * z); p.R[1] = 2.0f * (x * y - r * z);        p.R[2] = 2.0f * (x * z + r * y);
    p.R[3] = 2.0f * (x * y + r * z);        p.R[4] = 1.0f - 2.0f * (x * x + z * z); p.R[5] = 2.0f * (y * z - r * x);
    p.R[6] = 2.0f * (x * z - r * y);        p.R[7] = 2.0f * (y * z + r * x);        p.R[8] = 1.0f - 2.0f * (x * x + y * y);
    return p;
}

__device__ __forceinline__ float4 quat_mult(const float* a, float4 b)  // slam_helpers.py:21-28, a = q1, b = q2 (w, x, y, z)
{
    float4 o;
    o.x = ((a[0] * b.x - a[1] * b.y) - a[2] * b.z) - a[3] * b.w;
    o.y = ((a[0] * b.y + a[1] * b.x) + a[2] * b.w) - a[3] * b.z;
    o.z = ((a[0] * b.z - a[1] * b.w) + a[2] * b.x) + a[3] * b.y;
    o.w = ((a[0] * b.w + a[1] * b.z) - a[2] * b.y) + a[3] * b.x;
    return o;
}

__device__ __forceinline__ float norm4(float4 v) { return sqrtf(((v.x * v.x + v.y * v.y) + v.z * v.z) + v.w * v.w); }

__device__ __forceinline__ float4 normalize4(float4 v, float* n_out = nullptr)
{
    const float n = norm4(v);
    if (n_out) *n_out = n;
    const float d = fmaxf(n, 1e-12f);
    return make_float4(v.x / d, v.y / d, v.z / d, v.w / d);
}

// adjoint of y = v / |v|:  (g - y (y.g)) / |v|
__device__ __forceinline__ float4 normalize_adjoint(float4 v, float4 g)
{
    const float n = fmaxf(norm4(v), 1e-12f);
    const float4 y = make_float4(v.x / n, v.y / n, v.z / n, v.w / n);
    const float d = y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w;
    return make_float4((g.x - y.x * d) / n, (g.y - y.y * d) / n, (g.z - y.z * d) / n, (g.w - y.w * d) / n);
}

struct PrepArgs {
    int P, S, transform_rots, rot_source, num_frames, time_idx;
    const float *means3D, *unnorm_rotations, *logit_opacities, *log_scales, *cam_unnorm_rots, *cam_trans, *w2c;
};

__global__ __launch_bounds__(PREP_BLOCK) void frame_prep_forward_kernel(PrepArgs a, float* __restrict__ out_means3D,
                                                                       float* __restrict__ out_unnorm_rot,
                                                                       float* __restrict__ out_rotations,
                                                                       float* __restrict__ out_opacities, float* __restrict__ out_scales,
                                                                       float* __restrict__ out_depth_sil)
{
    const int p = blockIdx.x * PREP_BLOCK + threadIdx.x;
    if (p >= a.P) return;
    const Pose ps = load_pose(a.cam_unnorm_rots, a.cam_trans, a.num_frames, a.time_idx);
    const float x0 = a.means3D[3 * p], x1 = a.means3D[3 * p + 1], x2 = a.means3D[3 * p + 2];
    float m[3];
#pragma unroll
    for (int i = 0; i < 3; i++) m[i] = ((x0 * ps.R[3 * i] + x1 * ps.R[3 * i + 1]) + x2 * ps.R[3 * i + 2]) + ps.t[i] * 1.0f;
    out_means3D[3 * p] = m[0]; out_means3D[3 * p + 1] = m[1]; out_means3D[3 * p + 2] = m[2];
    const float4 u = reinterpret_cast<const float4*>(a.unnorm_rotations)[p];
    float4 tr = u;
    if (a.transform_rots) tr = quat_mult(ps.q, normalize4(u));
    if (out_unnorm_rot) reinterpret_cast<float4*>(out_unnorm_rot)[p] = tr;
    reinterpret_cast<float4*>(out_rotations)[p] = normalize4(a.rot_source == HSR_PREP_ROT_PARAMS ? u : tr);
    out_opacities[p] = 1.0f / (1.0f + expf(-a.logit_opacities[p]));
    if (a.S == 1) {
        const float e = expf(a.log_scales[p]);
        out_scales[3 * p] = e; out_scales[3 * p + 1] = e; out_scales[3 * p + 2] = e;
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) out_scales[3 * p + i] = expf(a.log_scales[3 * p + i]);
    }
    if (out_depth_sil) {
        const float z = ((m[0] * a.w2c[8] + m[1] * a.w2c[9]) + m[2] * a.w2c[10]) + a.w2c[11];
        out_depth_sil[3 * p] = z; out_depth_sil[3 * p + 1] = 1.0f; out_depth_sil[3 * p + 2] = z * z;
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(PREP_BLOCK) void frame_prep_backward_kernel(
    PrepArgs a, const float* __restrict__ g_means, const float* __restrict__ g_unnorm, const float* __restrict__ g_rot,
    const float* __restrict__ g_opac, const float* __restrict__ g_scales, const float* __restrict__ g_sil, float* __restrict__ d_means3D,
    float* __restrict__ d_unnorm, float* __restrict__ d_logit, float* __restrict__ d_log_scales, float* __restrict__ partials)
{
    __shared__ float s_part[PREP_BLOCK / 64][PREP_SUMS];
    const Pose ps = load_pose(a.cam_unnorm_rots, a.cam_trans, a.num_frames, a.time_idx);
    float sums[PREP_SUMS];
#pragma unroll
    for (int k = 0; k < PREP_SUMS; k++) sums[k] = 0.0f;
    for (int it = 0; it < PREP_BWD_ITEMS; it++) {
        const int p = (blockIdx.x * PREP_BWD_ITEMS + it) * PREP_BLOCK + threadIdx.x;
        if (p >= a.P) break;
        const float x[3] = {a.means3D[3 * p], a.means3D[3 * p + 1], a.means3D[3 * p + 2]};
        float g[3] = {0.0f, 0.0f, 0.0f};
        if (g_means) { g[0] = g_means[3 * p]; g[1] = g_means[3 * p + 1]; g[2] = g_means[3 * p + 2]; }
        if (g_sil) {  // colours {z, 1, z^2} (slam_helpers.py:234-237): dz = g0 + 2 z g2, chained into the camera-frame mean
            float m[3];
#pragma unroll
            for (int i = 0; i < 3; i++) m[i] = ((x[0] * ps.R[3 * i] + x[1] * ps.R[3 * i + 1]) + x[2] * ps.R[3 * i + 2]) + ps.t[i];
            const float z = ((m[0] * a.w2c[8] + m[1] * a.w2c[9]) + m[2] * a.w2c[10]) + a.w2c[11];
            const float dz = g_sil[3 * p] + 2.0f * z * g_sil[3 * p + 2];
            g[0] += dz * a.w2c[8]; g[1] += dz * a.w2c[9]; g[2] += dz * a.w2c[10];
        }
        if (d_means3D) {
#pragma unroll
            for (int j = 0; j < 3; j++) d_means3D[3 * p + j] = (g[0] * ps.R[j] + g[1] * ps.R[3 + j]) + g[2] * ps.R[6 + j];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            sums[i] += g[i];
#pragma unroll
            for (int j = 0; j < 3; j++) sums[3 + 3 * i + j] += g[i] * x[j];
        }
        // rotations
        const float4 u = reinterpret_cast<const float4*>(a.unnorm_rotations)[p];
        const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float4 gr = g_rot ? reinterpret_cast<const float4*>(g_rot)[p] : zero4;
        float4 gtr = g_unnorm ? reinterpret_cast<const float4*>(g_unnorm)[p] : zero4;
        float4 gu = zero4;
        float4 un = zero4;
        if (a.transform_rots) un = normalize4(u);
        if (a.rot_source == HSR_PREP_ROT_PARAMS) {
            gu = normalize_adjoint(u, gr);
        } else {
            const float4 tr = a.transform_rots ? quat_mult(ps.q, un) : u;
            const float4 t4 = normalize_adjoint(tr, gr);
            gtr.x += t4.x; gtr.y += t4.y; gtr.z += t4.z; gtr.w += t4.w;
        }
        if (a.transform_rots) {
            const float gw = gtr.x, gx = gtr.y, gy = gtr.z, gz = gtr.w;
            sums[12] += gw * un.x + gx * un.y + gy * un.z + gz * un.w;
            sums[13] += -gw * un.y + gx * un.x - gy * un.w + gz * un.z;
            sums[14] += -gw * un.z + gx * un.w + gy * un.x - gz * un.y;
            sums[15] += -gw * un.w - gx * un.z + gy * un.y + gz * un.x;
            const float w1 = ps.q[0], x1 = ps.q[1], y1 = ps.q[2], z1 = ps.q[3];
            const float4 gun = make_float4(gw * w1 + gx * x1 + gy * y1 + gz * z1, -gw * x1 + gx * w1 + gy * z1 - gz * y1,
                                           -gw * y1 - gx * z1 + gy * w1 + gz * x1, -gw * z1 + gx * y1 - gy * x1 + gz * w1);
            const float4 t4 = normalize_adjoint(u, gun);
            gu.x += t4.x; gu.y += t4.y; gu.z += t4.z; gu.w += t4.w;
        } else {
            gu.x += gtr.x; gu.y += gtr.y; gu.z += gtr.z; gu.w += gtr.w;
        }
        if (d_unnorm) reinterpret_cast<float4*>(d_unnorm)[p] = gu;
        if (d_logit) {
            const float s = 1.0f / (1.0f + expf(-a.logit_opacities[p]));
            d_logit[p] = (g_opac ? g_opac[p] : 0.0f) * s * (1.0f - s);
        }
        if (d_log_scales) {
            if (a.S == 1) {
                const float e = expf(a.log_scales[p]);
                d_log_scales[p] = g_scales ? (g_scales[3 * p] * e + g_scales[3 * p + 1] * e) + g_scales[3 * p + 2] * e : 0.0f;
            } else {
#pragma unroll
                for (int i = 0; i < 3; i++) d_log_scales[3 * p + i] = g_scales ? g_scales[3 * p + i] * expf(a.log_scales[3 * p + i]) : 0.0f;
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < PREP_SUMS; k++) {
        const float v = wave_sum(sums[k]);
        if (lane == 0) s_part[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < PREP_SUMS) {
        const int k = threadIdx.x;
        partials[(size_t)blockIdx.x * PREP_SUMS + k] = ((s_part[0][k] + s_part[1][k]) + s_part[2][k]) + s_part[3][k];
    }
}

// One block: sums the per-block partials in a fixed order (double), then the 3x3 -> quaternion adjoint and the two
// normalisation adjoints (build_rotation's and F.normalize's), and writes the 4 + 3 pose gradients.
// param_shaped: d_cam_rot / d_cam_tran are shaped like the parameters ([4][num_frames] / [3][num_frames]): every column but time_idx is
// zeroed here and the gradient lands in column time_idx — what the caller would otherwise do with two fills and two strided copies.
__global__ __launch_bounds__(PREP_FINISH_BLOCK) void frame_prep_finish_kernel(PrepArgs a, const float* __restrict__ partials, int nblocks,
                                                                      float* __restrict__ d_cam_rot, float* __restrict__ d_cam_tran,
                                                                      int param_shaped)
{
    const int cs = param_shaped ? a.num_frames : 1;          // stride between the components of the pose gradient
    const int c0 = param_shaped ? a.time_idx : 0;            // ... and where component 0 sits
    if (param_shaped) {
        if (d_cam_rot) for (int i = threadIdx.x; i < 4 * a.num_frames; i += PREP_FINISH_BLOCK) d_cam_rot[i] = 0.f;
        if (d_cam_tran) for (int i = threadIdx.x; i < 3 * a.num_frames; i += PREP_FINISH_BLOCK) d_cam_tran[i] = 0.f;
    }
    constexpr int ROWS = PREP_FINISH_BLOCK / PREP_SUMS;
    __shared__ double s_acc[ROWS][PREP_SUMS];
    const int k = threadIdx.x % PREP_SUMS, j = threadIdx.x / PREP_SUMS;
    double acc = 0.0;
    for (int b = j; b < nblocks; b += ROWS) acc += (double)partials[(size_t)b * PREP_SUMS + k];
    s_acc[j][k] = acc;
    __syncthreads();
    __shared__ double s_tot[PREP_SUMS];
    if (threadIdx.x < PREP_SUMS) {
        double v = 0.0;
        for (int r = 0; r < ROWS; r++) v += s_acc[r][threadIdx.x];
        s_tot[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double S[PREP_SUMS];
    for (int c = 0; c < PREP_SUMS; c++) S[c] = s_tot[c];
    const Pose ps = load_pose(a.cam_unnorm_rots, a.cam_trans, a.num_frames, a.time_idx);
    if (d_cam_tran) { d_cam_tran[c0] = (float)S[0]; d_cam_tran[c0 + cs] = (float)S[1]; d_cam_tran[c0 + 2 * cs] = (float)S[2]; }
    if (!d_cam_rot) return;
    const double r = ps.qq[0], x = ps.qq[1], y = ps.qq[2], z = ps.qq[3];
    const double* M = S + 3;  // M[3*i + j]
    double dq[4];
    dq[0] = 2.0 * (-z * M[1] + y * M[2] + z * M[3] - x * M[5] - y * M[6] + x * M[7]);
    dq[1] = 2.0 * (y * M[1] + z * M[2] + y * M[3] - 2.0 * x * M[4] - r * M[5] + z * M[6] + r * M[7] - 2.0 * x * M[8]);
    dq[2] = 2.0 * (-2.0 * y * M[0] + x * M[1] + r * M[2] + x * M[3] + z * M[5] - r * M[6] + z * M[7] - 2.0 * y * M[8]);
    dq[3] = 2.0 * (-2.0 * z * M[0] - r * M[1] + x * M[2] + r * M[3] - 2.0 * z * M[4] + y * M[5] + x * M[6] + y * M[7]);
    {   // build_rotation normalises its (already unit) argument again: q -> q / |q|
        const double n = (double)ps.n2;
        double d = 0.0;
        for (int c = 0; c < 4; c++) d += (double)ps.qq[c] * dq[c];
        for (int c = 0; c < 4; c++) dq[c] = (dq[c] - (double)ps.qq[c] * d) / n;
    }
    if (a.transform_rots)
        for (int c = 0; c < 4; c++) dq[c] += S[12 + c];
    {   // F.normalize(cam_unnorm_rots[..., t])
        const double n = fmax((double)ps.nhat, 1e-12);
        double d = 0.0;
        for (int c = 0; c < 4; c++) d += (double)ps.q[c] * dq[c];
        for (int c = 0; c < 4; c++) d_cam_rot[c0 + c * cs] = (float)((dq[c] - (double)ps.q[c] * d) / n);
    }
}

int check_common(int P, int S, int rot_source, const float* means3D, const float* unnorm_rotations, const float* logit_opacities,
                 const float* log_scales, const float* cam_unnorm_rots, const float* cam_trans, int num_frames, int time_idx)
{
    if (P < 0 || (S != 1 && S != 3)) {
        hsr_set_error("frame_prep: invalid sizes P=%d S=%d (log_scales must be [P,1] or [P,3])", P, S);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (rot_source != HSR_PREP_ROT_PARAMS && rot_source != HSR_PREP_ROT_TRANSFORMED) {
        hsr_set_error("frame_prep: rot_source must be HSR_PREP_ROT_PARAMS or HSR_PREP_ROT_TRANSFORMED");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!cam_unnorm_rots || !cam_trans || num_frames < 1 || time_idx < 0 || time_idx >= num_frames) {
        hsr_set_error("frame_prep: camera pose arrays missing or time_idx=%d outside [0, %d)", time_idx, num_frames);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (P > 0 && (!means3D || !unnorm_rotations || !logit_opacities || !log_scales)) {
        hsr_set_error("frame_prep: means3D, unnorm_rotations, logit_opacities and log_scales are required");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    return HSR_OK;
}

}  // namespace

extern "C" size_t hsr_frame_prep_scratch_bytes(int P)
{
    const size_t per_block = (size_t)PREP_BLOCK * PREP_BWD_ITEMS;
    const size_t nblocks = P > 0 ? ((size_t)P + per_block - 1) / per_block : 0;
    return (nblocks + 1) * PREP_SUMS * sizeof(float);
}

extern "C" int hsr_frame_prep_forward(int P, int S, int transform_rots, int rot_source, const float* means3D,
                                      const float* unnorm_rotations, const float* logit_opacities, const float* log_scales,
                                      const float* cam_unnorm_rots, const float* cam_trans, int num_frames, int time_idx,
                                      const float* w2c, float* out_means3D, float* out_unnorm_rot, float* out_rotations,
                                      float* out_opacities, float* out_scales, float* out_depth_sil, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_common(P, S, rot_source, means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots, cam_trans,
                          num_frames, time_idx);
    if (rc != HSR_OK) return rc;
    if ((out_depth_sil != nullptr) != (w2c != nullptr)) {
        hsr_set_error("frame_prep: w2c and out_depth_sil must be given together");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return HSR_OK;
    if (!out_means3D || !out_rotations || !out_opacities || !out_scales) {
        hsr_set_error("frame_prep: an output pointer is NULL");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    PrepArgs a{P, S, transform_rots != 0, rot_source, num_frames, time_idx, means3D, unnorm_rotations, logit_opacities, log_scales,
               cam_unnorm_rots, cam_trans, w2c};
    frame_prep_forward_kernel<<<(P + PREP_BLOCK - 1) / PREP_BLOCK, PREP_BLOCK, 0, stream>>>(a, out_means3D, out_unnorm_rot,
                                                                                          out_rotations, out_opacities, out_scales,
                                                                                          out_depth_sil);
    HSR_HIP_CHECK(hipGetLastError());
    return HSR_OK;
}

namespace {
int frame_prep_backward_impl(int P, int S, int transform_rots, int rot_source, const float* means3D,
                             const float* unnorm_rotations, const float* logit_opacities, const float* log_scales,
                             const float* cam_unnorm_rots, const float* cam_trans, int num_frames, int time_idx,
                             const float* w2c, const float* dL_dout_means3D, const float* dL_dout_unnorm_rot,
                             const float* dL_dout_rotations, const float* dL_dout_opacities, const float* dL_dout_scales,
                             const float* dL_dout_depth_sil, float* dL_dmeans3D, float* dL_dunnorm_rotations,
                             float* dL_dlogit_opacities, float* dL_dlog_scales, float* dL_dcam_unnorm_rot,
                             float* dL_dcam_tran, char* scratch, size_t scratch_bytes, void* stream_, int param_shaped)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_common(P, S, rot_source, means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots, cam_trans,
                          num_frames, time_idx);
    if (rc != HSR_OK) return rc;
    if (dL_dout_depth_sil && !w2c) {
        hsr_set_error("frame_prep: dL_dout_depth_sil needs w2c");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!scratch || scratch_bytes < hsr_frame_prep_scratch_bytes(P)) {
        hsr_set_error("frame_prep: scratch too small: %zu bytes needed", hsr_frame_prep_scratch_bytes(P));
        return HSR_ERR_BUFFER_TOO_SMALL;
    }
    PrepArgs a{P, S, transform_rots != 0, rot_source, num_frames, time_idx, means3D, unnorm_rotations, logit_opacities, log_scales,
               cam_unnorm_rots, cam_trans, w2c};
    float* partials = reinterpret_cast<float*>(scratch);
    const int per_block = PREP_BLOCK * PREP_BWD_ITEMS;
    const int nblocks = P > 0 ? (int)(((size_t)P + per_block - 1) / per_block) : 0;
    if (nblocks > 0) {
        frame_prep_backward_kernel<<<nblocks, PREP_BLOCK, 0, stream>>>(a, dL_dout_means3D, dL_dout_unnorm_rot, dL_dout_rotations,
                                                                       dL_dout_opacities, dL_dout_scales, dL_dout_depth_sil,
                                                                       dL_dmeans3D, dL_dunnorm_rotations, dL_dlogit_opacities,
                                                                       dL_dlog_scales, partials);
        HSR_HIP_CHECK(hipGetLastError());
    }
    if (dL_dcam_unnorm_rot || dL_dcam_tran) {
        frame_prep_finish_kernel<<<1, PREP_FINISH_BLOCK, 0, stream>>>(a, partials, nblocks, dL_dcam_unnorm_rot, dL_dcam_tran, param_shaped);
        HSR_HIP_CHECK(hipGetLastError());
    }
    return HSR_OK;
}
}  // namespace

extern "C" int hsr_frame_prep_backward(int P, int S, int transform_rots, int rot_source, const float* means3D,
                                       const float* unnorm_rotations, const float* logit_opacities, const float* log_scales,
                                       const float* cam_unnorm_rots, const float* cam_trans, int num_frames, int time_idx,
                                       const float* w2c, const float* dL_dout_means3D, const float* dL_dout_unnorm_rot,
                                       const float* dL_dout_rotations, const float* dL_dout_opacities, const float* dL_dout_scales,
                                       const float* dL_dout_depth_sil, float* dL_dmeans3D, float* dL_dunnorm_rotations,
                                       float* dL_dlogit_opacities, float* dL_dlog_scales, float* dL_dcam_unnorm_rot,
                                       float* dL_dcam_tran, char* scratch, size_t scratch_bytes, void* stream_)
{
    return frame_prep_backward_impl(P, S, transform_rots, rot_source, means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots,
                                    cam_trans, num_frames, time_idx, w2c, dL_dout_means3D, dL_dout_unnorm_rot, dL_dout_rotations,
                                    dL_dout_opacities, dL_dout_scales, dL_dout_depth_sil, dL_dmeans3D, dL_dunnorm_rotations,
                                    dL_dlogit_opacities, dL_dlog_scales, dL_dcam_unnorm_rot, dL_dcam_tran, scratch, scratch_bytes, stream_, 0);
}

extern "C" int hsr_frame_prep_backward_params(int P, int S, int transform_rots, int rot_source, const float* means3D,
                                              const float* unnorm_rotations, const float* logit_opacities, const float* log_scales,
                                              const float* cam_unnorm_rots, const float* cam_trans, int num_frames, int time_idx,
                                              const float* w2c, const float* dL_dout_means3D, const float* dL_dout_unnorm_rot,
                                              const float* dL_dout_rotations, const float* dL_dout_opacities, const float* dL_dout_scales,
                                              const float* dL_dout_depth_sil, float* dL_dmeans3D, float* dL_dunnorm_rotations,
                                              float* dL_dlogit_opacities, float* dL_dlog_scales, float* dL_dcam_unnorm_rots,
                                              float* dL_dcam_trans, char* scratch, size_t scratch_bytes, void* stream_)
{
    return frame_prep_backward_impl(P, S, transform_rots, rot_source, means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots,
                                    cam_trans, num_frames, time_idx, w2c, dL_dout_means3D, dL_dout_unnorm_rot, dL_dout_rotations,
                                    dL_dout_opacities, dL_dout_scales, dL_dout_depth_sil, dL_dmeans3D, dL_dunnorm_rotations,
                                    dL_dlogit_opacities, dL_dlog_scales, dL_dcam_unnorm_rots, dL_dcam_trans, scratch, scratch_bytes, stream_, 1);
}
