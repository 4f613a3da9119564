"""CPU restatement (numpy) of the reference's rasterizer-input preparation — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(hier-slam_amd/csrc/hsr_frame_prep.hip behind include/hsr_frame_prep.h) never does.

PARITY UNPINNED: the reference functions restated here allocate on 'cuda' unconditionally
(utils/slam_helpers.py:298, :319, :234; utils/slam_external.py:28), so they cannot run in a GPU-less container,
and the reference ships no fixtures for them.  What pins this file instead is tests/test_frame_prep.py: an
independently written torch (CPU, float64) expression of the same maths differentiated by torch.autograd.

Follows, line by line in meaning (not in code):
  transform_to_frame                      utils/slam_helpers.py:278-330
  build_rotation                          utils/slam_external.py:25-42
  quat_mult                               utils/slam_helpers.py:21-28
  transformed_params2rendervar            utils/slam_helpers.py:124-139
  transformed_params2rendervar_semantic   utils/slam_helpers.py:195-219
  get_depth_and_silhouette                utils/slam_helpers.py:222-239
Forward runs in float32 (the reference's dtype); backward is the analytic adjoint evaluated in float64.
"""
import numpy as np

ROT_PARAMS = 0       # rotations = F.normalize(params['unnorm_rotations'])            slam_helpers.py:212
ROT_TRANSFORMED = 1  # rotations = F.normalize(transformed_gaussians['unnorm_rotations'])   slam_helpers.py:134, :270
EPS = 1e-12          # torch.nn.functional.normalize default eps


def _normalize_rows(x):
    n = np.sqrt((x * x).sum(axis=-1, keepdims=True, dtype=x.dtype))
    return x / np.maximum(n, x.dtype.type(EPS)), n


def _rotation(q):
    """build_rotation for ONE quaternion (r, x, y, z): normalise again, then the 3x3 (slam_external.py:25-42)."""
    t = q.dtype.type
    n = np.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    r, x, y, z = (q / n)
    R = np.empty((3, 3), dtype=q.dtype)
    R[0] = (t(1) - t(2) * (y * y + z * z), t(2) * (x * y - r * z), t(2) * (x * z + r * y))
    R[1] = (t(2) * (x * y + r * z), t(1) - t(2) * (x * x + z * z), t(2) * (y * z - r * x))
    R[2] = (t(2) * (x * z - r * y), t(2) * (y * z + r * x), t(1) - t(2) * (x * x + y * y))
    return R, n


def _quat_mult(q1, q2):
    """q1: [4], q2: [P,4] (slam_helpers.py:21-28)."""
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2[:, 0], q2[:, 1], q2[:, 2], q2[:, 3]
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                     w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=1)


def forward(means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots, cam_trans, time_idx,
            transform_rots=None, rot_source=ROT_PARAMS, w2c=None, dtype=np.float32):
    """Returns dict(means3D, unnorm_rotations, rotations, opacities, scales[, depth_sil])."""
    f = lambda a: np.asarray(a, dtype=dtype)
    x, u, lo, ls = f(means3D), f(unnorm_rotations), f(logit_opacities), f(log_scales)
    P, S = x.shape[0], ls.shape[1]
    if transform_rots is None:
        transform_rots = S != 1                                     # slam_helpers.py:302-306
    qhat = f(cam_unnorm_rots)[0, :, time_idx]
    q, _ = _normalize_rows(qhat[None]); q = q[0]                    # :293
    t = f(cam_trans)[0, :, time_idx]                                # :294
    R, _ = _rotation(q)                                             # :299
    one = dtype(1)
    out = {}
    # (rel_w2c @ pts4.T).T[:, :3]  (:318-321): row . [x, y, z, 1], accumulated left to right
    out["means3D"] = ((x[:, 0:1] * R[None, :, 0] + x[:, 1:2] * R[None, :, 1]) + x[:, 2:3] * R[None, :, 2]) + t[None, :] * one
    if transform_rots:
        un, _ = _normalize_rows(u)                                  # :325
        tr = _quat_mult(q, un)                                      # :326
    else:
        tr = u                                                      # :329
    out["unnorm_rotations"] = tr
    out["rotations"] = _normalize_rows(u if rot_source == ROT_PARAMS else tr)[0]
    out["opacities"] = one / (one + np.exp(-lo))                    # torch.sigmoid
    out["scales"] = np.exp(np.tile(ls, (1, 3)) if S == 1 else ls)   # :127-130, :215
    if w2c is not None:
        m = f(w2c)
        z = ((out["means3D"][:, 0] * m[2, 0] + out["means3D"][:, 1] * m[2, 1]) + out["means3D"][:, 2] * m[2, 2]) + m[2, 3]
        out["depth_sil"] = np.stack([z, np.ones_like(z), z * z], axis=1)   # :234-237
    return out


def _normalize_adjoint(x, g):
    """adjoint of y = x / max(|x|, eps) along the last axis (|x| > eps assumed, as in every test)."""
    n = np.sqrt((x * x).sum(axis=-1, keepdims=True))
    y = x / n
    return (g - y * (y * g).sum(axis=-1, keepdims=True)) / n


def backward(means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots, cam_trans, time_idx, grads,
             transform_rots=None, rot_source=ROT_PARAMS, w2c=None):
    """grads: dict with any of means3D, unnorm_rotations, rotations, opacities, scales, depth_sil (missing = zero).
    Returns dict(means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rot[4], cam_tran[3]), float64."""
    d = np.float64
    f = lambda a: np.asarray(a, dtype=d)
    x, u, lo, ls = f(means3D), f(unnorm_rotations), f(logit_opacities), f(log_scales)
    P, S = x.shape[0], ls.shape[1]
    if transform_rots is None:
        transform_rots = S != 1
    g = lambda k, shape: f(grads[k]) if grads.get(k) is not None else np.zeros(shape, d)
    g_m, g_tr, g_rot = g("means3D", (P, 3)).copy(), g("unnorm_rotations", (P, 4)).copy(), g("rotations", (P, 4))
    g_op, g_sc = g("opacities", (P, 1)), g("scales", (P, 3))
    qhat = f(cam_unnorm_rots)[0, :, time_idx]
    q = qhat / max(np.sqrt((qhat * qhat).sum()), EPS)
    tvec = f(cam_trans)[0, :, time_idx]
    R, n2 = _rotation(q)
    qq = q / n2
    if w2c is not None and grads.get("depth_sil") is not None:
        m = f(w2c)
        xc = x @ R.T + tvec
        z = xc @ m[2, :3] + m[2, 3]
        gd = f(grads["depth_sil"])
        dz = gd[:, 0] + 2.0 * z * gd[:, 2]
        g_m += dz[:, None] * m[2, :3][None, :]
    out = {}
    out["means3D"] = g_m @ R                                   # R^T g per row
    out["cam_tran"] = g_m.sum(axis=0)
    M = g_m.T @ x                                              # dL/dR[i][j] = sum_p g_i x_j
    r, qx, qy, qz = qq
    dq = 2.0 * np.array([
        -qz * M[0, 1] + qy * M[0, 2] + qz * M[1, 0] - qx * M[1, 2] - qy * M[2, 0] + qx * M[2, 1],
        qy * M[0, 1] + qz * M[0, 2] + qy * M[1, 0] - 2 * qx * M[1, 1] - r * M[1, 2] + qz * M[2, 0] + r * M[2, 1] - 2 * qx * M[2, 2],
        -2 * qy * M[0, 0] + qx * M[0, 1] + r * M[0, 2] + qx * M[1, 0] + qz * M[1, 2] - r * M[2, 0] + qz * M[2, 1] - 2 * qy * M[2, 2],
        -2 * qz * M[0, 0] - r * M[0, 1] + qx * M[0, 2] + r * M[1, 0] - 2 * qz * M[1, 1] + qy * M[1, 2] + qx * M[2, 0] + qy * M[2, 1]])
    dq = _normalize_adjoint(q[None], dq[None])[0]              # build_rotation's own normalisation
    # rotations
    g_u = np.zeros((P, 4), d)
    if rot_source == ROT_PARAMS:
        g_u += _normalize_adjoint(u, g_rot)
    else:
        if transform_rots:
            un = u / np.sqrt((u * u).sum(axis=1, keepdims=True))
            tr = _quat_mult(q, un)
        else:
            tr = u
        g_tr += _normalize_adjoint(tr, g_rot)
    if transform_rots:
        un = u / np.sqrt((u * u).sum(axis=1, keepdims=True))
        gw, gx, gy, gz = g_tr[:, 0], g_tr[:, 1], g_tr[:, 2], g_tr[:, 3]
        w2_, x2, y2, z2 = un[:, 0], un[:, 1], un[:, 2], un[:, 3]
        dq += np.array([(gw * w2_ + gx * x2 + gy * y2 + gz * z2).sum(), (-gw * x2 + gx * w2_ - gy * z2 + gz * y2).sum(),
                        (-gw * y2 + gx * z2 + gy * w2_ - gz * x2).sum(), (-gw * z2 - gx * y2 + gy * x2 + gz * w2_).sum()])
        w1, x1, y1, z1 = q
        g_un = np.stack([gw * w1 + gx * x1 + gy * y1 + gz * z1, -gw * x1 + gx * w1 + gy * z1 - gz * y1,
                         -gw * y1 - gx * z1 + gy * w1 + gz * x1, -gw * z1 + gx * y1 - gy * x1 + gz * w1], axis=1)
        g_u += _normalize_adjoint(u, g_un)
    else:
        g_u += g_tr
    out["unnorm_rotations"] = g_u
    out["cam_unnorm_rot"] = _normalize_adjoint(qhat[None], dq[None])[0]   # F.normalize at :293
    s = 1.0 / (1.0 + np.exp(-lo))
    out["logit_opacities"] = g_op * s * (1.0 - s)
    e = np.exp(np.tile(ls, (1, 3)) if S == 1 else ls)
    ge = g_sc * e
    out["log_scales"] = ge.sum(axis=1, keepdims=True) if S == 1 else ge
    return out
