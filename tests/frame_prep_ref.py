"""Independent torch (CPU, float64) expression of the rasterizer-input preparation, differentiated by
torch.autograd: the pin for oracle/frame_prep_oracle.py (the reference's own functions need a CUDA device,
utils/slam_helpers.py:298, so they cannot be run here)."""
import torch
import torch.nn.functional as F


def _rot(q):
    q = q / q.norm()
    r, x, y, z = q
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)]),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)]),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)])])


def _hamilton(a, b):
    """a: [4] (w, x, y, z), b: [P, 4]."""
    aw, av = a[0], a[1:]
    bw, bv = b[:, :1], b[:, 1:]
    w = aw * bw - (bv * av).sum(dim=1, keepdim=True)
    v = aw * bv + bw * av + torch.cross(av.expand_as(bv), bv, dim=1)
    return torch.cat([w, v], dim=1)


def prep(means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots, cam_trans, time_idx, transform_rots,
         rot_source, w2c=None):
    q = F.normalize(cam_unnorm_rots[0, :, time_idx], dim=0)
    t = cam_trans[0, :, time_idx]
    out = {"means3D": means3D @ _rot(q).T + t}
    tr = _hamilton(q, F.normalize(unnorm_rotations, dim=1)) if transform_rots else unnorm_rotations
    out["unnorm_rotations"] = tr
    out["rotations"] = F.normalize(unnorm_rotations if rot_source == 0 else tr, dim=1)
    out["opacities"] = torch.sigmoid(logit_opacities)
    out["scales"] = torch.exp(log_scales.expand(-1, 3) if log_scales.shape[1] == 1 else log_scales)
    if w2c is not None:
        z = out["means3D"] @ w2c[2, :3] + w2c[2, 3]
        out["depth_sil"] = torch.stack([z, torch.ones_like(z), z * z], dim=1)
    return out
