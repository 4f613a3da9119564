"""Camera tensors as the reference's `setup_camera` builds them (utils/recon_helpers.py:4-28).

`viewmatrix` is w2c transposed (column-major w2c in memory), `projmatrix` is
viewmatrix @ opengl_proj^T, tanfov = w/(2fx), h/(2fy).  Built with torch on CPU in fp32 with the
same operation order as the reference so the bits of the matrices match what hierslam.py feeds
the rasterizer; moved to `device` afterwards.
"""
import numpy as np
import torch


def setup_camera_tensors(w, h, k, w2c, near=0.01, far=100, device="cpu"):
    fx, fy, cx, cy = k[0][0], k[1][1], k[0][2], k[1][2]
    w2c = torch.tensor(np.asarray(w2c)).float()
    cam_center = torch.inverse(w2c)[:3, 3]
    w2c = w2c.unsqueeze(0).transpose(1, 2)
    opengl_proj = torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                                [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                                [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                                [0.0, 0.0, 1.0, 0.0]]).float().unsqueeze(0).transpose(1, 2)
    full_proj = w2c.bmm(opengl_proj)
    return dict(
        image_height=int(h), image_width=int(w),
        tanfovx=w / (2 * fx), tanfovy=h / (2 * fy),
        bg=torch.tensor([0, 0, 0], dtype=torch.float32, device=device),
        scale_modifier=1.0,
        viewmatrix=w2c.contiguous().to(device), projmatrix=full_proj.contiguous().to(device),
        sh_degree=0, campos=cam_center.contiguous().to(device),
        prefiltered=False, debug=False,
    )


def setup_camera(w, h, k, w2c, near=0.01, far=100, device="cuda"):
    """Drop-in for utils/recon_helpers.py:setup_camera -> GaussianRasterizationSettings."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    return Camera(**setup_camera_tensors(w, h, k, w2c, near, far, device))


def replica_intrinsics(w=1200, h=680):
    """configs/data/replica_semantic.yaml:5-8 (fx=fy=600, cx=599.5, cy=339.5 at 1200x680), scaled."""
    sx, sy = w / 1200.0, h / 680.0
    return np.array([[600.0 * sx, 0, 599.5 * sx], [0, 600.0 * sy, 339.5 * sy], [0, 0, 1]], dtype=np.float64)
