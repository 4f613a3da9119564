"""Deterministic synthetic scenes (SURVEY.md §8d / BASELINE.md §2).

"slam": Gaussians as scripts/hierslam.py initialises them from an RGB-D frame (:361-389):
mean = unprojection of a uniform pixel at depth U(0.5, 6) m, isotropic scale z/f * lognormal,
identity rotation, opacity sigmoid(N(0,1.5^2)), colour U(0,1)^3, semantics U(0,1)^K.
"aniso": random unit quaternions and per-axis log-normal scales (exercises the covariance paths).
Generated with a CPU torch.Generator so the same seed gives the same bits everywhere.
"""
import numpy as np
import torch


def make_scene(P, W, H, K, kmat, seed=0, kind="slam", w2c=None, scale_mult=1.0, behind_frac=0.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    fx, fy, cx, cy = float(kmat[0][0]), float(kmat[1][1]), float(kmat[0][2]), float(kmat[1][2])
    u = torch.rand(P, generator=g) * (W + 32) - 16
    v = torch.rand(P, generator=g) * (H + 32) - 16
    z = torch.rand(P, generator=g) * 5.5 + 0.5
    if behind_frac > 0:  # some points behind / too near the camera: exercises the z<=0.2 cull
        nb = int(P * behind_frac)
        z[:nb] = torch.rand(nb, generator=g) * 0.6 - 0.3
    x = (u - cx) / fx * z
    y = (v - cy) / fy * z
    means_cam = torch.stack([x, y, z], dim=1)
    if w2c is not None:  # place in world so that w2c maps them back in front of the camera
        c2w = torch.inverse(torch.tensor(np.asarray(w2c)).float())
        means = means_cam @ c2w[:3, :3].T + c2w[:3, 3]
    else:
        means = means_cam
    f = 0.5 * (fx + fy)
    if kind == "slam":
        s = (z.abs().clamp_min(0.05) / f) * torch.exp(torch.randn(P, generator=g) * 0.35) * scale_mult
        scales = s[:, None].repeat(1, 3)
        rots = torch.zeros(P, 4); rots[:, 0] = 1.0
    elif kind == "aniso":
        s = (z.abs().clamp_min(0.05) / f)[:, None] * torch.exp(torch.randn(P, 3, generator=g) * 0.6) * scale_mult * 1.5
        scales = s
        q = torch.randn(P, 4, generator=g)
        rots = q / q.norm(dim=1, keepdim=True)
    else:
        raise ValueError(kind)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5)
    colors = torch.rand(P, 3, generator=g)
    sem = torch.rand(P, max(K, 1), generator=g)[:, :K] if K > 0 else torch.zeros(P, 0)
    return dict(means3D=means.float().contiguous(), scales=scales.float().contiguous(),
                rotations=rots.float().contiguous(), opacities=opac.float().contiguous(),
                colors_precomp=colors.float().contiguous(), semantics_precomp=sem.float().contiguous())


def make_upstream_grads(W, H, K, seed=1):
    """Upstream grads N(0,1)/N for colour[3], semantic[K], depth, median depth, final opacity."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    N = W * H
    mk = lambda c: (torch.randn(c, H, W, generator=g) / N).float().contiguous()
    return dict(color=mk(3), semantic=mk(K) if K > 0 else torch.zeros(0, H, W), depth=mk(1), median=mk(1), opacity=mk(1))
