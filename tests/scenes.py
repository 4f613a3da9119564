"""Seeded test scenes shared by the CPU and GPU suites (inputs only; expectations come from the oracle)."""
import math

import numpy as np
import torch

from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
from hsr_utils.synthetic import make_scene, make_upstream_grads


def tilted_w2c(angle=0.2, t=(0.1, -0.05, 0.2)):
    w2c = np.eye(4)
    w2c[:3, :3] = np.array([[math.cos(angle), 0, math.sin(angle)], [0, 1, 0], [-math.sin(angle), 0, math.cos(angle)]])
    w2c[:3, 3] = t
    return w2c


def build(W, H, P, K, seed=0, kind="aniso", scale_mult=2.0, tilt=True, bg=(0.0, 0.0, 0.0), behind_frac=0.0, grad_seed=1,
          grad_scale=None):
    k = replica_intrinsics(W, H)
    w2c = tilted_w2c() if tilt else np.eye(4)
    cam = setup_camera_tensors(W, H, k, w2c)
    cam["bg"] = torch.tensor(bg, dtype=torch.float32)
    sc = make_scene(P, W, H, K, k, seed=seed, kind=kind, scale_mult=scale_mult, w2c=w2c if tilt else None,
                    behind_frac=behind_frac)
    up = make_upstream_grads(W, H, K, seed=grad_seed)
    s = float(W * H) if grad_scale is None else grad_scale
    up = {n: v * s for n, v in up.items()}  # O(1) upstream grads: errors then read as relative
    return cam, sc, up


def cov3d_from_scene(sc, mod=1.0):
    """world covariance [P,6] from scales/rotations (float64 maths, rounded to fp32) for the cov3D_precomp path"""
    s = sc["scales"].double() * mod
    q = sc["rotations"].double()
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    Sig = R @ torch.diag_embed(s * s) @ R.transpose(1, 2)
    return torch.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], 1).float().contiguous()


def random_sh(P, M, seed=7):
    g = torch.Generator(device="cpu").manual_seed(seed)
    sh = torch.randn(P, M, 3, generator=g) * 0.3
    sh[:, 0, :] += 0.8
    return sh.float().contiguous()
