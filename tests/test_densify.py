"""CPU suite: oracle/densify_oracle.py against the reference's lines written with torch CPU ops (torch.median, boolean-mask
indexing, torch.inverse): scripts/hierslam.py:1271-1278, :1289-1290, :144-194, :1157."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import densify_oracle as DO  # noqa: E402


def make_frame(H, W, seed=0, hole=True):
    g = np.random.default_rng(seed)
    gt = (g.random((H, W)) * 4 + 0.5).astype(np.float32)
    gt[: H // 8] = 0                                              # invalid depth rows
    rd = (gt + g.normal(0, 0.01, (H, W))).astype(np.float32)
    rd[H // 2:, : W // 3] += 2.0                                   # rendered surface far behind the measured one
    sil = np.clip(g.normal(0.98, 0.01, (H, W)), 0, 1).astype(np.float32)
    if hole:
        sil[H // 3: H // 2, W // 2:] = 0.1                         # unseen region
    col = g.random((3, H, W)).astype(np.float32)
    K = np.array([[W * 0.5, 0, W / 2 - 0.5], [0, W * 0.5, H / 2 - 0.5], [0, 0, 1]], np.float32)
    q = np.array([0.98, 0.1, -0.12, 0.05]); q /= np.linalg.norm(q)
    r, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                  [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
    w2c = np.eye(4, dtype=np.float32); w2c[:3, :3] = R; w2c[:3, 3] = (0.2, -0.1, 0.3)
    return sil, rd, gt, col, K, w2c


def torch_reference(sil, rd, gt, col, K, w2c, sil_thres):
    """the reference's expressions on CPU tensors"""
    sil, rd, gt, col, K, w2c = map(torch.tensor, (sil, rd, gt, col, K, w2c))
    depth_error = torch.abs(gt - rd) * (gt > 0)
    mask = (sil < sil_thres) | ((rd > gt) * (depth_error > 50 * depth_error.median()))
    mask = mask.reshape(-1) & (gt > 0).reshape(-1)
    H, W = gt.shape
    xg, yg = torch.meshgrid(torch.arange(W).float(), torch.arange(H).float(), indexing='xy')
    xx, yy = ((xg - K[0][2]) / K[0][0]).reshape(-1), ((yg - K[1][2]) / K[1][1]).reshape(-1)
    z = gt.reshape(-1)
    pts4 = torch.cat((torch.stack((xx * z, yy * z, z), dim=-1), torch.ones(H * W, 1)), dim=1)
    pts = (torch.inverse(w2c) @ pts4.T).T[:, :3]
    msd = (z / ((K[0][0] + K[1][1]) / 2)) ** 2
    cols = torch.permute(col, (1, 2, 0)).reshape(-1, 3)
    return mask.numpy(), float(depth_error.median()), pts[mask].numpy(), cols[mask].numpy(), msd[mask].numpy()


def test_oracle_matches_torch_expressions():
    for seed, (H, W) in enumerate([(48, 64), (37, 53), (16, 16)]):
        sil, rd, gt, col, K, w2c = make_frame(H, W, seed)
        o = DO.non_presence_points(sil, rd, gt, col, K, np.linalg.inv(w2c.astype(np.float64)).astype(np.float32), 0.5)
        mask, med, pts, cols, msd = torch_reference(sil, rd, gt, col, K, w2c, 0.5)
        assert np.array_equal(o["mask"], mask) and o["median"] == np.float32(med)
        assert 0 < mask.sum() < mask.size
        np.testing.assert_allclose(o["means3D"], pts, rtol=2e-5, atol=2e-5)
        np.testing.assert_array_equal(o["rgb"], cols)
        np.testing.assert_allclose(o["mean3_sq_dist"], msd, rtol=1e-6)
        np.testing.assert_allclose(o["log_scales"], np.log(np.sqrt(msd)), rtol=1e-5, atol=1e-6)


def test_oracle_nothing_to_add():
    sil, rd, gt, col, K, w2c = make_frame(32, 40, 5, hole=False)
    rd = gt.copy()
    o = DO.non_presence_points(sil, rd, gt, col, K, np.linalg.inv(w2c), 0.5)
    assert o["mask"].sum() == 0 and o["means3D"].shape == (0, 3)
