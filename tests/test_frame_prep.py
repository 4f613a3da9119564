"""CPU suite for the rasterizer-input preparation row: the numpy oracle against an independent
float64 torch.autograd expression, on every variant the reference has."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import frame_prep_oracle as O  # noqa: E402
from frame_prep_ref import prep as ref_prep  # noqa: E402


def make_inputs(P, S, frames=5, seed=0):
    g = np.random.default_rng(seed)
    return dict(means3D=g.normal(0, 2, (P, 3)).astype(np.float32), unnorm_rotations=g.normal(0, 1, (P, 4)).astype(np.float32),
                logit_opacities=g.normal(0, 1.5, (P, 1)).astype(np.float32), log_scales=g.normal(-4, 0.5, (P, S)).astype(np.float32),
                cam_unnorm_rots=(g.normal(0, 1, (1, 4, frames)) * 1.7).astype(np.float32),
                cam_trans=g.normal(0, 0.5, (1, 3, frames)).astype(np.float32))


def make_grads(P, with_sil, seed=1):
    g = np.random.default_rng(seed)
    d = dict(means3D=g.normal(0, 1, (P, 3)), unnorm_rotations=g.normal(0, 1, (P, 4)), rotations=g.normal(0, 1, (P, 4)),
             opacities=g.normal(0, 1, (P, 1)), scales=g.normal(0, 1, (P, 3)))
    if with_sil:
        d["depth_sil"] = g.normal(0, 1, (P, 3))
    return {k: v.astype(np.float32) for k, v in d.items()}


VARIANTS = [(1, False, O.ROT_PARAMS, False), (1, False, O.ROT_TRANSFORMED, True), (3, True, O.ROT_TRANSFORMED, False),
            (3, True, O.ROT_TRANSFORMED, True), (3, True, O.ROT_PARAMS, False), (3, False, O.ROT_TRANSFORMED, False)]


@pytest.mark.parametrize("S,transform_rots,rot_source,with_sil", VARIANTS)
def test_oracle_matches_autograd(S, transform_rots, rot_source, with_sil):
    P, tidx = 257, 3
    inp = make_inputs(P, S)
    grads = make_grads(P, with_sil)
    w2c = None
    if with_sil:
        w2c = np.eye(4, dtype=np.float32); w2c[:3, :3] = O._rotation(np.array([0.9, 0.1, -0.3, 0.2], np.float32))[0]; w2c[:3, 3] = (0.1, -0.2, 0.3)
    fo = O.forward(**inp, time_idx=tidx, transform_rots=transform_rots, rot_source=rot_source, w2c=w2c)
    bo = O.backward(**inp, time_idx=tidx, grads=grads, transform_rots=transform_rots, rot_source=rot_source, w2c=w2c)
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in inp.items()}
    fr = ref_prep(**t, time_idx=tidx, transform_rots=transform_rots, rot_source=rot_source,
                  w2c=None if w2c is None else torch.tensor(w2c, dtype=torch.float64))
    for k, v in fr.items():
        np.testing.assert_allclose(fo[k], v.detach().numpy(), rtol=2e-6, atol=2e-6, err_msg=k)
    loss = sum((fr[k] * torch.tensor(grads[k], dtype=torch.float64)).sum() for k in grads)
    loss.backward()
    for k, tk in (("means3D", "means3D"), ("unnorm_rotations", "unnorm_rotations"), ("logit_opacities", "logit_opacities"),
                  ("log_scales", "log_scales")):
        np.testing.assert_allclose(bo[k], t[tk].grad.numpy(), rtol=1e-9, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(bo["cam_unnorm_rot"], t["cam_unnorm_rots"].grad.numpy()[0, :, tidx], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(bo["cam_tran"], t["cam_trans"].grad.numpy()[0, :, tidx], rtol=1e-9, atol=1e-9)
    other = np.delete(t["cam_unnorm_rots"].grad.numpy()[0], tidx, axis=1)
    assert not other.any()


def test_oracle_empty_and_single():
    for P in (0, 1):
        inp = make_inputs(P, 1)
        fo = O.forward(**inp, time_idx=0)
        assert fo["means3D"].shape == (P, 3) and fo["scales"].shape == (P, 3)
        bo = O.backward(**inp, time_idx=0, grads=make_grads(P, False))
        assert bo["cam_tran"].shape == (3,) and np.isfinite(bo["cam_unnorm_rot"]).all()
