"""CPU suite for the on-disk map formats: params.npz schema round trip and the PLY byte layout."""
import os

import numpy as np
import pytest
import torch

from hsr_utils import map_io as M


def _params(P=50, K=26, F=7, S=1):
    g = np.random.default_rng(0)
    t = lambda *s: torch.tensor(g.normal(size=s).astype(np.float32))
    return {"means3D": t(P, 3), "rgb_colors": torch.rand(P, 3), "unnorm_rotations": t(P, 4), "logit_opacities": t(P, 1),
            "log_scales": t(P, S), "semantic": t(P, K), "cam_unnorm_rots": t(1, 4, F), "cam_trans": t(1, 3, F)}


def test_params_npz_round_trip_and_schema(tmp_path):
    p = _params()
    p["means3D"].requires_grad_(True)
    variables = {"timestep": torch.arange(50).float()}
    full = M.finalize_params(p, variables, torch.eye(3), torch.eye(4), 1200, 680, [torch.eye(4)] * 7, [0, 3, 6])
    path = M.save_params(full, str(tmp_path))
    assert os.path.basename(path) == "params.npz"
    z = dict(np.load(path, allow_pickle=True))          # what the reference's consumers call
    assert set(M.GAUSSIAN_KEYS + M.CAMERA_KEYS + M.EXTRA_KEYS) <= set(z)
    assert z["gt_w2c_all_frames"].shape == (7, 4, 4) and list(z["keyframe_time_indices"]) == [0, 3, 6]
    assert int(z["org_width"]) == 1200 and z["means3D"].dtype == np.float32
    back = M.load_params(path)
    np.testing.assert_array_equal(back["semantic"], p["semantic"].numpy())
    assert M.check_schema(back) == (50, 7)
    assert os.path.basename(M.save_params_ckpt(p, str(tmp_path), 12)) == "params12.npz"
    bad = dict(back); bad["logit_opacities"] = back["logit_opacities"][:, 0]
    with pytest.raises(ValueError, match="logit_opacities"):
        M.check_schema(bad)


def test_ply_layout_matches_the_reference_attribute_order(tmp_path):
    p = {k: v.numpy() for k, v in _params(P=9).items()}
    path = M.save_ply(str(tmp_path / "a.ply"), p["means3D"], p["log_scales"], p["unnorm_rotations"], p["rgb_colors"], p["logit_opacities"])
    raw = open(path, "rb").read()
    header = raw[:raw.index(b"end_header\n") + 11].decode().split("\n")
    assert header[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 9"]
    names = [l.split()[2] for l in header if l.startswith("property")]
    assert names == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
                     "rot_0", "rot_1", "rot_2", "rot_3"]
    assert all(l.split()[1] == "float" for l in header if l.startswith("property"))
    assert len(raw) == len("\n".join(header)) + 9 * 17 * 4
    el = M.read_ply(path)
    np.testing.assert_allclose(el["x"], p["means3D"][:, 0])
    np.testing.assert_allclose(el["f_dc_1"], (p["rgb_colors"][:, 1] - 0.5) / M.C0, rtol=1e-6)
    np.testing.assert_allclose(el["scale_2"], p["log_scales"][:, 0])       # isotropic scale tiled to three columns
    assert not el["nx"].any()


def test_semantic_ply_and_tree_labels(tmp_path):
    p = {k: v.numpy() for k, v in _params(P=11, K=12).items()}
    labels = M.transfer_tree_label(p["semantic"], [2, 4, 6, 40])            # three levels + the leaf-class count
    assert labels.shape == (3, 11) and labels[1].max() < 4
    np.testing.assert_array_equal(labels[2], np.argmax(p["semantic"][:, 6:12], axis=1))
    colors = (np.arange(33).reshape(11, 3) * 7 % 256).astype(np.uint8)
    path = M.save_ply_semantic(str(tmp_path / "s.ply"), p["means3D"], p["log_scales"], p["unnorm_rotations"], colors, p["logit_opacities"])
    el = M.read_ply(path)
    assert el.dtype["red"] == np.uint8 and np.array_equal(np.stack([el["red"], el["green"], el["blue"]], 1), colors)
    assert el.dtype.itemsize == 14 * 4 + 3
