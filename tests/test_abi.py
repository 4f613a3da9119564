"""CPU suite: the C-ABI library loads and exports exactly what include/hsr_rasterizer.h declares, the
ctypes glue agrees with the header's parameter lists, and host-only entry points behave (no compute calls:
there is no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hsr_rasterizer.h")
HEADERS = [HEADER, os.path.join(ROOT, "include", "hsr_frame_prep.h"), os.path.join(ROOT, "include", "hsr_losses.h"),
           os.path.join(ROOT, "include", "hsr_densify.h")]


def _prototypes():
    src = "\n".join(open(h).read() for h in HEADERS)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(hsr_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        n = 0 if params in ("", "void") else len([p for p in params.split(",") if p.strip()])
        protos[name] = n
    return protos


def test_library_exports_every_declared_symbol():
    from diff_gaussian_rasterization import _C
    protos = _prototypes()
    assert {"hsr_forward", "hsr_forward_semantic", "hsr_backward", "hsr_backward_semantic", "hsr_mark_visible",
            "hsr_required_geometry_bytes", "hsr_required_image_bytes", "hsr_required_binning_bytes", "hsr_last_error",
            "hsr_version", "hsr_get_state_layout", "hsr_profile_enable", "hsr_profile_read", "hsr_stage_name",
            "hsr_frame_prep_forward", "hsr_frame_prep_backward", "hsr_frame_prep_scratch_bytes",
            "hsr_loss_l1", "hsr_loss_ssim", "hsr_loss_tree_ce", "hsr_loss_leaf_mlp_ce", "hsr_loss_scratch_bytes", "hsr_densify_frame", "hsr_densify_scratch_bytes",
            "hsr_prune_mask", "hsr_compact_append_rows", "hsr_compact_scratch_bytes"} <= set(protos)
    lib = C.CDLL(_C._LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), "libhsr_rast.so does not export %s" % name


def test_ctypes_signatures_match_header():
    from diff_gaussian_rasterization import _C
    protos = _prototypes()
    for name in ("hsr_forward", "hsr_forward_semantic", "hsr_backward", "hsr_backward_semantic", "hsr_mark_visible",
                 "hsr_get_state_layout", "hsr_required_binning_bytes", "hsr_required_image_bytes"):
        fn = getattr(_C._lib, name)
        assert len(fn.argtypes) == protos[name], (name, len(fn.argtypes), protos[name])
    from hsr_utils import slam_helpers
    for name in ("hsr_frame_prep_forward", "hsr_frame_prep_backward", "hsr_frame_prep_backward_params", "hsr_frame_prep_scratch_bytes"):
        fn = getattr(slam_helpers._lib, name)
        assert len(fn.argtypes) == protos[name], (name, len(fn.argtypes), protos[name])
    from hsr_utils import densify
    for name in ("hsr_densify_frame", "hsr_densify_scratch_bytes"):
        fn = getattr(densify._lib, name)
        assert len(fn.argtypes) == protos[name], (name, len(fn.argtypes), protos[name])
    from hsr_utils import slam_external
    for name in ("hsr_prune_mask", "hsr_compact_append_rows", "hsr_compact_scratch_bytes"):
        fn = getattr(slam_external._lib, name)
        assert len(fn.argtypes) == protos[name], (name, len(fn.argtypes), protos[name])
    from hsr_utils import losses
    for name in ("hsr_loss_l1", "hsr_loss_l1_grad", "hsr_loss_ssim", "hsr_loss_ssim_value", "hsr_loss_ssim_grad", "hsr_loss_tree_ce", "hsr_loss_tree_ce_value", "hsr_loss_tree_ce_grad", "hsr_loss_tree_ce_scratch_bytes",
                 "hsr_loss_tracking_value", "hsr_loss_tracking_grad", "hsr_loss_tracking_scratch_bytes", "hsr_loss_leaf_mlp_ce", "hsr_loss_scratch_bytes"):
        fn = getattr(losses._lib, name)
        assert len(fn.argtypes) == protos[name], (name, len(fn.argtypes), protos[name])


def test_host_only_entry_points():
    from diff_gaussian_rasterization import _C
    lib = _C._lib
    assert b"gfx950" in lib.hsr_version()
    g1, g2 = lib.hsr_required_geometry_bytes(1000), lib.hsr_required_geometry_bytes(2000)
    assert 0 < g1 < g2
    assert lib.hsr_required_image_bytes(1200, 680) > 8 * 1200 * 680
    assert lib.hsr_required_binning_bytes(0) > 0 and lib.hsr_required_binning_bytes(10 ** 6) > 24 * 10 ** 6
    lay = _C.state_layout(1000, 64, 48, 5000)
    offs = [lay[k] for k in ("geom_depths", "geom_means2D", "geom_conic_opacity", "geom_cov3D")]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in lay.values())
    for i in range(9):
        assert lib.hsr_stage_name(i) not in (None, b"?")
    assert lib.hsr_stage_name(99) == b"?"


def test_argument_validation_without_gpu():
    """invalid sizes are rejected before any device work, with a message in hsr_last_error()"""
    from diff_gaussian_rasterization import _C
    lib = _C._lib
    null = None
    rc = lib.hsr_forward_semantic(None, None, None, 10, 0, 0, 26, null, 0, 48, *([null] * 6), 1.0, *([null] * 5), 1.0, 1.0, 0,
                                  *([null] * 6), 0, null)
    assert rc == -1 and b"invalid sizes" in lib.hsr_last_error()
    rc = lib.hsr_mark_visible(-1, null, null, null, null, null)
    assert rc == -1
    from hsr_utils import slam_helpers  # sets the argtypes of the frame-prep entry points
    assert slam_helpers._lib.hsr_frame_prep_scratch_bytes(0) > 0
    assert slam_helpers._lib.hsr_frame_prep_scratch_bytes(10 ** 6) >= (10 ** 6 // 1024) * 64
    rc = lib.hsr_frame_prep_forward(10, 2, 0, 0, *([null] * 6), 1, 0, *([null] * 8))
    assert rc == -1 and b"log_scales must be" in lib.hsr_last_error()
    rc = lib.hsr_frame_prep_backward(0, 1, 0, 0, *([null] * 6), 1, 5, *([null] * 14), 0, null)
    assert rc == -1 and b"time_idx" in lib.hsr_last_error()
    from hsr_utils import losses  # sets the argtypes of the loss entry points
    assert losses._lib.hsr_loss_scratch_bytes(3, 680, 1200) >= 3 * 3 * 680 * 1200 * 4
    assert lib.hsr_loss_l1(0, 8, 8, null, null, null, 0, null, null, null, 0, null) == -1 and b"loss_l1" in lib.hsr_last_error()
    assert lib.hsr_loss_ssim(3, 0, 8, null, null, null, null, null, 0, null) == -1
    assert lib.hsr_loss_tree_ce(4, 8, 8, 1, None, None, null, null, -100, null, null, null, 0, null) == -1
    assert lib.hsr_loss_leaf_mlp_ce(40, 10, 8, 8, null, null, null, null, -100, null, null, null, null, null, 0, null) == -1
    assert b"K <= 31" in lib.hsr_last_error()
    from hsr_utils import densify  # sets the argtypes of the densification entry points
    assert densify._lib.hsr_densify_scratch_bytes(680, 1200) > 680 * 1200 // 256 * 4
    assert lib.hsr_densify_frame(0, 8, *([null] * 4), 1.0, 1.0, 0.0, 0.0, null, 0.5, 50.0, 0, *([null] * 7), null, 0, null) == -1


def test_no_cpu_fallback_and_reference_error_messages():
    """CPU tensors must fail loudly (there is no fallback path); the either/or argument checks keep the
    reference's messages (diff_gaussian_rasterization/__init__.py:195-199)"""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizer_semantic
    import scenes
    cam, sc, _ = scenes.build(32, 32, 10, 4)
    rs = GaussianRasterizationSettings(**cam)
    m2 = torch.zeros(10, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        GaussianRasterizer(rs)(means3D=sc["means3D"], means2D=m2, opacities=sc["opacities"], scales=sc["scales"],
                               rotations=sc["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        GaussianRasterizer_semantic(rs)(means3D=sc["means3D"], means2D=m2, opacities=sc["opacities"],
                                        colors_precomp=sc["colors_precomp"], scales=sc["scales"])
    with pytest.raises(RuntimeError, match="no CPU path"):
        GaussianRasterizer_semantic(rs)(means3D=sc["means3D"], means2D=m2, opacities=sc["opacities"],
                                        colors_precomp=sc["colors_precomp"], scales=sc["scales"], rotations=sc["rotations"],
                                        semantics_precomp=sc["semantics_precomp"])
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        GaussianRasterizer(rs)(means3D=torch.zeros(10, 4), means2D=m2, opacities=sc["opacities"],
                               colors_precomp=sc["colors_precomp"], scales=sc["scales"], rotations=sc["rotations"])
    assert GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                                     "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    import sys
    monkeypatch.setenv("HSR_RAST_LIB", str(tmp_path / "nope.so"))
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.startswith("diff_gaussian_rasterization")}
    try:
        with pytest.raises(ImportError, match="no CPU fallback"):
            importlib.import_module("diff_gaussian_rasterization")
    finally:
        for k in [k for k in sys.modules if k.startswith("diff_gaussian_rasterization")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_camera_matches_reference_convention():
    """setup_camera restatement (utils/recon_helpers.py:4-28): viewmatrix = w2c^T, projmatrix = viewmatrix @ P^T"""
    import numpy as np
    from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
    import scenes
    w2c = scenes.tilted_w2c()
    cam = setup_camera_tensors(1200, 680, replica_intrinsics(), w2c)
    assert np.allclose(cam["viewmatrix"][0].numpy(), w2c.T, atol=1e-7)
    assert abs(cam["tanfovx"] - 1.0) < 1e-12 and abs(cam["tanfovy"] - 680 / 1200.0) < 1e-12
    p = np.array([0.3, -0.2, 2.0, 1.0], np.float32)
    hom = p @ cam["projmatrix"][0].numpy()
    cam_pt = w2c @ p
    assert abs(hom[3] - cam_pt[2]) < 1e-5  # w = view-space depth for this projection
    assert np.allclose(cam["campos"].numpy(), np.linalg.inv(w2c)[:3, 3], atol=1e-6)
