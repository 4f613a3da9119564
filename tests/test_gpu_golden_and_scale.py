"""-m gpu: (1) the HIP path against the committed golden fixtures (tests/golden/*.npz; no oracle, no
/root/reference at run time); (2) BASELINE.json's full sizes through size-independent properties."""
import os

import numpy as np
import pytest
import torch

import scenes
from harness import _cam_to, assert_close, run_gpu

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".npz") and not f.startswith(("loss_", "densify_"))))
def test_hip_matches_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    cam = dict(image_height=int(z["H"]), image_width=int(z["W"]), tanfovx=float(z["tanfovx"]), tanfovy=float(z["tanfovy"]),
               bg=torch.tensor(z["bg"]), scale_modifier=float(z["scale_modifier"]), viewmatrix=torch.tensor(z["viewmatrix"]),
               projmatrix=torch.tensor(z["projmatrix"]), sh_degree=int(z["sh_degree"]), campos=torch.tensor(z["campos"]),
               prefiltered=False, debug=False)
    semantic = bool(z["semantic"])
    sc = {n: torch.tensor(z[n]) for n in ("means3D", "opacities", "colors_precomp", "scales", "rotations", "semantics_precomp")}
    up = dict(color=torch.tensor(z["up_color"]), semantic=torch.tensor(z["up_semantic"]), depth=torch.tensor(z["up_depth"]),
              median=torch.tensor(z["up_median"]), opacity=torch.tensor(z["up_opacity"]))
    out, gr, st = run_gpu(cam, sc, up, semantic=semantic)
    assert st["num_rendered"] == int(z["exp_num_rendered"])
    assert np.array_equal(out["radii"], z["exp_radii"])
    for n in ("keys", "vals", "ranges", "tiles_touched"):
        assert np.array_equal(st[n], z["exp_" + n]), n
    assert int((st["n_contrib"] != z["exp_n_contrib"]).sum()) <= 2
    for n in ("color", "depth", "opacity") + (("semantic",) if semantic else ("mask",)):
        assert_close(n, out[n], z["exp_" + n])
    assert int((np.abs(out["median_depth"] - z["exp_median_depth"]) > 1e-4).sum()) <= 2
    for n in ("means3D", "means2D", "opacities", "colors_precomp", "scales", "rotations") + (("semantics_precomp",) if semantic else ()):
        assert_close("grad " + n, gr[n], z["exp_grad_" + n])


def _full_size(P, W, H, K, kind="slam"):
    from diff_gaussian_rasterization import GaussianRasterizer_semantic, _C
    from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
    from hsr_utils.synthetic import make_scene
    dev = torch.device("cuda:0")
    kmat = replica_intrinsics(W, H)
    cam = setup_camera_tensors(W, H, kmat, np.eye(4))
    sc = make_scene(P, W, H, K, kmat, seed=0, kind=kind)
    leaf = {n: sc[n].to(dev).requires_grad_(True) for n in ("means3D", "opacities", "colors_precomp", "scales", "rotations",
                                                            "semantics_precomp")}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    outs = GaussianRasterizer_semantic(_cam_to(cam, dev))(means2D=m2, **leaf)
    return cam, sc, leaf, m2, outs


@pytest.mark.parametrize("P,W,H,K", [(500000, 1200, 680, 26), (300000, 1200, 680, 16), (2000000, 1920, 1080, 74)])
def test_full_size_properties(P, W, H, K):
    """headline / config-2 / stress sizes: sortedness, range consistency, compositing identities, finiteness"""
    from diff_gaussian_rasterization import _C
    cam, sc, leaf, m2, (color, radii, sem, depth, median, opac) = _full_size(P, W, H, K)
    node = color.grad_fn
    R = node.num_rendered
    saved = node.saved_tensors
    geom, binning, img = saved[-3], saved[-2], saved[-1]
    lay = _C.state_layout(P, W, H, R)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    b0 = (-binning.data_ptr()) % 256
    keys = binning[b0 + lay["bin_keys"]: b0 + lay["bin_keys"] + 8 * R].view(torch.int64)
    vals = binning[b0 + lay["bin_vals"]: b0 + lay["bin_vals"] + 4 * R].view(torch.int32)
    i0 = (-img.data_ptr()) % 256
    ranges = img[i0 + lay["img_ranges"]: i0 + lay["img_ranges"] + 8 * T].view(torch.int32).reshape(T, 2).long()
    g0 = (-geom.data_ptr()) % 256
    touched = geom[g0 + lay["geom_tiles_touched"]: g0 + lay["geom_tiles_touched"] + 4 * P].view(torch.int32).long()
    depths = geom[g0 + lay["geom_depths"]: g0 + lay["geom_depths"] + 4 * P].view(torch.float32)
    # num_rendered = sum of tile counts; keys sorted; key = (tile << 32) | depth bits of its Gaussian
    assert int(touched.sum()) == R and int((radii > 0).sum()) == int((touched > 0).sum())
    assert bool((keys[1:] >= keys[:-1]).all())
    tiles = keys >> 32
    assert int(tiles.min()) >= 0 and int(tiles.max()) < T
    assert bool(((keys & 0xFFFFFFFF).int() == depths[vals.long()].view(torch.int32)).all())
    # ranges partition [0, R) by tile, in order
    lens = ranges[:, 1] - ranges[:, 0]
    assert int(lens.sum()) == R and bool((lens >= 0).all())
    cnt = torch.bincount(tiles, minlength=T)
    assert bool((cnt == lens).all())
    nz = lens > 0
    assert bool((tiles[ranges[nz, 0]] == torch.nonzero(nz).squeeze(1)).all())
    # every Gaussian appears exactly tiles_touched times
    assert bool((torch.bincount(vals.long(), minlength=P) == touched).all())
    # images: finite, opacity in [0, 1), colour bounded by opacity (colours in [0,1))
    for t in (color, sem, depth, median, opac):
        assert bool(torch.isfinite(t).all())
    assert float(opac.min()) >= 0 and float(opac.max()) < 1
    assert float((color - opac).max()) <= 1e-5
    # linearity of the backward in the upstream gradient + identities with all-ones upstream
    ones = [torch.zeros_like(color), torch.ones_like(sem), torch.ones_like(depth), torch.zeros_like(median), torch.zeros_like(opac)]
    torch.autograd.backward([color, sem, depth, median, opac], ones, inputs=list(leaf.values()) + [m2])
    tot = float(opac.double().sum())
    assert abs(float(leaf["semantics_precomp"].grad.double().sum()) - K * tot) <= 2e-4 * K * tot
    for n, p in leaf.items():
        assert bool(torch.isfinite(p.grad).all()), n
    # colour got no upstream gradient
    assert float(leaf["colors_precomp"].grad.abs().max()) == 0


def test_headline_size_parity_vs_oracle():
    """BASELINE.json's headline workload — 1200x680, 500k SLAM-like Gaussians, K = 26 — HIP against the oracle on the same
    inputs: the comparison bench.py also prints in its `parity` block (tests/harness.parity_report)."""
    import json
    from harness import parity_report, run_oracle
    from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
    from hsr_utils.synthetic import make_scene, make_upstream_grads
    W, H, P, K = 1200, 680, 500000, 26
    kmat = replica_intrinsics(W, H)
    cam = setup_camera_tensors(W, H, kmat, np.eye(4))
    sc = make_scene(P, W, H, K, kmat, seed=0, kind="slam")
    up = {n: v * float(W * H) for n, v in make_upstream_grads(W, H, K, seed=1).items()}   # O(1) upstream gradients
    out_g, gr_g, st_g = run_gpu(cam, sc, up, semantic=True)
    out_o, gr_o, st_o = run_oracle(cam, sc, up, semantic=True)
    rep = parity_report(out_g, gr_g, st_g, out_o, gr_o, st_o, semantic=True)
    print("headline parity:", json.dumps(rep))
    for k in ("num_rendered_equal", "radii_equal", "tiles_touched_equal", "keys_equal", "vals_equal", "ranges_equal"):
        assert rep[k], k
    assert rep["n_contrib_mismatch"] <= W * H // 2000 and rep["median_depth_outliers"] <= W * H // 2000
    for n in ("color", "depth", "opacity", "semantic"):
        assert_close(n, out_g[n], out_o[n])
    for n in gr_o:
        assert_close("grad " + n, gr_g[n], gr_o[n])
    st_o.free()


def test_forward_is_deterministic_and_backward_reproducible_within_fp32_noise():
    cam, sc, up = scenes.build(320, 200, 20000, 26, seed=9, kind="slam", scale_mult=2.0)
    o1, g1, s1 = run_gpu(cam, sc, up)
    o2, g2, s2 = run_gpu(cam, sc, up)
    for n in o1:
        assert np.array_equal(o1[n], o2[n]), n  # forward: no atomics, bit-reproducible
    assert np.array_equal(s1["keys"], s2["keys"]) and np.array_equal(s1["vals"], s2["vals"])
    for n in g1:
        assert_close("grad " + n, g1[n], g2[n], rtol=1e-5, atol=1e-6)  # fp32 atomics: order-dependent last bits


def test_experimental_modes_are_not_in_the_product_library():
    """the per-instance rows accumulation (and the moments kernels) were measured slower and live in the
    ablate build only (csrc/experiments/, `make -C hier-slam_amd/csrc ablate`): the product refuses the mode"""
    from diff_gaussian_rasterization import _C
    with pytest.raises(RuntimeError, match="ablate build"):
        _C.set_backward_mode("rows")
    assert int(_C._lib.hsr_get_backward_mode()) == 0


@pytest.mark.parametrize("mode", ["legacy"])
@pytest.mark.parametrize("name", ["replica_tree_k26", "scannet_tree_k16", "generic_k5_white_bg", "plain_mask", "huge_splats",
                                  "culled_behind_camera", "large_tree_k74"])
def test_parity_other_accumulation_modes(name, mode):
    """the default 'packed' mode is what test_gpu_parity.py exercises; 'legacy' = atomics straight into the reference's six arrays"""
    from diff_gaussian_rasterization import _C
    from test_gpu_parity import CASES, _compare
    W, H, P, K, kind, sm, semantic, variant, bg, behind = CASES[name]
    cam, sc, up = scenes.build(W, H, P, K, seed=11, kind=kind, scale_mult=sm, bg=bg, behind_frac=behind)
    _C.set_backward_mode(mode)
    try:
        _compare(cam, sc, up, semantic, variant, None)
    finally:
        _C.set_backward_mode("packed")


ABLATE_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hier-slam_amd", "libhsr_rast_ablate.so")


def _ablate_env(**extra):
    """environment of a child process that runs the diagnostic library (`make -C hier-slam_amd/csrc ablate`: csrc/experiments/ linked
    in, ablation switches live) through the ctypes glue; skips the test when that library was not built"""
    if not os.path.exists(ABLATE_LIB):
        pytest.skip("libhsr_rast_ablate.so not built (make -C hier-slam_amd/csrc ablate): experimental kernel families are not in the product")
    return dict(os.environ, HSR_RAST_LIB=ABLATE_LIB, HSR_GLUE="ctypes", **extra)


@pytest.mark.parametrize("impl", ["mfma", "valu", "sub"])
def test_parity_alternate_kernels(impl):
    """kernel families are selected per process (HSR_FWD_IMPL / HSR_BWD_IMPL; defaults: per-lane forward and matrix-core backward
    on 4x4 sub-block lists — since round 4 the Q-panel kernels of hsr_render_bwd_q.hip for K <= 27 and the geometry-only path; "sub" =
    round 3's butterfly kernels, kept in the product library for A/B timing).  "valu" = quadrant-list per-lane kernels both ways — the product's fallback for the legacy accumulation
    mode and beyond 2^30 row elements; "mfma" = round 1's quadrant-list matrix-core backward, since round 3 in the ablate build only
    (csrc/experiments/).  Parity cases, each family in a child process"""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0]=['hier-slam_amd','tests'];import scenes;from test_gpu_parity import CASES,_compare;"
            "W,H,P,K,kind,sm,sem,var,bg,beh=CASES['replica_tree_k26'];cam,sc,up=scenes.build(W,H,P,K,seed=11,kind=kind,scale_mult=sm,bg=bg);"
            "_compare(cam,sc,up,sem,var,None);"
            "[_compare(*((lambda W,H,P,K,kind,sm,sem,var,bg,beh: (lambda csu: (csu[0],csu[1],csu[2],sem,var,None))(scenes.build(W,H,P,K,seed=11,kind=kind,scale_mult=sm,bg=bg,behind_frac=beh)))(*CASES[n]))) "
            "for n in ('scannet_tree_k16','generic_k5_white_bg','plain_mask','huge_splats','deep_tiles_3000','large_tree_k74')];print('ok')")
    # "valu": the all-VALU backward in the PRODUCT library (its fallback kernel); the quadrant-list forward of the same name exists in
    # the ablate build only and is covered by test_parity_round1_wide_kernels_in_the_ablate_build
    env = _ablate_env(HSR_BWD_IMPL=impl) if impl == "mfma" else dict(os.environ, HSR_BWD_IMPL=impl)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_parity_wide_tree_kernel_selectors():
    """wide trees (K > 27) default to ONE backward pass of up to 112 columns at two waves per SIMD; HSR_BWD_WIDE_PASS=split
    selects the earlier 64-column passes (what K + 5 > 112 still takes); 49..80 columns contract their panels on the bf16 matrix
    cores (exact three-way split of the fp32 operands), HSR_BWD_WIDE_MMA=f32 keeps them on the fp32 matrix instructions:
    same results, parity cases in a child process"""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0]=['hier-slam_amd','tests'];import scenes;from test_gpu_parity import CASES,_compare;"
            "[_compare(*((lambda W,H,P,K,kind,sm,sem,var,bg,beh: (lambda csu: (csu[0],csu[1],csu[2],sem,var,None))(scenes.build(W,H,P,K,seed=11,kind=kind,scale_mult=sm,bg=bg,behind_frac=beh)))(*CASES[n]))) "
            "for n in ('generic_k40_two_chunks','large_tree_k74','flat_k102','odd_k33','odd_k75_ragged','k52_four_column_groups','k124_widest_single_pass','k130_chunked','wide_deep_tiles_k76')];print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, HSR_BWD_WIDE_PASS="split"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # the fp32-matrix-instruction variants of the 4- and 5-group passes and the K = 74 forward with its rows parked in registers are
    # A/B selectors of the ablate build
    if os.path.exists(ABLATE_LIB):
        for extra in (dict(HSR_BWD_WIDE_MMA="f32"), dict(HSR_FWD_PF="0")):
            r = subprocess.run([sys.executable, "-c", code], cwd=root, env=_ablate_env(**extra), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and "ok" in r.stdout, str(extra) + r.stdout[-2000:] + r.stderr[-2000:]


def test_parity_round1_wide_kernels_in_the_ablate_build():
    """round 1's matrix-core forward (29 <= K <= 124) and quadrant-list wide backward: csrc/experiments/, ablate build only"""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0]=['hier-slam_amd','tests'];import scenes;from test_gpu_parity import CASES,_compare;"
            "[_compare(*((lambda W,H,P,K,kind,sm,sem,var,bg,beh: (lambda csu: (csu[0],csu[1],csu[2],sem,var,None))(scenes.build(W,H,P,K,seed=11,kind=kind,scale_mult=sm,bg=bg,behind_frac=beh)))(*CASES[n]))) "
            "for n in ('generic_k40_two_chunks','large_tree_k74','flat_k102','odd_k33','odd_k75_ragged','k124_widest_single_pass','wide_deep_tiles_k76')];print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # + round 3's sub-block forward on the fp32 matrix cores (hsr_render_fwd_mma.hip)
    for extra in (dict(HSR_FWD_IMPL="wide"), dict(HSR_BWD_IMPL="mfma"), dict(HSR_FWD_IMPL="mma"), dict(HSR_FWD_IMPL="valu")):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=_ablate_env(**extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, str(extra) + r.stdout[-2000:] + r.stderr[-2000:]


def test_ctypes_glue_matches_too():
    """The four rasterize entry points run through the compiled torch glue (diff_gaussian_rasterization._hsr_torch) when it is
    built; HSR_GLUE=ctypes selects the pure-Python glue over the same C ABI.  Parity cases through that one, in a child process."""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0]=['hier-slam_amd','tests'];import scenes;from test_gpu_parity import CASES,_compare;"
            "from diff_gaussian_rasterization import _C; assert _C._ext is None;"
            "[_compare(*((lambda W,H,P,K,kind,sm,sem,var,bg,beh: (lambda csu: (csu[0],csu[1],csu[2],sem,var,None))(scenes.build(W,H,P,K,seed=11,kind=kind,scale_mult=sm,bg=bg,behind_frac=beh)))(*CASES[n]))) "
            "for n in ('replica_tree_k26','plain_mask','huge_splats','large_tree_k74')];print('ok')")
    env = dict(os.environ, HSR_GLUE="ctypes")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_compiled_glue_is_loaded():
    from diff_gaussian_rasterization import _C
    if os.environ.get("HSR_GLUE") == "ctypes":
        pytest.skip("HSR_GLUE=ctypes asks for the pure-Python glue")
    assert _C._ext is not None, "diff_gaussian_rasterization._hsr_torch is not built (python hier-slam_amd/csrc/build_torch_ext.py)"


@pytest.mark.parametrize("impl", ["radix", "wave", "block", "default"])
def test_radix_binning_path_matches(impl):
    """HSR_SORT_IMPL=radix: emission in Gaussian order + stable tile-bit radix passes + per-tile sort (the path images of more
    than 8192 tiles take) instead of direct tile binning — same sorted keys, values, ranges and offsets, bit for bit.
    HSR_SORT_IMPL=wave / block: the per-tile sort by one wave per tile / by whole workgroups instead of the default two waves per
    tile (tile_sort_pair_kernel): every tile size class (<= 256, <= 512, <= 1024, <= 2048, beyond) occurs in the four scenes."""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0]=['hier-slam_amd','tests'];import scenes;from test_gpu_parity import CASES,_compare;"
            "[_compare(*((lambda W,H,P,K,kind,sm,sem,var,bg,beh: (lambda csu: (csu[0],csu[1],csu[2],sem,var,None))(scenes.build(W,H,P,K,seed=11,kind=kind,scale_mult=sm,bg=bg,behind_frac=beh)))(*CASES[n]))) "
            "for n in ('replica_tree_k26','plain_mask','huge_splats','deep_tiles_3000')];"
            # screen-filling splats: every tile holds (nearly) all P entries, one scene per size class of the per-tile sort and its edges
            "[_compare(*scenes.build(64,48,P,3,seed=P,kind='aniso',scale_mult=40.0), True, 'sr', None) for P in (60,128,129,250,257,500,513,800,1024,1025,1500,2100)];"
            "print('ok')")
    env = dict(os.environ, HSR_SORT_IMPL=impl)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_more_than_8192_tiles_take_the_radix_path():
    """2064 x 1040 = 129 x 65 = 8385 tiles: beyond the LDS tile counters of the direct binning, so the library switches to the
    radix path by itself; state and outputs against the oracle as everywhere else."""
    import scenes
    from test_gpu_parity import _compare
    cam, sc, up = scenes.build(2064, 1040, 4000, 5, seed=3, kind="slam", scale_mult=3.0, bg=(0, 0, 0), behind_frac=0.0)
    _compare(cam, sc, up, True, "sr", None)
