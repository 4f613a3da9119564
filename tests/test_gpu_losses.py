"""GPU parity for the fused loss heads (include/hsr_losses.h): against the committed outputs of the reference's own
calc_ssim / l1_loss_v1 (tests/golden/loss_ssim_l1.npz), torch.nn.CrossEntropyLoss per tree level (loss_tree_ce.npz),
oracle/loss_oracle.py on ragged sizes and masks, size-independent properties at 1200x680, and chained behind the rasterizer."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")

VAL_TOL = 2e-6        # loss values (fp32 sums of ~1e4..1e6 terms, two-stage, finish in double)
SSIM_VAL_TOL = 1e-5   # SSIM value: the fp32 cancellation below moves the reference's own value 1e-6 off float64, ours 4e-6
SSIM_GRAD_TOL = 5e-4  # of the largest entry: sigma = E[x^2] - mu^2 cancels in fp32, in the reference too (test_loss_oracle.py)
GRAD_TOL = 2e-6       # L1 / CE gradients, of the largest entry


def _relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("case", ["random_40x56", "smooth_68x120", "tiny_7x9"])
def test_ssim_and_l1_match_reference_outputs(case):
    from hsr_utils import losses as L
    d = np.load(os.path.join(GOLD, "loss_ssim_l1.npz"))
    x = torch.tensor(d[case + "/img1"], device="cuda", requires_grad=True)
    y = torch.tensor(d[case + "/img2"], device="cuda")
    s = L.calc_ssim(x, y)
    s.backward()
    assert abs(float(s) - float(d[case + "/ssim"])) < SSIM_VAL_TOL
    assert _relmax(x.grad.cpu().numpy(), d[case + "/ssim_grad"]) < SSIM_GRAD_TOL
    x.grad = None
    l = L.l1_loss_v1(x, y)
    l.backward()
    assert abs(float(l) - float(d[case + "/l1"])) < VAL_TOL
    assert _relmax(x.grad.cpu().numpy(), d[case + "/l1_grad"]) < GRAD_TOL


def test_tree_ce_matches_torch_fixture():
    from hsr_utils import losses as L
    d = np.load(os.path.join(GOLD, "loss_tree_ce.npz"))
    z = torch.tensor(d["logits"], device="cuda", requires_grad=True)
    total, levels = L.tree_cross_entropy(z, torch.tensor(d["labels"], device="cuda"), list(d["level_sizes"]), return_levels=True)
    total.backward()
    np.testing.assert_allclose(levels.cpu().numpy(), d["per_level"], rtol=3e-6)
    assert abs(float(total) - d["per_level"].sum()) < 1e-5
    assert _relmax(z.grad.cpu().numpy(), d["grad"]) < GRAD_TOL


@pytest.mark.parametrize("C,H,W", [(3, 37, 53), (1, 64, 48), (3, 16, 16), (2, 5, 130)])
def test_ssim_and_l1_against_oracle_on_ragged_sizes(C, H, W):
    import loss_oracle as LO
    from hsr_utils import losses as L
    g = np.random.default_rng(H * W)
    x, y = g.random((C, H, W)).astype(np.float32), g.random((C, H, W)).astype(np.float32)
    tx = torch.tensor(x, device="cuda", requires_grad=True)
    ty = torch.tensor(y, device="cuda")
    s = L.calc_ssim(tx, ty)
    s.backward()
    v, gr = LO.ssim(x, y)
    assert abs(float(s) - v) < SSIM_VAL_TOL and _relmax(tx.grad.cpu().numpy(), gr) < SSIM_GRAD_TOL
    for reduction in ("sum", "mean"):
        for frac in (0.6, 0.0, None):
            mask = None if frac is None else (g.random((H, W)) < frac)
            tx.grad = None
            l = L.masked_l1(tx, ty, None if mask is None else torch.tensor(mask, device="cuda"), reduction)
            l.backward()
            lo, go = LO.masked_l1(x, y, np.ones((H, W), bool) if mask is None else mask, reduction)
            if np.isnan(lo):
                assert np.isnan(float(l)) and not tx.grad.any()      # mean over an empty selection, like torch
            else:
                assert abs(float(l) - lo) <= 3e-6 * max(1.0, abs(lo)), (reduction, frac)
                assert _relmax(tx.grad.cpu().numpy(), go) < GRAD_TOL if np.abs(go).max() > 0 else not tx.grad.any()


def test_masked_l1_ignores_nan_outside_the_mask():
    """the reference's mask exists to drop NaN depths (scripts/hierslam.py:910): unselected NaNs must not leak"""
    from hsr_utils import losses as L
    d = torch.rand(1, 33, 47, device="cuda")
    gt = torch.rand(1, 33, 47, device="cuda")
    d[0, 3, 4] = float("nan")
    mask = ~torch.isnan(d)
    x = d.clone().requires_grad_(True)
    l = L.masked_l1(x, gt, mask, "mean")
    l.backward()
    ref = torch.abs(gt - d)[mask].mean()
    assert torch.isfinite(l) and abs(float(l) - float(ref)) < 1e-6 and torch.isfinite(x.grad).all() and x.grad[0, 3, 4] == 0


def test_tree_ce_against_oracle_with_weights_ignore_and_spare_channels():
    import loss_oracle as LO
    from hsr_utils import losses as L
    g = np.random.default_rng(5)
    sizes, K, H, W = [3, 5, 9], 20, 31, 45            # 3 channels behind the last level
    z = g.normal(0, 3, (K, H, W)).astype(np.float32)
    lab = np.stack([g.integers(0, n, (H, W)) for n in sizes] + [g.integers(0, 7, (H, W))]).astype(np.int64)
    lab[0, :4] = -100
    lab[2] = -100                                      # a level with no valid pixel: NaN loss, like torch
    w = [1.0, 5.0, 0.5]
    tz = torch.tensor(z, device="cuda", requires_grad=True)
    total, levels = L.tree_cross_entropy(tz, torch.tensor(lab, device="cuda"), sizes, weights=w, return_levels=True)
    lo, go = LO.tree_cross_entropy(z, lab, sizes)
    lv = levels.cpu().numpy()
    np.testing.assert_allclose(lv[:2], lo[:2], rtol=3e-6)
    assert np.isnan(lv[2]) and np.isnan(lo[2])
    (levels[0] * 0 + total).backward() if False else None
    # gradient of the weighted sum over the levels that have valid pixels
    tz.grad = None
    total2, _ = L.tree_cross_entropy(tz, torch.tensor(lab[:2], device="cuda"), sizes[:2], weights=w[:2], return_levels=True)
    total2.backward()
    gw = go.copy(); gw[:3] *= w[0]; gw[3:8] *= w[1]; gw[8:] = 0
    assert _relmax(tz.grad.cpu().numpy(), gw) < GRAD_TOL and not tz.grad[8:].any()
    # flat classes == one level
    tz.grad = None
    flat = L.cross_entropy_planar(tz, torch.tensor(g.integers(0, K, (H, W)), device="cuda"))
    flat.backward()
    assert torch.isfinite(flat) and abs(float(tz.grad.sum())) < 1e-4


def test_losses_reject_bad_input():
    from hsr_utils import losses as L
    a, b = torch.rand(3, 8, 8), torch.rand(3, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        L.calc_ssim(a, b)
    with pytest.raises(RuntimeError):
        L.l1_loss_v1(a.cuda(), torch.rand(3, 8, 9, device="cuda"))
    with pytest.raises(RuntimeError, match="levels cover"):
        L.tree_cross_entropy(a.cuda(), torch.zeros(2, 8, 8, dtype=torch.long, device="cuda"), [2, 2])
    with pytest.raises(NotImplementedError):
        L.calc_ssim(a.cuda(), b.cuda(), window_size=7)


def test_full_size_properties_and_reproducibility():
    """1200x680, K=26 Replica tree: identities that do not need the oracle"""
    from hsr_utils import losses as L
    H, W, sizes = 680, 1200, [2, 4, 6, 6, 8]
    g = torch.Generator(device="cpu").manual_seed(0)
    im = torch.rand(3, H, W, generator=g).cuda().requires_grad_(True)
    s = L.calc_ssim(im, im.detach())
    s.backward()
    assert abs(float(s) - 1.0) < 1e-6 and float(im.grad.abs().max()) < 1e-6      # SSIM(x, x) = 1 is a maximum
    gt = torch.rand(3, H, W, generator=g).cuda()
    im.grad = None
    l = L.l1_loss_v1(im, gt)
    l.backward()
    assert abs(float(l) - float((im.detach() - gt).abs().mean())) < 1e-6
    assert torch.equal(im.grad.abs(), torch.full_like(im, 1.0 / im.numel()))
    z = (torch.randn(26, H, W, generator=g) * 2).cuda().requires_grad_(True)
    lab = torch.stack([torch.randint(0, n, (H, W), generator=g) for n in sizes]).cuda()
    t1 = L.tree_cross_entropy(z, lab, sizes)
    t1.backward()
    g1 = z.grad.clone()
    b = 0
    for n in sizes:                                                              # softmax - onehot sums to 0 per pixel and level
        assert float(g1[b:b + n].sum(dim=0).abs().max()) < 1e-9
        b += n
    ref = sum(torch.nn.functional.cross_entropy(z.detach()[b0:b0 + n].permute(1, 2, 0).reshape(-1, n), lab[i].view(-1))
              for i, (b0, n) in enumerate(zip(np.cumsum([0] + sizes[:-1]), sizes)))
    assert abs(float(t1) - float(ref)) < 2e-5
    z.grad = None
    t2 = L.tree_cross_entropy(z, lab, sizes)
    t2.backward()
    assert torch.equal(t1, t2) and torch.equal(g1, z.grad)                       # fixed-order reductions


def test_losses_chained_behind_the_rasterizer():
    """mapping-style step: render -> 0.8 L1 + 0.2 (1 - SSIM) + masked depth L1 + tree CE -> backward; every Gaussian gradient
    against the same chain with torch eager losses in place of the fused heads"""
    from diff_gaussian_rasterization import GaussianRasterizer_semantic
    from hsr_utils import losses as L, setup_camera, make_scene
    import torch.nn.functional as F
    W, H, P, sizes = 320, 240, 20000, [2, 4, 6, 6, 8]
    K = sum(sizes)
    kmat = np.array([[300.0, 0, 159.5], [0, 300.0, 119.5], [0, 0, 1]])
    cam = setup_camera(W, H, kmat, np.eye(4), device="cuda")
    sc = make_scene(P, W, H, K, kmat, seed=9)
    g = torch.Generator().manual_seed(1)
    gt_im, gt_d = torch.rand(3, H, W, generator=g).cuda(), (torch.rand(1, H, W, generator=g) * 5).cuda()
    gt_d[0, :10] = 0                                                             # invalid depth rows
    lab = torch.stack([torch.randint(0, n, (H, W), generator=g) for n in sizes]).cuda()

    def window():
        w1 = torch.tensor([np.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
        w1 = (w1 / w1.sum()).unsqueeze(1)
        return w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0).expand(3, 1, 11, 11).contiguous().cuda()

    def eager_ssim(a, b):
        w = window()
        mu1, mu2 = F.conv2d(a, w, padding=5, groups=3), F.conv2d(b, w, padding=5, groups=3)
        s1 = F.conv2d(a * a, w, padding=5, groups=3) - mu1 ** 2
        s2 = F.conv2d(b * b, w, padding=5, groups=3) - mu2 ** 2
        s12 = F.conv2d(a * b, w, padding=5, groups=3) - mu1 * mu2
        return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 ** 2 + mu2 ** 2 + 1e-4) * (s1 + s2 + 9e-4))).mean()

    def run(fused):
        p = {k: v.clone().cuda().requires_grad_(True) for k, v in sc.items()}
        m2 = torch.zeros(P, 3, device="cuda", requires_grad=True)
        im, radius, sem, depth, med, opac = GaussianRasterizer_semantic(raster_settings=cam)(means2D=m2, **p)
        mask = (gt_d > 0) & ~torch.isnan(depth)
        if fused:
            loss = L.mapping_image_loss(im, gt_im) + L.masked_l1(depth, gt_d, mask, "mean") + 0.1 * L.tree_cross_entropy(sem, lab, sizes)
        else:
            ce, b = 0.0, 0
            for i, n in enumerate(sizes):
                ce = ce + F.cross_entropy(sem[b:b + n].permute(1, 2, 0).reshape(-1, n), lab[i].view(-1))
                b += n
            loss = 0.8 * torch.abs(im - gt_im).mean() + 0.2 * (1.0 - eager_ssim(im, gt_im)) + torch.abs(gt_d - depth)[mask].mean() + 0.1 * ce
        loss.backward()
        return float(loss), {k: v.grad.clone() for k, v in p.items()}
    la, ga = run(True)
    lb, gb = run(False)
    assert abs(la - lb) < 1e-5 * max(1.0, abs(lb))
    for k in ga:
        assert _relmax(ga[k].cpu().numpy(), gb[k].cpu().numpy()) < 1e-3, k


def test_leaf_mlp_head_matches_torch_fixture():
    from hsr_utils import losses as L
    d = np.load(os.path.join(GOLD, "loss_leaf_mlp.npz"))
    sem = torch.tensor(d["sem"], device="cuda", requires_grad=True)
    mlp = torch.nn.Conv2d(d["sem"].shape[0], d["weight"].shape[0], kernel_size=1).cuda()
    with torch.no_grad():
        mlp.weight.copy_(torch.tensor(d["weight"]).view_as(mlp.weight))
        mlp.bias.copy_(torch.tensor(d["bias"]))
    loss = L.leaf_mlp_cross_entropy(sem, mlp, torch.tensor(d["labels"], device="cuda"))
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 5e-6
    assert _relmax(sem.grad.cpu().numpy(), d["d_sem"]) < 1e-5
    assert _relmax(mlp.weight.grad.cpu().numpy().reshape(d["d_weight"].shape), d["d_weight"]) < 1e-5
    assert _relmax(mlp.bias.grad.cpu().numpy(), d["d_bias"]) < 1e-5


@pytest.mark.parametrize("K,C,H,W", [(16, 41, 33, 47), (26, 102, 64, 80), (5, 3, 17, 9), (31, 128, 24, 40), (8, 17, 300, 500)])
def test_leaf_mlp_head_against_oracle(K, C, H, W):
    import loss_oracle as LO
    from hsr_utils import losses as L
    g = np.random.default_rng(K * C)
    sem = g.normal(0, 1.5, (K, H, W)).astype(np.float32)
    w, b = g.normal(0, 0.4, (C, K)).astype(np.float32), g.normal(0, 0.3, (C,)).astype(np.float32)
    lab = g.integers(0, C, (H, W)).astype(np.int64)
    lab[0, :5] = -100
    ts = torch.tensor(sem, device="cuda", requires_grad=True)
    tw = torch.tensor(w.reshape(C, K, 1, 1), device="cuda", requires_grad=True)
    tb = torch.tensor(b, device="cuda", requires_grad=True)
    loss = L.leaf_mlp_cross_entropy(ts, (tw, tb), torch.tensor(lab, device="cuda"))
    (2.0 * loss).backward()
    lo, ds, dw, db = LO.leaf_mlp_cross_entropy(sem, w, b, lab)
    assert abs(float(loss) - lo) < 5e-6 * max(1.0, abs(lo))
    assert _relmax(ts.grad.cpu().numpy(), 2 * ds) < 1e-5
    assert _relmax(tw.grad.cpu().numpy().reshape(C, K), 2 * dw) < 2e-5      # sums over up to 150k pixels in fp32 partials
    assert _relmax(tb.grad.cpu().numpy(), 2 * db) < 2e-5
    # reproducible bit for bit (fixed partition, fixed-order finish)
    ts2 = torch.tensor(sem, device="cuda", requires_grad=True)
    tw2 = tw.detach().clone().requires_grad_(True)
    loss2 = L.leaf_mlp_cross_entropy(ts2, (tw2, tb.detach().clone().requires_grad_(True)), torch.tensor(lab, device="cuda"))
    (2.0 * loss2).backward()
    assert torch.equal(loss, loss2) and torch.equal(tw.grad, tw2.grad) and torch.equal(ts.grad, ts2.grad)


def test_leaf_mlp_head_wide_falls_back_to_conv_plus_fused_ce():
    from hsr_utils import losses as L
    K, C, H, W = 40, 150, 20, 30
    g = torch.Generator().manual_seed(0)
    sem = torch.randn(K, H, W, generator=g).cuda().requires_grad_(True)
    mlp = torch.nn.Conv2d(K, C, kernel_size=1).cuda()
    lab = torch.randint(0, C, (H, W), generator=g).cuda()
    loss = L.leaf_mlp_cross_entropy(sem, mlp, lab)
    loss.backward()
    ref = torch.nn.functional.cross_entropy(mlp(sem.detach().unsqueeze(0)).squeeze(0).view(C, -1).permute(1, 0), lab.view(-1))
    assert abs(float(loss) - float(ref)) < 1e-5 and sem.grad is not None and mlp.weight.grad is not None


def test_tree_ce_two_pass_form_matches_the_one_pass_entry_point_and_scales_by_the_upstream_gradient():
    """hsr_loss_tree_ce_value / _grad (what the autograd node calls since round 4) against hsr_loss_tree_ce (value and gradient in one
    pass, still exported): same level losses, same gradient; an upstream gradient g != 1 scales it inside the gradient pass; a level
    wider than the 16 channels the kernel holds in registers takes its streaming path."""
    import ctypes as C
    from hsr_utils import losses as L
    g = np.random.default_rng(17)
    for sizes, K, H, W in (([2, 4, 6, 6, 8], 26, 37, 61), ([3, 21, 5], 31, 19, 33)):   # second: a 21-channel level
        z = torch.tensor(g.normal(0, 3, (K, H, W)).astype(np.float32), device="cuda")
        lab_np = np.stack([g.integers(0, n, (H, W)) for n in sizes]).astype(np.int64)
        lab_np[1, :3] = -100
        lab = torch.tensor(lab_np, device="cuda")
        w = [1.0, 0.25, 2.0, 1.5, 0.5][:len(sizes)]
        n = len(sizes)
        csz, cw = (C.c_int * n)(*sizes), (C.c_float * n)(*w)
        out1, grad1 = torch.empty(n, device="cuda"), torch.empty_like(z)
        sc = torch.empty(int(L._lib.hsr_loss_scratch_bytes(K, H, W)), dtype=torch.uint8, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        assert L._lib.hsr_loss_tree_ce(K, H, W, n, csz, cw, z.data_ptr(), lab.data_ptr(), -100, out1.data_ptr(), grad1.data_ptr(),
                                       sc.data_ptr(), sc.numel(), s) == 0
        zz = z.clone().requires_grad_(True)
        total, levels = L.tree_cross_entropy(zz, lab, sizes, weights=w, return_levels=True)
        (3.5 * total).backward()
        torch.cuda.synchronize()
        np.testing.assert_allclose(levels.cpu().numpy(), out1.cpu().numpy(), rtol=2e-6)
        assert _relmax(zz.grad.cpu().numpy(), 3.5 * grad1.cpu().numpy()) < GRAD_TOL


def test_weighted_sum_is_the_python_arithmetic_it_replaces():
    from hsr_utils import losses as L
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(3, 40, 56, generator=g).cuda().requires_grad_(True), torch.rand(3, 40, 56, generator=g).cuda()
    ref = 0.5 * (0.8 * L.l1_loss_v1(a, b) + 0.2 * (1.0 - L.calc_ssim(a, b)))
    ref.backward()
    g_ref, a.grad = a.grad.clone(), None
    got = L.weighted_sum((L.l1_loss_v1(a, b), L.calc_ssim(a, b)), (0.4, -0.1), constant=0.1)
    got.backward()
    assert abs(float(got) - float(ref)) < 1e-6
    assert _relmax(a.grad.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6
    a.grad = None
    m = L.mapping_image_loss(a, b)
    m.backward()
    assert abs(float(m) - 2.0 * float(ref)) < 2e-6 and _relmax(a.grad.cpu().numpy(), 2.0 * g_ref.cpu().numpy()) < 1e-6
    with pytest.raises(RuntimeError, match="weighted_sum"):
        L.weighted_sum((got,), (1.0, 2.0))


def test_semantic_loss_mlp_is_the_two_heads_it_fuses():
    """L.semantic_loss_mlp (tree levels + leaf head as one node, the leaf head's stashed gradient riding in the tree gradient pass) against
    the two separate heads composed with Python arithmetic: value, d sem, d weight, d bias, with an upstream gradient != 1."""
    from hsr_utils import losses as L
    g = torch.Generator().manual_seed(11)
    sizes, K, Cc, H, W = [2, 4, 6, 6, 8], 26, 102, 45, 77
    sem = (torch.randn(K, H, W, generator=g) * 2).cuda()
    mlp = torch.nn.Conv2d(K, Cc, kernel_size=1).cuda()
    lab = torch.stack([torch.randint(0, n, (H, W), generator=g) for n in sizes] + [torch.randint(0, Cc, (H, W), generator=g)]).cuda()
    lab[1, :2] = -100
    lab[-1, 5:7] = -100
    a = sem.clone().requires_grad_(True)
    ref = 0.1 * L.tree_cross_entropy(a, lab[:5], sizes) + 0.5 * L.leaf_mlp_cross_entropy(a, mlp, lab[-1])
    (2.5 * ref).backward()
    want = (a.grad.clone(), mlp.weight.grad.clone(), mlp.bias.grad.clone())
    mlp.weight.grad = mlp.bias.grad = None
    b = sem.clone().requires_grad_(True)
    got, levels, leaf = L.semantic_loss_mlp(b, lab, sizes, mlp, weight_sem=(0.1, 0.5), return_parts=True)
    (2.5 * got).backward()
    torch.cuda.synchronize()
    assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref))
    assert abs(float(0.1 * levels.sum() + 0.5 * leaf) - float(got)) <= 2e-6 * abs(float(got))
    for name, x, y in (("d sem", b.grad, want[0]), ("d weight", mlp.weight.grad, want[1]), ("d bias", mlp.bias.grad, want[2])):
        assert _relmax(x.cpu().numpy(), y.cpu().numpy()) < 2e-6, name


@pytest.mark.parametrize("use_sil", [True, False])
def test_tracking_loss_is_the_reference_tracking_branch(use_sil):
    """L.tracking_loss against the torch expressions of scripts/hierslam.py:903-937 (mask from gt depth > 0, ~isnan(depth), silhouette >
    sil_thres; masked |.| sums; weights im 0.5 / depth 1.0), on a ragged size, with NaNs outside the mask, invalid gt depth, and an
    upstream gradient != 1."""
    from hsr_utils import losses as L
    g = torch.Generator().manual_seed(5)
    H, W = 61, 93
    im, gt_im = torch.rand(3, H, W, generator=g).cuda(), torch.rand(3, H, W, generator=g).cuda()
    depth, gt_d = (torch.rand(1, H, W, generator=g) * 5).cuda(), (torch.rand(1, H, W, generator=g) * 5).cuda()
    sil = torch.rand(1, H, W, generator=g).cuda()
    gt_d[0, :7] = 0.0
    depth[0, 10, 5:20] = float("nan")
    im[:, 30, 40] = gt_im[:, 30, 40]              # exact zeros of the error: gradient 0 there, as torch's sign(0)
    a, d = im.clone().requires_grad_(True), depth.clone().requires_grad_(True)
    mask = (gt_d > 0) & ~torch.isnan(d)
    if use_sil:
        mask = mask & (sil > 0.6)
    mask = mask.detach()
    ref_d = torch.abs(gt_d - d)[mask].sum()
    ref_c = torch.abs(gt_im - a)[torch.tile(mask, (3, 1, 1))].sum()
    ref = 1.0 * ref_d + 0.5 * ref_c
    (1.7 * ref).backward()
    want = (a.grad.clone(), d.grad.clone())
    a2, d2 = im.clone().requires_grad_(True), depth.clone().requires_grad_(True)
    got, parts = L.tracking_loss(a2, gt_im, d2, gt_d, sil if use_sil else None, sil_thres=0.6, use_sil_for_loss=use_sil, return_parts=True)
    (1.7 * got).backward()
    torch.cuda.synchronize()
    assert abs(float(got) - float(ref)) <= VAL_TOL * abs(float(ref))
    assert abs(float(parts[0]) - float(ref_d)) <= VAL_TOL * abs(float(ref_d)) and abs(float(parts[1]) - float(ref_c)) <= VAL_TOL * abs(float(ref_c))
    assert torch.equal(a2.grad, want[0]) and torch.equal(torch.nan_to_num(d2.grad), torch.nan_to_num(want[1]))
    assert not torch.isnan(d2.grad).any()          # unselected pixels get 0 whatever they hold


def test_tracking_loss_rejects_bad_input():
    from hsr_utils import losses as L
    z = torch.zeros(3, 8, 8, device="cuda")
    d = torch.zeros(1, 8, 8, device="cuda")
    with pytest.raises(RuntimeError, match="silhouette"):
        L.tracking_loss(z, z, d, d, None)
    with pytest.raises(RuntimeError, match="one size"):
        L.tracking_loss(z, z, torch.zeros(1, 8, 9, device="cuda"), d, d)
    with pytest.raises(RuntimeError, match="HIP device"):
        L.tracking_loss(z.cpu(), z, d, d, d)


def test_mapping_depth_loss_is_the_reference_mapping_branch():
    """L.mapping_depth_loss against torch.abs(gt - depth)[(gt > 0) & ~isnan(depth)].mean() (scripts/hierslam.py:905-927), value and gradient with
    an upstream gradient != 1; an empty selection gives NaN like torch."""
    from hsr_utils import losses as L
    g = torch.Generator().manual_seed(8)
    H, W = 57, 131
    depth, gt_d = (torch.rand(1, H, W, generator=g) * 5).cuda(), (torch.rand(1, H, W, generator=g) * 5).cuda()
    gt_d[0, :9] = 0.0
    depth[0, 20, 3:30] = float("nan")
    d = depth.clone().requires_grad_(True)
    mask = ((gt_d > 0) & ~torch.isnan(d)).detach()
    ref = torch.abs(gt_d - d)[mask].mean()
    (0.3 * ref).backward()
    want = d.grad.clone()
    d2 = depth.clone().requires_grad_(True)
    got = L.mapping_depth_loss(d2, gt_d)
    (0.3 * got).backward()
    torch.cuda.synchronize()
    assert abs(float(got) - float(ref)) <= VAL_TOL * abs(float(ref))
    assert _relmax(torch.nan_to_num(d2.grad).cpu().numpy(), torch.nan_to_num(want).cpu().numpy()) < GRAD_TOL and not torch.isnan(d2.grad).any()
    assert torch.isnan(L.mapping_depth_loss(depth, torch.zeros_like(gt_d)))
