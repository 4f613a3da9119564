"""CPU suite for the loss-head row: oracle/loss_oracle.py against the committed outputs of the
reference's own helpers (tests/golden/loss_ssim_l1.npz: calc_ssim and l1_loss_v1 imported from the reference, values and
autograd gradients) and of torch.nn.CrossEntropyLoss applied per tree level as the reference applies it."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import loss_oracle as LO  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SSIM_GRAD_TOL = 5e-4


@pytest.mark.parametrize("case", ["random_40x56", "smooth_68x120", "tiny_7x9"])
def test_ssim_and_l1_match_reference_outputs(case):
    d = np.load(os.path.join(GOLD, "loss_ssim_l1.npz"))
    x, y = d[case + "/img1"], d[case + "/img2"]
    v, g = LO.ssim(x, y)
    assert abs(v - float(d[case + "/ssim"])) < 2e-6
    ref_g = d[case + "/ssim_grad"]
    # the reference evaluates sigma = E[x^2] - mu^2 in fp32: on smooth images (sigma ~ 1e-3) that cancellation leaves its
    # own gradient ~2e-4 (relative to the largest entry) away from the float64 value; random images agree to 2e-6
    assert np.abs(g - ref_g).max() <= SSIM_GRAD_TOL * np.abs(ref_g).max()
    l, lg = LO.l1_mean(x, y)
    assert abs(l - float(d[case + "/l1"])) < 1e-6
    np.testing.assert_allclose(lg, d[case + "/l1_grad"], rtol=1e-6, atol=1e-12)


def test_window_is_the_reference_window():
    w = LO.window_2d()
    assert w.shape == (11, 11) and w.dtype == np.float32
    assert abs(float(w.sum()) - 1.0) < 1e-6 and np.array_equal(w, w.T) and np.array_equal(w, w[::-1, ::-1])


def test_tree_cross_entropy_matches_torch():
    d = np.load(os.path.join(GOLD, "loss_tree_ce.npz"))
    losses, grad = LO.tree_cross_entropy(d["logits"], d["labels"], list(d["level_sizes"]))
    np.testing.assert_allclose(losses, d["per_level"], rtol=2e-6)
    assert np.abs(grad - d["grad"]).max() <= 2e-6 * np.abs(d["grad"]).max()


@pytest.mark.parametrize("reduction", ["sum", "mean"])
def test_masked_l1_matches_torch_indexing(reduction):
    g = np.random.default_rng(3)
    pred, gt = g.random((3, 20, 30)).astype(np.float32), g.random((3, 20, 30)).astype(np.float32)
    mask = g.random((20, 30)) > 0.4
    tp = torch.tensor(pred, requires_grad=True)
    sel = torch.abs(torch.tensor(gt) - tp)[torch.tile(torch.tensor(mask), (3, 1, 1))]   # scripts/hierslam.py:933-935
    ref = sel.sum() if reduction == "sum" else sel.mean()
    ref.backward()
    loss, grad = LO.masked_l1(pred, gt, mask, reduction)
    assert abs(loss - float(ref)) <= 1e-5 * abs(float(ref))
    np.testing.assert_allclose(grad, tp.grad.numpy(), rtol=1e-6, atol=1e-12)


def test_leaf_mlp_head_matches_torch_fixture():
    d = np.load(os.path.join(GOLD, "loss_leaf_mlp.npz"))
    loss, ds, dw, db = LO.leaf_mlp_cross_entropy(d["sem"], d["weight"], d["bias"], d["labels"])
    assert abs(loss - float(d["loss"])) < 2e-6
    for got, key in ((ds, "d_sem"), (dw, "d_weight"), (db, "d_bias")):
        assert np.abs(got - d[key]).max() <= 3e-6 * np.abs(d[key]).max(), key
