"""CPU restatement (numpy, float64) of the loss heads the reference applies to the rendered maps — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's / tools' cpu_baseline legs may import this file; the product path
(hier-slam_amd/csrc/hsr_losses.hip behind include/hsr_losses.h) never does.

PINNED by the reference itself where the reference can run: `l1_loss_v1` (utils/slam_helpers.py:5-6) and `calc_ssim`
(utils/slam_external.py:66-97) are device-free and import on CPU, so tests/golden/loss_*.npz hold THEIR outputs and
autograd gradients (generator: tests/golden/make_loss_golden.py, run in the build container where /root/reference exists).
The masked L1 sums and the per-level cross-entropy are expressions inside get_loss_semantic_mlp (scripts/hierslam.py:921-1016,
not importable: needs cv2, wandb, a GPU); they are restated from those lines and pinned against torch's own
abs/sum/mean and torch.nn.CrossEntropyLoss, which is what those lines call.

Follows:
  l1_loss_v1                          utils/slam_helpers.py:5-6
  gaussian / create_window / _ssim    utils/slam_external.py:54-97
  masked depth / colour L1            scripts/hierslam.py:921-939
  transfer_tree_rendered_labelmap     scripts/hierslam.py:91-111
  multi-level cross-entropy           scripts/hierslam.py:963-974, :993-1003   (torch.nn.CrossEntropyLoss: mean over pixels,
                                                                                 ignore_index -100)
"""
import math

import numpy as np

C1, C2 = 0.01 ** 2, 0.03 ** 2


def window_1d(window_size=11, sigma=1.5):
    """gaussian() of the reference, evaluated the way it is: python floats -> torch.Tensor (float32) -> normalise in float32."""
    g = np.array([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=np.float32)
    return g / g.sum(dtype=np.float32)


def window_2d(window_size=11):
    w = window_1d(window_size)
    return (w[:, None] * w[None, :]).astype(np.float32)   # _1D_window.mm(_1D_window.t()) in float32


def _conv_same(img, w):
    """zero-padded cross-correlation of [C,H,W] with a [k,k] window (func.conv2d(..., padding=k//2, groups=C))."""
    k = w.shape[0]
    r = k // 2
    C, H, W = img.shape
    pad = np.zeros((C, H + 2 * r, W + 2 * r), dtype=np.float64)
    pad[:, r:r + H, r:r + W] = img
    out = np.zeros((C, H, W), dtype=np.float64)
    for dy in range(k):
        for dx in range(k):
            out += float(w[dy, dx]) * pad[:, dy:dy + H, dx:dx + W]
    return out


def l1_mean(x, y):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    return np.abs(x - y).mean(), np.sign(x - y) / x.size


def masked_l1(pred, gt, mask, reduction):
    """torch.abs(gt - pred)[mask].sum() or .mean()  (scripts/hierslam.py:925-927, :935); mask broadcast over channels.
    Returns (loss, dloss/dpred)."""
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    m = np.broadcast_to(np.asarray(mask, bool), pred.shape)
    d = np.where(m, pred - gt, 0.0)
    cnt = int(m.sum())
    scale = 1.0 if reduction == "sum" else (1.0 / cnt if cnt else float("nan"))
    loss = np.abs(d).sum() * scale if (reduction == "sum" or cnt) else float("nan")
    return loss, np.sign(d) * (scale if cnt or reduction == "sum" else 0.0)


def ssim(img1, img2, window_size=11):
    """calc_ssim(img1, img2) with size_average=True; returns (value, d value / d img1)."""
    x, y = np.asarray(img1, np.float64), np.asarray(img2, np.float64)
    w = window_2d(window_size)
    mu1, mu2 = _conv_same(x, w), _conv_same(y, w)
    s11, s22, s12 = _conv_same(x * x, w), _conv_same(y * y, w), _conv_same(x * y, w)
    sig1, sig2, sig12 = s11 - mu1 * mu1, s22 - mu2 * mu2, s12 - mu1 * mu2
    A1, A2 = 2 * mu1 * mu2 + C1, 2 * sig12 + C2
    B1, B2 = mu1 * mu1 + mu2 * mu2 + C1, sig1 + sig2 + C2
    smap = (A1 * A2) / (B1 * B2)
    n = smap.size
    # partials of the map w.r.t. the three window moments that depend on img1
    d_mu1 = (2 * mu2 * A2 - 2 * mu2 * A1) / (B1 * B2) - smap * (2 * mu1 / B1 - 2 * mu1 / B2)
    d_s11 = -smap / B2
    d_s12 = 2 * A1 / (B1 * B2)
    wf = w[::-1, ::-1]  # adjoint of a cross-correlation is the correlation with the flipped window (symmetric here)
    grad = (_conv_same(d_mu1, wf) + 2 * x * _conv_same(d_s11, wf) + y * _conv_same(d_s12, wf)) / n
    return smap.mean(), grad


def tree_cross_entropy(logits, labels, level_sizes, ignore_index=-100):
    """sum over levels of CrossEntropyLoss(logits[begin:end] as [H*W, n_l], labels[l])  (scripts/hierslam.py:963-974).
    logits [K,H,W]; labels [L,H,W] integer, L >= len(level_sizes).  Returns (per-level losses, d sum / d logits)."""
    z = np.asarray(logits, np.float64)
    K, H, W = z.shape
    grad = np.zeros_like(z)
    losses = []
    begin = 0
    for l, n_l in enumerate(level_sizes):
        zl = z[begin:begin + n_l].reshape(n_l, -1)
        lab = np.asarray(labels[l]).reshape(-1).astype(np.int64)
        valid = lab != ignore_index
        cnt = int(valid.sum())
        m = zl.max(axis=0)
        e = np.exp(zl - m)
        lse = m + np.log(e.sum(axis=0))
        safe = np.where(valid, lab, 0)
        picked = zl[safe, np.arange(zl.shape[1])]
        losses.append(((lse - picked) * valid).sum() / cnt if cnt else float("nan"))
        sm = e / e.sum(axis=0)
        sm[safe, np.arange(zl.shape[1])] -= 1.0
        if cnt:
            grad[begin:begin + n_l] = (sm * valid / cnt).reshape(n_l, H, W)
        begin += n_l
    return np.array(losses), grad


def leaf_mlp_cross_entropy(sem, weight, bias, labels, ignore_index=-100):
    """logits = Conv2d(K, C, 1)(sem) (scripts/hierslam.py:1756, :976-978), CrossEntropyLoss()(logits as [H*W, C], labels)
    (:979-982).  sem [K,H,W], weight [C,K], bias [C], labels [H,W].  Returns (loss, d_sem, d_weight, d_bias)."""
    s_ = np.asarray(sem, np.float64)
    K, H, W = s_.shape
    w, b = np.asarray(weight, np.float64).reshape(-1, K), np.asarray(bias, np.float64)
    x = s_.reshape(K, -1)
    z = w @ x + b[:, None]
    lab = np.asarray(labels).reshape(-1).astype(np.int64)
    valid = lab != ignore_index
    cnt = int(valid.sum())
    m = z.max(axis=0)
    e = np.exp(z - m)
    lse = m + np.log(e.sum(axis=0))
    safe = np.where(valid, lab, 0)
    loss = ((lse - z[safe, np.arange(z.shape[1])]) * valid).sum() / cnt if cnt else float("nan")
    g = e / e.sum(axis=0)
    g[safe, np.arange(z.shape[1])] -= 1.0
    g = g * valid / cnt if cnt else np.zeros_like(g)
    return loss, (w.T @ g).reshape(K, H, W), g @ x.T, g.sum(axis=1)
