"""Independent float64 torch restatement of the rasterizer's *differentiable maths*, used to pin the
C oracle (which is otherwise "parity unpinned": the reference has no fixtures).

It is written from the textbook formulation (EWA splatting: cov2D = J W Sigma W^T J^T + 0.3 I,
front-to-back alpha compositing), NOT from the oracle's code, and is differentiated by
torch.autograd instead of hand-derived chain rules.  The discrete structure (per-tile depth-sorted
lists) is taken from the oracle as data; thresholds (power>0, alpha<1/255, T<1e-4, 0.99 clamp,
median crossing) are applied as non-differentiable masks the way the reference's backward treats them
(backward.cu:585-592: the clamp passes gradient straight through; :175-176: frustum clamp zeroes the
x/y gradient only).
"""
import numpy as np
import torch


def _quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]  # (r,x,y,z), not normalised (forward.cu:127-131)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R


def dense_render(cam, p, point_list, ranges, semantic=True, sem_alpha_exact=False):
    """p: dict of float64 leaf tensors means3D, scales, rotations, opacities, colors, semantics, ndc_delta[P,2].
    Returns dict color[3,H,W], semantic[K,H,W], depth, median, opacity, mask, final_T."""
    g = (lambda k: cam[k]) if isinstance(cam, dict) else (lambda k: getattr(cam, k))
    W, H = int(g("image_width")), int(g("image_height"))
    dt = torch.float64
    V = torch.as_tensor(g("viewmatrix")).reshape(4, 4).to(dt).cpu()
    Pm = torch.as_tensor(g("projmatrix")).reshape(4, 4).to(dt).cpu()
    tanx, tany = float(g("tanfovx")), float(g("tanfovy"))
    # fp32 focal as the reference computes it (rasterizer_impl.cu:226-227)
    fx = float(np.float32(W) / (np.float32(2.0) * np.float32(tanx)))
    fy = float(np.float32(H) / (np.float32(2.0) * np.float32(tany)))
    mod = float(g("scale_modifier"))
    m = p["means3D"]
    P = m.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([m, ones], 1) @ Pm
    pw = 1.0 / (hom[:, 3] + 1e-7)
    proj = hom[:, :3] * pw[:, None]
    t = torch.cat([m, ones], 1) @ V
    tz = t[:, 2]
    depth = tz
    # covariance
    if p.get("cov3D") is not None:
        c = p["cov3D"]
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    else:
        R = _quat_to_rot(p["rotations"])
        S2 = torch.diag_embed((mod * p["scales"]) ** 2)
        Sigma = R @ S2 @ R.transpose(1, 2)
    limx, limy = 1.3 * float(np.float32(tanx)), 1.3 * float(np.float32(tany))
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx_used = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), t[:, 0])
    ty_used = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx_used) / (tz * tz), zero, fy / tz, -(fy * ty_used) / (tz * tz)], 1).reshape(-1, 2, 3)
    Wr = V[:3, :3].T  # world->camera rotation
    A = J @ Wr
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c2 = cov2[:, 1, 1] + 0.3
    det = a * c2 - b * b
    conic = torch.stack([c2 / det, -b / det, a / det], 1)
    px = ((proj[:, 0] + p["ndc_delta"][:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((proj[:, 1] + p["ndc_delta"][:, 1] + 1.0) * H - 1.0) * 0.5
    opac = p["opacities"].reshape(-1)
    col = p["colors"]
    sem = p["semantics"]
    # The reference treats final opacity as one more blended channel whose per-Gaussian "colour" is the
    # constant 1 (backward.cu:628-632, :859-864) and adds that channel's colour-gradient sum(alpha*T*dL)
    # INTO dL_dopacity, on top of the true alpha-path term.  `unit` is that constant as a leaf, so the
    # quirk term is d L / d unit; sum_i w_i * unit_i == 1 - T_final.
    unit = p["unit"]
    K = sem.shape[1] if semantic else 0
    tiles_x = (W + 15) // 16
    tiles_y = (H + 15) // 16
    out_c = torch.zeros(3, H, W, dtype=dt)
    out_s = torch.zeros(K, H, W, dtype=dt)
    out_d = torch.zeros(H, W, dtype=dt)
    out_m = torch.full((H, W), 15.0, dtype=dt)
    out_T = torch.ones(H, W, dtype=dt)
    out_o = torch.zeros(H, W, dtype=dt)
    out_mask = torch.zeros(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    pl = torch.as_tensor(np.asarray(point_list).astype(np.int64))
    for ty_ in range(tiles_y):
        for tx_ in range(tiles_x):
            r0, r1 = int(ranges[ty_ * tiles_x + tx_][0]), int(ranges[ty_ * tiles_x + tx_][1])
            y0, y1 = ty_ * 16, min(ty_ * 16 + 16, H)
            x0, x1 = tx_ * 16, min(tx_ * 16 + 16, W)
            ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
            pfx, pfy = xs.to(dt), ys.to(dt)
            T = torch.ones_like(pfx)
            done = torch.zeros_like(pfx, dtype=torch.bool)
            C = [torch.zeros_like(pfx) for _ in range(3)]
            S = [torch.zeros_like(pfx) for _ in range(K)]
            Dd = torch.zeros_like(pfx)
            Mk = torch.zeros_like(pfx)
            Op = torch.zeros_like(pfx)
            med = torch.full_like(pfx, 15.0)
            ncon = torch.zeros_like(pfx, dtype=torch.int64)
            for pos, i in enumerate(range(r0, r1)):
                gid = int(pl[i])
                dx, dy = px[gid] - pfx, py[gid] - pfy
                power = -0.5 * (conic[gid, 0] * dx * dx + conic[gid, 2] * dy * dy) - conic[gid, 1] * dx * dy
                valid = (~done) & (power <= 0)
                araw = opac[gid] * torch.exp(power)
                alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()
                valid = valid & (alpha >= 1.0 / 255.0)
                test_T = T * (1 - alpha)
                newly = valid & (test_T < 0.0001)
                done = done | newly
                valid = valid & ~newly
                w = torch.where(valid, alpha * T, torch.zeros_like(T))
                for ch in range(3):
                    C[ch] = C[ch] + w * col[gid, ch]
                # reference-as-observed: the semantic loss reaches only dL_dsemantics, never alpha
                # (backward.cu:834 reads an unwritten scratch buffer -> 0); exact mode keeps the term
                ws = w if sem_alpha_exact else w.detach()
                for ch in range(K):
                    S[ch] = S[ch] + ws * sem[gid, ch]
                Dd = Dd + w * depth[gid]
                Op = Op + w * unit[gid]
                Mk = Mk + w
                cross = valid & (T > 0.5) & (test_T < 0.5)
                med = torch.where(cross, depth[gid].expand_as(med), med)
                T = torch.where(valid, test_T, T)
                ncon = torch.where(valid, torch.full_like(ncon, pos + 1), ncon)
            for ch in range(3):
                out_c[ch, y0:y1, x0:x1] = C[ch]
            for ch in range(K):
                out_s[ch, y0:y1, x0:x1] = S[ch]
            out_d[y0:y1, x0:x1] = Dd
            out_m[y0:y1, x0:x1] = med
            out_T[y0:y1, x0:x1] = T
            out_o[y0:y1, x0:x1] = Op
            out_mask[y0:y1, x0:x1] = Mk
            n_contrib[y0:y1, x0:x1] = ncon
    return dict(color=out_c, semantic=out_s, depth=out_d[None], median=out_m[None], opacity=out_o[None],
                mask=out_mask[None], final_T=out_T, n_contrib=n_contrib)


def dense_loss_and_grads(cam, scene, grads, point_list, ranges, semantic=True, use_cov3d=False, sem_alpha_exact=False):
    """Builds float64 leaves from `scene` (dict of fp32 tensors / arrays), renders, forms
    L = sum(out * upstream) (+ T_final * bg . dL_dcolor: the background term the reference's backward
    assumes, backward.cu:641-644), and returns (outputs, grads dict)."""
    dt = torch.float64
    tt = lambda a: torch.as_tensor(np.asarray(a)).to(dt).clone().requires_grad_(True)
    p = dict(means3D=tt(scene["means3D"]), opacities=tt(scene["opacities"]), colors=tt(scene["colors_precomp"]),
             semantics=tt(scene["semantics_precomp"]) if semantic else torch.zeros(len(scene["means3D"]), 0, dtype=dt))
    if use_cov3d:
        p["cov3D"] = tt(scene["cov3D_precomp"])
    else:
        p["scales"] = tt(scene["scales"])
        p["rotations"] = tt(scene["rotations"])
    p["ndc_delta"] = torch.zeros(len(scene["means3D"]), 2, dtype=dt, requires_grad=True)
    p["unit"] = torch.ones(len(scene["means3D"]), dtype=dt, requires_grad=True)
    out = dense_render(cam, p, point_list, ranges, semantic=semantic, sem_alpha_exact=sem_alpha_exact)
    g = (lambda k: cam[k]) if isinstance(cam, dict) else (lambda k: getattr(cam, k))
    bg = torch.as_tensor(g("bg")).to(dt).cpu()
    G = {k: torch.as_tensor(np.asarray(v)).to(dt) for k, v in grads.items() if v is not None}
    L = (out["color"] * G["color"]).sum() + (out["depth"] * G["depth"]).sum() + (out["median"] * G["median"]).sum() \
        + (out["opacity"] * G["opacity"]).sum()
    L = L + (out["final_T"][None] * bg[:, None, None] * G["color"]).sum()
    if semantic and out["semantic"].numel():
        L = L + (out["semantic"] * G["semantic"]).sum()
    leaves = {k: v for k, v in p.items() if v.requires_grad}
    gr = torch.autograd.grad(L, list(leaves.values()), allow_unused=True)
    res = {k: (gv if gv is not None else torch.zeros_like(v)) for (k, v), gv in zip(leaves.items(), gr)}
    # reference quirk (see `unit` above): dL_dopacity = true alpha-path gradient + d L / d unit
    res["opacities_ref"] = res["opacities"] + res["unit"].reshape(res["opacities"].shape)
    return out, res


# ---- view-dependent colour from spherical harmonics (reference forward.cu:20-71, backward.cu:20-139) --------------------
# Independent of the oracle's (and the HIP kernels') hard-coded degree-0..3 polynomials: the basis is built from the
# general definition of the real spherical harmonics with the Condon-Shortley phase, m = -l..l inside a band,
#     Y_lm(d) = (-1)^m sqrt(2) N_lm  [d^|m|/dz^|m| P_l](z)  Re|Im (x + i y)^|m|      (m != 0;  m = 0: N_l0 P_l(z))
#     N_lm = sqrt((2l+1)/(4 pi) (l-|m|)!/(l+|m|)!)
# (Legendre coefficients from numpy.polynomial; tests/test_oracle.py checks the basis against scipy.special's complex
# harmonics.)  On the unit sphere these equal the reference's polynomials; off the sphere they differ only radially,
# which the direction normalisation projects out of every gradient.
def real_sh_basis(d, deg):
    """d: [P,3] float64 unit directions -> [P,(deg+1)^2] in the reference's coefficient order."""
    from math import factorial, pi, sqrt
    from numpy.polynomial import legendre as L
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    cols = []
    for l in range(deg + 1):
        base = L.leg2poly([0.0] * l + [1.0])              # P_l as ascending power coefficients
        for m in range(-l, l + 1):
            a = abs(m)
            co = np.polynomial.polynomial.polyder(base, a) if a else base
            pz = torch.zeros_like(z)
            for c in co[::-1]:                            # Horner
                pz = pz * z + float(c)
            n = sqrt((2 * l + 1) / (4 * pi) * factorial(l - a) / factorial(l + a))
            if m == 0:
                cols.append(n * pz)
                continue
            re, im = torch.ones_like(x), torch.zeros_like(x)
            for _ in range(a):                            # (x + i y)^a
                re, im = re * x - im * y, re * y + im * x
            cols.append(((-1) ** a) * sqrt(2.0) * n * pz * (re if m > 0 else im))
    return torch.stack(cols, 1)


def sh_colors(shs, means3D, campos, deg):
    """float64 colours [P,3] = max(sum_lm Y_lm(dir) sh_lm + 0.5, 0), dir = normalise(mean - campos)."""
    d = means3D - campos.reshape(1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    nb = (deg + 1) ** 2
    Y = real_sh_basis(d, deg)
    rgb = (Y[:, :, None] * shs[:, :nb, :]).sum(1) + 0.5
    return torch.clamp(rgb, min=0.0)    # autograd: zero gradient where clamped (backward.cu:33-38)


def dense_loss_and_grads_sh(cam, scene, shs, deg, grads, point_list, ranges, semantic=True):
    """as dense_loss_and_grads with colours from SH coefficients: returns (outputs, grads incl. 'shs'; the means3D
    gradient includes the view-direction term)."""
    dt = torch.float64
    tt = lambda a: torch.as_tensor(np.asarray(a)).to(dt).clone().requires_grad_(True)
    g = (lambda k: cam[k]) if isinstance(cam, dict) else (lambda k: getattr(cam, k))
    P = len(scene["means3D"])
    p = dict(means3D=tt(scene["means3D"]), opacities=tt(scene["opacities"]), shs=tt(shs), scales=tt(scene["scales"]),
             rotations=tt(scene["rotations"]),
             semantics=tt(scene["semantics_precomp"]) if semantic else torch.zeros(P, 0, dtype=dt))
    p["ndc_delta"] = torch.zeros(P, 2, dtype=dt, requires_grad=True)
    p["unit"] = torch.ones(P, dtype=dt, requires_grad=True)
    q = dict(p)
    q["colors"] = sh_colors(p["shs"], p["means3D"], torch.as_tensor(g("campos")).to(dt).cpu(), deg)
    out = dense_render(cam, q, point_list, ranges, semantic=semantic)
    bg = torch.as_tensor(g("bg")).to(dt).cpu()
    G = {k: torch.as_tensor(np.asarray(v)).to(dt) for k, v in grads.items() if v is not None}
    Lz = (out["color"] * G["color"]).sum() + (out["depth"] * G["depth"]).sum() + (out["median"] * G["median"]).sum() \
        + (out["opacity"] * G["opacity"]).sum() + (out["final_T"][None] * bg[:, None, None] * G["color"]).sum()
    if semantic and out["semantic"].numel():
        Lz = Lz + (out["semantic"] * G["semantic"]).sum()
    leaves = {k: v for k, v in p.items() if v.requires_grad}
    gr = torch.autograd.grad(Lz, list(leaves.values()), allow_unused=True)
    res = {k: (gv if gv is not None else torch.zeros_like(v)) for (k, v), gv in zip(leaves.items(), gr)}
    res["opacities_ref"] = res["opacities"] + res["unit"].reshape(res["opacities"].shape)
    res["colors_value"] = q["colors"].detach()
    return out, res
