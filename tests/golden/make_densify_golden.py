"""Generates tests/golden/densify_prune_concat.npz: inputs and outputs of the REFERENCE's own optimizer-surgery helpers —
prune_gaussians / remove_points (utils/slam_external.py:139-188) and cat_params_to_optimizer (:121-137) — run on CPU
tensors with a torch.optim.Adam whose state is populated by real steps, imported from /root/reference in the build
container.  Only data is stored (no reference source).  Run: python tests/golden/make_densify_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from utils.slam_external import cat_params_to_optimizer, prune_gaussians  # noqa: E402

KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales", "semantic")
COLS = dict(means3D=3, rgb_colors=3, unnorm_rotations=4, logit_opacities=1, log_scales=1, semantic=26)


def make_state(P, seed, scales_cols=1, steps=3):
    g = torch.Generator().manual_seed(seed)
    cols = dict(COLS, log_scales=scales_cols)
    params = {k: torch.nn.Parameter(torch.randn(P, cols[k], generator=g)) for k in KEYS}
    with torch.no_grad():
        params["logit_opacities"].mul_(3.0)                      # sigmoid spread over (0, 1): some below any threshold
        params["log_scales"].mul_(0.7).sub_(2.0)                 # exp() around 0.13: some above 0.1 * scene_radius
    params["cam_unnorm_rots"] = torch.nn.Parameter(torch.randn(1, 4, 5, generator=g))
    params["cam_trans"] = torch.nn.Parameter(torch.randn(1, 3, 5, generator=g))
    opt = torch.optim.Adam([{"params": [v], "name": k, "lr": 1e-2} for k, v in params.items()])
    for s in range(steps):                                       # real Adam state: exp_avg, exp_avg_sq, step
        opt.zero_grad()
        loss = sum((v * torch.randn(v.shape, generator=g)).sum() for v in params.values())
        loss.backward()
        opt.step()
    variables = dict(means2D_gradient_accum=torch.rand(P, generator=g), denom=torch.rand(P, generator=g).round(),
                     max_2D_radius=torch.rand(P, generator=g) * 9, timestep=torch.randint(0, 7, (P,), generator=g).float(),
                     scene_radius=torch.tensor(1.7))
    return params, variables, opt


def dump(prefix, params, variables, opt, out):
    for k, v in params.items():
        out["%s/param/%s" % (prefix, k)] = v.detach().numpy().copy()
        st = opt.state.get([g for g in opt.param_groups if g["name"] == k][0]["params"][0], None)
        if st is not None:
            out["%s/exp_avg/%s" % (prefix, k)] = st["exp_avg"].numpy().copy()
            out["%s/exp_avg_sq/%s" % (prefix, k)] = st["exp_avg_sq"].numpy().copy()
            out["%s/step/%s" % (prefix, k)] = np.asarray(float(st["step"]))
    for k, v in variables.items():
        out["%s/var/%s" % (prefix, k)] = v.numpy().copy()


def main():
    out = {}
    # ---- prune_gaussians: (case, P, iteration, scales columns) ----
    prune_dict = dict(start_after=0, remove_big_after=0, stop_after=20, prune_every=20, removal_opacity_threshold=0.005,
                      final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=500)
    cases = [("prune_iter0", 600, 0, 1, dict(prune_dict, removal_opacity_threshold=0.3)),
             ("prune_final_aniso", 401, 20, 3, dict(prune_dict, final_removal_opacity_threshold=0.45)),
             ("prune_no_big", 300, 0, 1, dict(prune_dict, remove_big_after=5, removal_opacity_threshold=0.5)),
             ("prune_not_this_iter", 200, 7, 1, dict(prune_dict, removal_opacity_threshold=0.5)),
             ("prune_reset_opacities", 256, 40, 1, dict(prune_dict, stop_after=100, prune_every=20, reset_opacities=True,
                                                        reset_opacities_every=40, removal_opacity_threshold=0.2))]
    for name, P, it, sc, pd in cases:
        params, variables, opt = make_state(P, seed=len(name) + P, scales_cols=sc)
        dump(name + "/in", params, variables, opt, out)
        out[name + "/iter"] = np.asarray(it)
        for k, v in pd.items():
            out[name + "/prune_dict/" + k] = np.asarray(float(v))
        params, variables = prune_gaussians(params, variables, opt, it, pd)
        dump(name + "/out", params, variables, opt, out)
    # ---- cat_params_to_optimizer ----
    for name, P, M in (("cat_small", 300, 123), ("cat_empty_map", 0, 64), ("cat_nothing_new", 64, 0)):
        params, variables, opt = make_state(max(P, 1), seed=M + 3)
        if P == 0:
            params, variables, opt = make_state(1, seed=M + 3, steps=0)   # an optimizer that has not stepped: no state yet
        dump(name + "/in", params, variables, opt, out)
        g = torch.Generator().manual_seed(M + 11)
        new = {k: torch.randn(M, COLS[k], generator=g) for k in KEYS}
        for k, v in new.items():
            out["%s/new/%s" % (name, k)] = v.numpy().copy()
        params = cat_params_to_optimizer(new, params, opt)
        dump(name + "/out", params, {}, opt, out)
    np.savez_compressed(os.path.join(HERE, "densify_prune_concat.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
