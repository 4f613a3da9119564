#!/usr/bin/env python
"""Generates tests/golden/*.npz: seeded inputs + the C oracle's outputs, intermediates and gradients.

The reference (CUDA) cannot be built or run in this pipeline and ships no fixtures, so
these vectors come from oracle/hsr_oracle.c, which tests/test_oracle.py pins against an independent
float64 autograd derivation.  They are DATA (inputs and expected outputs) — no reference source.
Usage: python tests/golden/make_golden.py   (rewrites the fixtures in place)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
import scenes  # noqa: E402

CASES = {
    # name: (W, H, P, K, kind, scale_mult, semantic, bg)
    "sem_k26_aniso_64x48": (64, 48, 300, 26, "aniso", 2.5, True, (0.0, 0.0, 0.0)),
    "sem_k16_slam_80x56_bg": (80, 56, 400, 16, "slam", 3.0, True, (0.2, 0.4, 0.6)),
    "plain_mask_72x40": (72, 40, 300, 0, "aniso", 2.5, False, (0.0, 0.0, 0.0)),
}


def main():
    for name, (W, H, P, K, kind, sm, semantic, bg) in CASES.items():
        cam, sc, up = scenes.build(W, H, P, K, seed=21, kind=kind, scale_mult=sm, bg=bg, behind_frac=0.1)
        kw = dict(colors_precomp=sc["colors_precomp"], scales=sc["scales"], rotations=sc["rotations"])
        if semantic:
            kw["semantics_precomp"] = sc["semantics_precomp"]
        out, st = O.forward(cam, sc["means3D"], sc["opacities"], threads=1, **kw)
        g = dict(color=up["color"].numpy(), semantic=up["semantic"].numpy() if semantic else None, depth=up["depth"].numpy(),
                 median=up["median"].numpy(), opacity=up["opacity"].numpy())
        gr = O.backward(st, cam, sc["means3D"], g, threads=1, **kw)
        d = dict(W=W, H=H, semantic=semantic, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=cam["bg"].numpy(),
                 scale_modifier=cam["scale_modifier"], viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(),
                 sh_degree=cam["sh_degree"], campos=cam["campos"].numpy(),
                 means3D=sc["means3D"].numpy(), opacities=sc["opacities"].numpy(), colors_precomp=sc["colors_precomp"].numpy(),
                 scales=sc["scales"].numpy(), rotations=sc["rotations"].numpy(), semantics_precomp=sc["semantics_precomp"].numpy(),
                 up_color=g["color"], up_semantic=up["semantic"].numpy(), up_depth=g["depth"], up_median=g["median"],
                 up_opacity=g["opacity"], exp_num_rendered=out["num_rendered"], exp_radii=out["radii"])
        for n in ("keys", "vals", "ranges", "tiles_touched", "n_contrib"):
            d["exp_" + n] = st.field(n)
        for n in ("color", "depth", "median_depth", "opacity") + (("semantic",) if semantic else ("mask",)):
            d["exp_" + n] = out[n]
        for n in ("means3D", "means2D", "opacities", "colors_precomp", "scales", "rotations", "semantics_precomp"):
            d["exp_grad_" + n] = gr[n]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, "R =", out["num_rendered"], "visible =", int((out["radii"] > 0).sum()))
        st.free()


if __name__ == "__main__":
    main()
