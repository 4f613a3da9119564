"""Diagnostic: one configuration, chosen seeds: worst relative gradient error vs the oracle (relative to max |expected|),
forward image error and the number of pixels whose contributor count differs (alpha >= 1/255 / T < 1e-4 decisions flipped by
the last ulp of exp)."""
import sys; sys.path[:0]=['hier-slam_amd','tests']
import numpy as np, scenes
from harness import run_gpu, run_oracle
K=int(sys.argv[1]); seeds=[int(x) for x in sys.argv[2:]]
W,H,P,kind,sm=136,141,2500,"aniso",3.0
for seed in seeds:
    cam, sc, up = scenes.build(W, H, P, K, seed=seed, kind=kind, scale_mult=sm, bg=(0,0,0), behind_frac=0.0)
    og, gg, sg = run_gpu(cam, sc, up, semantic=True, variant="sr")
    oo, go, so = run_oracle(cam, sc, up, semantic=True, variant="sr")
    nc = int((sg["n_contrib"] != so.field("n_contrib")).sum())
    img = max(float(np.abs(np.asarray(og[k],np.float64)-np.asarray(oo[k],np.float64).reshape(np.asarray(og[k]).shape)).max()) for k in ("color","depth","opacity"))
    rels={k: float(np.abs(np.asarray(gg[k],np.float64)-np.asarray(go[k],np.float64).reshape(np.asarray(gg[k]).shape)).max()/max(np.abs(np.asarray(go[k])).max(),1e-30)) for k in gg if np.asarray(gg[k]).size}
    print("K", K, "seed", seed, "n_contrib mismatches", nc, "img err %.2e" % img, {k: "%.1e" % v for k,v in rels.items()})
