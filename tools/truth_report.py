#!/usr/bin/env python
"""HIP and the fp32 oracle, each against the truth build of the oracle (double arithmetic on the same fp32 lists), on one
synthetic workload at its full size: prints one JSON object (tests/harness.truth_report).  GPU box only.
usage: python tools/truth_report.py [--P 500000 --K 26 --width 1200 --height 680 --kind slam|aniso]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("hier-slam_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402

import harness  # noqa: E402
from hsr_utils.camera import replica_intrinsics, setup_camera_tensors  # noqa: E402
from hsr_utils.synthetic import make_scene, make_upstream_grads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=500000)
ap.add_argument("--K", type=int, default=26)
ap.add_argument("--width", type=int, default=1200)
ap.add_argument("--height", type=int, default=680)
ap.add_argument("--kind", default="slam")
a = ap.parse_args()
k = replica_intrinsics(a.width, a.height)
cam = setup_camera_tensors(a.width, a.height, k, np.eye(4))
sc = make_scene(a.P, a.width, a.height, a.K, k, seed=0, kind=a.kind)
up = make_upstream_grads(a.width, a.height, a.K, seed=1)
rep = harness.truth_report(cam, sc, up, semantic=True)
rep["workload"] = vars(a)
print(json.dumps(rep))
