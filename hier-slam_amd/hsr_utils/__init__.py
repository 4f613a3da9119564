"""Host-side helpers around the rasterizer path: camera construction and synthetic scenes.

These mirror caller-side conventions of the reference (utils/recon_helpers.py:4-28,
scripts/hierslam.py:361-389) so that tests and bench.py feed the rasterizer the same
shapes and layouts scripts/hierslam.py does.  Nothing here computes on the hot path.
"""
from .camera import setup_camera_tensors, setup_camera  # noqa: F401
from .synthetic import make_scene, make_upstream_grads  # noqa: F401
