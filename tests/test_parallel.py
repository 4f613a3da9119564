"""CPU suite: the keyframe-parallel exchange step over gloo, world_size 2.

Each rank computes the gradients of ITS keyframe (here with the CPU oracle — tests may use it; the product
path renders on the GPU) and the bucketed all-reduce must equal the serial sum over keyframes."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _keyframe_grads(rank):
    for p in (os.path.join(ROOT, "hier-slam_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import scenes
    from harness import run_oracle
    from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
    W, H, P, K = 48, 32, 120, 4
    cam, sc, up = scenes.build(W, H, P, K, seed=1, kind="aniso", scale_mult=3.0, tilt=False, grad_seed=10 + rank)
    w2c = np.eye(4)
    w2c[0, 3] = 0.02 * rank  # a different keyframe per rank, same Gaussians
    cam2 = setup_camera_tensors(W, H, replica_intrinsics(W, H), w2c)
    _, gr, st = run_oracle(cam2, sc, up, semantic=True, threads=1)
    st.free()
    names = ("means3D", "colors_precomp", "semantics_precomp", "opacities", "scales", "rotations")
    return names, [torch.tensor(gr[n]) for n in names]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))
    from hsr_utils.parallel import GradientBucket, allreduce_gradients, shard_keyframes
    names, grads = _keyframe_grads(rank)
    params = [torch.zeros_like(g).requires_grad_(True) for g in grads]
    for p, g in zip(params, grads):
        p.grad = g.clone()
    bucket = allreduce_gradients(params)
    assert bucket.flat.numel() == sum(g.numel() for g in grads)
    # second iteration reuses the bucket
    for p, g in zip(params, grads):
        p.grad = g.clone()
    allreduce_gradients(params, bucket=bucket)
    assert shard_keyframes(range(5), rank, world) == list(range(5))[rank::world]
    if rank == 0:
        q.put([p.grad.numpy() for p in params])
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = None
    for r in range(world):
        _, g = _keyframe_grads(r)
        expect = [a.numpy() for a in g] if expect is None else [e + a.numpy() for e, a in zip(expect, g)]
    for a, b in zip(got, expect):
        assert np.allclose(a, b, rtol=1e-6, atol=1e-6)


def test_bucket_single_process_noop():
    sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))
    from hsr_utils.parallel import GradientBucket
    b = GradientBucket([(4, 3), (4, 1)], "cpu")
    b.pack([torch.ones(4, 3), None])
    assert b.all_reduce() is None and float(b.flat.sum()) == 12.0
    assert b.views[0].shape == (4, 3) and b.views[1].shape == (4, 1)
