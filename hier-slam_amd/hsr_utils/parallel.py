"""Keyframe-parallel mapping step: the ONE exchange step of the data-parallel path.

Every rank holds the same Gaussian parameters, renders a different keyframe of the mapping window
(forward + backward, independent units — no collective inside the render), then the per-Gaussian
gradients are summed over ranks with a single bucketed all-reduce (RCCL over xGMI when the backend is
"nccl", gloo on CPU for tests).  The reference has no multi-GPU code (single process, `cuda:0`,
configs/replica/hierslam_semantic_run.py:8); this is new capability named by BASELINE.json's north_star.

One flat fp32 bucket (≈76 MB at P=500k, K=26) rather than one collective per tensor: xGMI is
point-to-point and per-link bound, so few large messages beat many small ones.
"""
import torch
import torch.distributed as dist


class GradientBucket:
    """Flattens a fixed list of gradient tensors into one contiguous buffer for a single all-reduce."""

    def __init__(self, shapes, device, dtype=torch.float32):
        self.shapes = [tuple(s) for s in shapes]
        self.sizes = [int(torch.Size(s).numel()) for s in self.shapes]
        self.flat = torch.zeros(sum(self.sizes), dtype=dtype, device=device)
        self.views = []
        off = 0
        for s, n in zip(self.shapes, self.sizes):
            self.views.append(self.flat[off:off + n].view(s))
            off += n

    def pack(self, grads):
        """copies each gradient (None = zeros) into its slot"""
        for v, g in zip(self.views, grads):
            if g is None:
                v.zero_()
            else:
                v.copy_(g)

    def all_reduce(self, group=None, average=False, async_op=False):
        """Sum (or mean) over the ranks of `group`, in place.  async_op=True returns a handle whose wait() completes the
        operation — the division of average=True included; None when there is nothing to exchange."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        scale = 1.0 / dist.get_world_size(group) if average else None
        if not async_op:
            if scale is not None:
                self.flat.mul_(scale)
            return None
        return _Pending(work, self.flat, scale)


class _Pending:
    """handle of an asynchronous bucket all-reduce: wait() blocks until the sum is in the bucket and applies the mean"""

    def __init__(self, work, flat, scale):
        self.work, self.flat, self.scale = work, flat, scale

    def wait(self):
        if self.work is not None:
            self.work.wait()
            if self.scale is not None:
                self.flat.mul_(self.scale)
            self.work = None


class PipelinedAllReduce:
    """`depth` gradient buckets in flight (default 2): submit() packs the gradients of step i into bucket i mod depth and
    starts its all-reduce on the communication stream; the bucket is waited for only when it is packed again (or at drain()),
    so the exchange of step i overlaps the render of step i + 1.  xGMI is per-link bound — a ring over the 76 MB bucket of
    the headline workload costs about as much as a render — which is why the exchange is taken off the critical path
    instead of being made faster.  reduced(i) returns the views of the bucket that held step i, after waiting for it."""

    def __init__(self, shapes, device, depth=2, group=None, average=False, dtype=torch.float32):
        self.buckets = [GradientBucket(shapes, device, dtype) for _ in range(depth)]
        self.pending = [None] * depth
        self.group, self.average, self.step = group, average, 0

    def submit(self, grads):
        b = self.step % len(self.buckets)
        self.step += 1
        if self.pending[b] is not None:
            self.pending[b].wait()
        self.buckets[b].pack(grads)
        self.pending[b] = self.buckets[b].all_reduce(group=self.group, average=self.average, async_op=True)
        return b

    def reduced(self, step):
        b = step % len(self.buckets)
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        return self.buckets[b].views

    def drain(self):
        for b in range(len(self.buckets)):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None


def allreduce_gradients(params, group=None, average=False, bucket=None):
    """Sums (or averages) `p.grad` of every tensor in `params` over the ranks of `group`, in place.
    Returns the bucket so callers can reuse it across iterations."""
    params = [p for p in params if p is not None]
    if bucket is None:
        bucket = GradientBucket([p.shape for p in params], params[0].device, params[0].dtype)
    bucket.pack([p.grad for p in params])
    bucket.all_reduce(group=group, average=average)
    for p, v in zip(params, bucket.views):
        if p.grad is None:
            p.grad = v.clone()
        else:
            p.grad.copy_(v)
    return bucket


def shard_keyframes(keyframe_ids, rank, world_size):
    """Round-robin assignment of the mapping window's keyframes to ranks: rank r renders ids[r::world]."""
    return list(keyframe_ids)[rank::world_size]
