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


class GradientExchange:
    """The exchange step of the keyframe-parallel mapping path with nothing copied and nothing exchanged that no rank produced.

    One optimizer step of the mapping loop (scripts/hierslam.py:1966-2057: render a keyframe, backward, Adam step) becomes, with
    N ranks: every rank renders G keyframes of the window (forward + backward each), the per-Gaussian gradients of its G keyframes
    ACCUMULATE in one bucket, ONE exchange sums the buckets over ranks, and only then may the optimizer step run:

        ex.begin_step()                       # clears the leaves' gradients, routes the backward's outputs into the bucket
        for kf in my_keyframes:               # shard_keyframes(window, rank, world)
            out = render(kf); ex.add_keyframe(out.radii); loss(out).backward()
        ex.submit(); grads = ex.reduced(step) # summed over all N * G keyframes; with depth = 1 (the default) this waits

    * **Zero copy.**  `params` maps gradient-sink names ("raster.means3D", "raster.colors_precomp", ... for tensors fed straight
      to the rasterizer; "params.means3D", "params.log_scales", ... for the fused input preparation of hsr_utils.slam_helpers:
      diff_gaussian_rasterization/_C.py set_gradient_sink) to the leaf tensors.  The sink hands the FIRST backward node that asks
      for a name a fresh view of the current bucket as its gradient output — once per step and name: a second producer of the same
      name (a second keyframe, a second node in one graph, a second backward() through a retained graph) gets nothing from the sink,
      allocates its own output, and autograd adds it INTO the leaf's `.grad` in place — which is the bucket's slice, because
      AccumulateGrad adopted the first producer's view instead of copying it.  After the last backward every `leaf.grad` is a slice
      of the bucket holding the sum over this rank's keyframes, and submit() starts the all-reduce on the bucket as it stands.
      (A leaf whose `.grad` is not the bucket's slice at submit() — a gradient assigned by hand — is copied into its slot;
      stats() counts both.)
    * **Visibility-sparse.**  A keyframe sees part of the map: rows of Gaussians with radii <= 0 in EVERY keyframe of EVERY rank
      are exact zeros everywhere — provided every gradient came from the rasterizer.  add_keyframe(radii) ORs the byte mask
      radii > 0 into the rank's mask; submit() all-reduces (max) the mask and exchanges only the rows of the union: gather into
      a compact buffer, ONE all-reduce, scatter back.  The result equals the dense all-reduce — bit for bit with two ranks, up to
      the order of the ring's fp32 additions with more.  The mask carries one more byte, "a gradient of this rank did not come
      from the sink" (another producer — a regulariser on the scales, say — whose rows need not be zero outside the union): if any
      rank sets it, every rank exchanges the bucket whole in that step.  When the union covers more than `dense_above` of the
      rows the bucket also goes whole — and after `probe_after` such steps in a row the mask itself (a P-byte all-reduce, a
      nonzero() and the host wait for both) is only exchanged every `probe_every`-th step until a probe finds the union sparse
      again.  Every rank takes these decisions from the same all-reduced bytes.  A step in which a keyframe gave no radii is dense.
    * **depth.**  1 (default): reduced(step) waits for the exchange submit() started — the gradients an optimizer step may use.
      depth = 2 is the "stale gradients" pipeline: two buckets, a bucket is waited for only when it is handed out again, so the
      exchange of step i overlaps the renders of step i + 1 — legitimate only for a caller that applies the gradients of step i
      after rendering step i + 1 (bench.py --stale-gradients); drain() at the end.
    """

    def __init__(self, params, device, depth=1, group=None, average=False, sparse=True, dense_above=0.75, probe_after=3, probe_every=16):
        if not isinstance(params, dict):
            raise TypeError("params: {gradient-sink name: leaf tensor}")
        self.names = list(params)
        self.leaves = [params[n] for n in self.names]
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.P = int(self.leaves[0].shape[0])
        for t in self.leaves:
            if t.dim() < 1 or int(t.shape[0]) != self.P:
                raise ValueError("every exchanged tensor needs one row per Gaussian")
        self.buckets = [GradientBucket([t.shape for t in self.leaves], device) for _ in range(depth)]
        self.pending = [None] * depth
        self.group, self.average, self.sparse, self.dense_above = group, average, bool(sparse), float(dense_above)
        self.step_index, self.cur = 0, None
        self.probe_after, self.probe_every = int(probe_after), int(probe_every)
        self._dense_run = 0        # consecutive steps whose union was dense (or that skipped the probe while in a dense run)
        self._vis = None           # this rank's visibility mask of the current step: uint8 [P + 1] (last byte: foreign-gradient flag)
        self._vis_complete = True  # every keyframe of the step gave its radii
        self._handed = set()
        self._prev_sink = None
        self._sink_installed = False
        self._stats = dict(steps=0, keyframes=0, zero_copy_tensors=0, copied_tensors=0, bytes_dense_equivalent=0, bytes_exchanged=0,
                           sparse_steps=0, union_rows=0, mask_bytes=0, masks_exchanged=0, foreign_dense_steps=0)

    # kept as a property: older callers (and tests) read ex.step
    @property
    def step(self):
        return self.step_index

    # ---- the sink ----
    def _sink(self, name, shape, dev):
        if self.cur is None or name not in self.names or name in self._handed:
            return None                                          # handed out at most once per step and name: see the class docstring
        i = self.names.index(name)
        b = self.buckets[self.cur]
        if tuple(shape) != b.shapes[i] or torch.device(dev) != self.device:
            return None
        if self.leaves[i].grad is not None:
            return None                                          # something already accumulated: let autograd add to it
        off = sum(b.sizes[:i])
        v = b.flat[off:off + b.sizes[i]].view(b.shapes[i])    # a NEW tensor object: nothing else references it, autograd adopts it
        self._handed.add(name)
        return v

    def _world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def begin_step(self, radii=None):
        """starts an optimizer step: picks the bucket, waits for its previous exchange, clears the leaves' gradients and routes the
        backward's gradient outputs into the bucket.  `radii`: shorthand for add_keyframe(radii) of a one-keyframe step."""
        from diff_gaussian_rasterization import _C
        if self._sink_installed:
            self.abort()
        b = self.step_index % len(self.buckets)
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        self.cur = b
        self._handed = set()
        self._vis = None
        self._vis_complete = True
        self._keyframes = 0
        for t in self.leaves:
            t.grad = None
        self._prev_sink = _C.set_gradient_sink(self._sink)
        self._sink_installed = True
        if radii is not None:
            self.add_keyframe(radii)

    def add_keyframe(self, radii):
        """between the forward and the backward of every keyframe this rank renders in the step: ORs radii > 0 into the rank's
        visibility mask (None: this keyframe's visibility is unknown — the step is exchanged dense)"""
        self._keyframes += 1
        if radii is None:
            self._vis_complete = False
            return
        if not self.sparse:
            return
        m = (radii > 0)
        if self._vis is None:
            self._vis = torch.zeros(self.P + 1, dtype=torch.uint8, device=radii.device)
        self._vis[:self.P] |= m.to(torch.uint8)

    def abort(self):
        """leaves a step without exchanging (an exception between begin_step and submit): the previous gradient sink is restored"""
        from diff_gaussian_rasterization import _C
        if self._sink_installed:
            _C.set_gradient_sink(self._prev_sink)
            self._sink_installed = False
        self.cur = None

    class _StepContext:
        def __init__(self, ex, radii):
            self.ex, self.radii = ex, radii

        def __enter__(self):
            self.ex.begin_step(self.radii)
            return self.ex

        def __exit__(self, et, ev, tb):
            if et is None:
                self.ex.submit()
            else:
                self.ex.abort()
            return False

    def step_scope(self, radii=None):
        """`with ex.step_scope(): ...renders and backwards...` = begin_step / submit, and abort() if the body raises"""
        return GradientExchange._StepContext(self, radii)

    def _probe_now(self):
        """exchange the visibility mask in this step?  Always, until `probe_after` dense steps in a row; then every `probe_every`-th"""
        return self._dense_run < self.probe_after or (self._dense_run - self.probe_after) % self.probe_every == 0

    def submit(self, radii=None):
        """call after the step's last backward(): starts the exchange of the accumulated gradients; returns the bucket index"""
        if radii is not None:
            self.add_keyframe(radii)
        self.abort_sink_only()
        b, bucket = self.cur, self.buckets[self.cur]
        foreign = False
        for i, (n, t) in enumerate(zip(self.names, self.leaves)):
            g = t.grad
            if g is not None and g.data_ptr() == bucket.views[i].data_ptr() and g.is_contiguous():
                self._stats["zero_copy_tensors"] += 1
            else:
                if g is None:
                    bucket.views[i].zero_()
                else:
                    bucket.views[i].copy_(g)
                    foreign = True          # not produced through the sink: its rows outside the visible union need not be zero
                t.grad = bucket.views[i]
                self._stats["copied_tensors"] += 1
        self.cur = None
        self.step_index += 1
        self._stats["steps"] += 1
        self._stats["keyframes"] += max(1, getattr(self, "_keyframes", 1))
        world = self._world()
        dense_bytes = bucket.flat.numel() * bucket.flat.element_size()
        self._stats["bytes_dense_equivalent"] += dense_bytes
        if world == 1:
            return b
        idx = None
        have_mask = self.sparse and self._vis is not None and self._vis_complete
        if have_mask:
            if self._probe_now():
                m = self._vis
                m[self.P] = 1 if foreign else 0
                dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)     # the rows ANY rank saw in ANY of its keyframes + the flag
                self._stats["mask_bytes"] += int(m.numel())
                self._stats["masks_exchanged"] += 1
                any_foreign = bool(m[self.P].item())
                if any_foreign:
                    self._stats["foreign_dense_steps"] += 1
                else:
                    idx = m[:self.P].nonzero(as_tuple=False).flatten()
                    if idx.numel() > self.dense_above * self.P:
                        idx = None
                        self._dense_run += 1
                    else:
                        self._dense_run = 0
            elif self._dense_run >= self.probe_after:
                self._dense_run += 1      # a skipped probe inside a dense run: stays dense
        self._vis = None
        if idx is None:
            self.pending[b] = bucket.all_reduce(group=self.group, average=self.average, async_op=True)
            self._stats["bytes_exchanged"] += dense_bytes
            return b
        U = int(idx.numel())
        if U == 0:
            # no rank saw anything: every gradient row is zero everywhere, so is the sum — no collective (the same decision on every
            # rank: the mask is the all-reduced one), and no zero-length all-reduce for the backend to trip over
            self._stats["sparse_steps"] += 1
            return b
        widths = [n // self.P for n in bucket.sizes]
        compact = torch.empty(U * sum(widths), dtype=bucket.flat.dtype, device=bucket.flat.device)
        parts, off = [], 0
        for v, wd in zip(bucket.views, widths):
            c = compact[off:off + U * wd].view(U, wd)
            torch.index_select(v.reshape(self.P, wd), 0, idx, out=c)
            parts.append(c)
            off += U * wd
        work = dist.all_reduce(compact, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending[b] = _SparsePending(work, compact, parts, [v.reshape(self.P, wd) for v, wd in zip(bucket.views, widths)], idx,
                                         1.0 / world if self.average else None)
        self._stats["bytes_exchanged"] += compact.numel() * compact.element_size()
        self._stats["sparse_steps"] += 1
        self._stats["union_rows"] += U
        return b

    def abort_sink_only(self):
        from diff_gaussian_rasterization import _C
        if self._sink_installed:
            _C.set_gradient_sink(self._prev_sink)
            self._sink_installed = False

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

    def stats(self):
        s = dict(self._stats)
        n = max(1, s["steps"])
        s["exchange_bytes_per_step"] = (s["bytes_exchanged"] + s["mask_bytes"]) / n
        s["dense_bytes_per_step"] = s["bytes_dense_equivalent"] / n
        s["keyframes_per_step"] = s["keyframes"] / n
        s["union_fraction"] = (s["union_rows"] / (s["sparse_steps"] * self.P)) if s["sparse_steps"] else None
        return s


class _SparsePending:
    """handle of a visibility-sparse exchange: wait() completes the all-reduce of the compact rows and scatters them back"""

    def __init__(self, work, compact, parts, views, idx, scale):
        self.work, self.compact, self.parts, self.views, self.idx, self.scale = work, compact, parts, views, idx, scale

    def wait(self):
        if self.work is None:
            return
        self.work.wait()
        if self.scale is not None:
            self.compact.mul_(self.scale)
        for v, c in zip(self.views, self.parts):
            v.index_copy_(0, self.idx, c)
        self.work = None
