"""Data-parallel pieces of the path: frame sharding for inference and the backbone training step.

  shard_frames           det3d/datasets/loader/sampler.py:74-96 (DistributedSampler, shuffle=False: rank::world)
  allreduce_grads_flat   det3d/core/utils/dist_utils.py:7-41 (_allreduce_coalesced: one flat bucket per dtype)
  wrap_sync_bn           det3d/torchie/apis/train.py:301 (SyncBatchNorm.convert_sync_batchnorm)
  fused_bn.fuse_for_training  the same statistics exchange inside the fused BatchNorm+add+ReLU kernels (default)
  train_step             det3d/core/utils/dist_utils.py:50-56 + trainer loop (zero_grad, backward, step)

One process per GPU; the only collective on the data path is the gradient all-reduce (RCCL over xGMI
when the backend is "nccl"; gloo in the CPU tests). The backbone holds ~2.7 M fp32 parameters
(10.8 MB): on point-to-point xGMI a ring all-reduce of that size is latency-bound, so the whole
gradient lives in ONE flat buffer (GradBuckets): a handful of contiguous slices, deepest stage first, are
reduced as soon as their gradients are complete, overlapping the remaining dgrad/wgrad kernels.
"""
import os
import time

import numpy as np
import torch


def shard_frames(n_frames, rank, world):
    """Frame indices of this rank (frame i -> rank i mod world), padded like DistributedSampler so
    every rank gets ceil(n/world) frames."""
    per = (n_frames + world - 1) // world
    idx = list(range(n_frames))
    idx += idx[: per * world - n_frames]
    return idx[rank::world][:per]


def _all_reduce_sum(t, dist, async_op=False):
    """Sum `t` over ranks in place. RCCL ("nccl") reduces device memory directly and can run asynchronously on its
    own stream; CPU rehearsal backends (gloo) stage device tensors through host memory."""
    if t.is_cuda and dist.get_backend() != "nccl":
        host = t.cpu()
        dist.all_reduce(host)
        t.copy_(host)
        return None
    return dist.all_reduce(t, async_op=async_op)


def allreduce_grads_flat(params, dist=None, world=1):
    """Average gradients over ranks through one flat fp32 bucket (det3d/core/utils/dist_utils.py:7-41). The bucket
    covers EVERY parameter in the given order -- a parameter without a gradient on this rank contributes zeros and
    receives the average -- so all ranks always reduce the same number of elements. Returns the bucket size in bytes."""
    params = list(params)
    if not params:
        return 0
    ref = next((p.grad for p in params if p.grad is not None), params[0])
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p, dtype=ref.dtype)).reshape(-1)
                      for p in params])
    if dist is not None:
        _all_reduce_sum(flat, dist)
        flat.div_(world)
        off = 0
        for p in params:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
    return flat.numel() * flat.element_size()


def _concurrent_stream(dev, tries=6):
    """A stream whose kernels really run beside the current stream's. HIP spreads its streams over a few hardware
    queues (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share a queue take turns: in a process that has
    created other streams before (captured plans, RCCL) a fresh stream lands on the current stream's queue about one
    time in four (measured: the training step then runs at the single-stream time). So candidates from torch's pool
    are probed with two ~0.5 ms spin kernels, one per stream; the first pair that takes about one kernel's time,
    not two, wins. Falls back to a plain new stream if the probe cannot run."""
    cand = torch.cuda.Stream(dev)
    try:
        main = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def spin(cycles, other=None):
            torch.cuda.synchronize(dev)
            if other is not None:
                other.wait_stream(main)
            e0.record(main)
            if other is not None:
                with torch.cuda.stream(other):
                    torch.cuda._sleep(cycles)
            torch.cuda._sleep(cycles)
            if other is not None:
                main.wait_stream(other)
            e1.record(main)
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1)

        cycles = 100_000
        spin(cycles)
        one = spin(cycles)
        if one <= 0:
            return cand
        cycles = int(min(max(cycles * 0.5 / one, 10_000), 50_000_000))      # about half a millisecond
        one = min(spin(cycles), spin(cycles))
        for _ in range(tries):
            both = min(spin(cycles, cand), spin(cycles, cand))
            if both < 1.5 * one:
                return cand
            cand = torch.cuda.Stream(dev)                                     # next stream of the pool, another queue
    except Exception:
        pass
    return cand


class _GradSink:
    """A parameter's slice of the flat gradient buffer, for backward functions that write a gradient in place instead
    of returning it (spconv's wgrad, the fused BatchNorm's parameter gradients): `view` is the destination, done()
    reports the gradient complete (what the post-accumulate hook reports for gradients that come through autograd),
    `uses` counts the parameter's uses in the current forward pass (a direct write is only valid for a single use)."""
    __slots__ = ("gb", "name", "lo", "view", "uses")

    def __init__(self, gb, name, lo, view):
        self.gb, self.name, self.lo, self.view, self.uses = gb, name, lo, view, 0

    def done(self):
        self.gb.direct += 1
        self.gb._arrived(self.name)


class GradBuckets:
    """The gradient exchange of the data-parallel step (DDP's job at det3d/torchie/apis/train.py:302-308).

    ONE flat fp32 buffer holds every gradient: each parameter's .grad is a view into it (no concatenation before and
    no copy-back after the all-reduce, and zeroing is one fill). The buffer is laid out in BACKWARD order -- the
    deepest stage first -- and cut into `n_buckets` contiguous slices. A post-accumulate hook counts the gradients
    that have arrived; when a bucket is complete its all-reduce is issued at once (asynchronously on RCCL's stream
    under the "nccl" backend), so the exchange of the deep stages overlaps the dgrad/wgrad kernels of the shallower
    ones. finish() issues whatever did not fire (parameters that got no gradient this step still take part, as
    zeros, so every rank reduces the same slices), waits, and divides by the world size.
    Bucketed and single-bucket exchanges give identical sums element by element.

    Collective ORDER across ranks. These all-reduces share the process group with the SyncBatchNorm sum exchanges of
    the backward pass, so every rank must issue them at the same points of its backward pass. A bucket fires from a
    hook only if it -- and every bucket before it -- is complete on EVERY rank: the first step fires nothing from
    hooks, and its finish() all-reduces (MIN) the number of leading buckets this rank completed; from then on exactly
    those buckets fire from hooks, on every rank at the same gradient, and the rest go out from finish(), again on
    every rank. The set of parameters that receive a gradient must therefore not change from step to step (the
    assumption DistributedDataParallel makes with find_unused_parameters=False); call reprobe() when it does.

    Direct writes: every parameter carries a `_vx_grad_sink` (its slice of the buffer); spconv's wgrad and the fused
    BatchNorm backward write their result there and report it complete, which removes autograd's accumulate add
    (88 small kernels per step of SpMiddleResNetFHD)."""

    def __init__(self, named_params, n_buckets=4, dist=None, world=1):
        named = [(n, p) for n, p in named_params if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        self.dist, self.world = dist, int(world)
        order = named[::-1]                                   # backward order: last layer's gradients arrive first
        dev, dtype = order[0][1].device, order[0][1].dtype
        total = sum(p.numel() for _, p in order)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        self.nbytes = total * self.flat.element_size()
        # contiguous buckets of about total / n_buckets elements
        n_buckets = max(1, min(int(n_buckets), len(order)))
        self.bounds, self.members = [], []
        off = start = 0
        target = total / n_buckets
        cur = []
        for i, (name, p) in enumerate(order):
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            p._vx_grad_sink = _GradSink(self, name, off, p.grad)
            off += n
            cur.append(name)
            if (off - start >= target and len(self.bounds) < n_buckets - 1) or i == len(order) - 1:
                self.bounds.append((start, off))
                self.members.append(cur)
                start, cur = off, []
        self.bucket_of = {}
        for b, names in enumerate(self.members):
            for name in names:
                self.bucket_of[name] = b
        self.params = dict(order)
        self.arrived = [0] * len(self.bounds)
        self.seen = set()
        self.direct = 0                                       # gradients written in place this step (no accumulate add)
        self.fired = [False] * len(self.bounds)
        self.next = 0                                         # next bucket to send
        self.works = []
        self.log = []                                         # (bucket, "hook" | "finish") in firing order
        # leading buckets that may fire from hooks: all of them on one rank, none until finish() has agreed on a
        # number otherwise (class docstring, "Collective ORDER")
        self.hook_limit = len(self.bounds) if (dist is None or self.world <= 1) else None
        # second stream for the weight-gradient kernels (spconv's backward puts them there): VX_WGRAD_STREAM=0 keeps
        # everything on the caller's stream
        self.side = None
        self.side_enabled = dev.type == "cuda" and os.environ.get("VX_WGRAD_STREAM", "1") != "0"
        self.side_used = False
        for name, p in order:
            p.register_post_accumulate_grad_hook(self._make_hook(name, p))

    def wgrad_stream(self, dev):
        """The stream direct gradient writes may run on (None: the caller's). Everything that reads the flat buffer --
        a bucket's all-reduce, finish(), and through it the optimizer -- first waits for it."""
        if not self.side_enabled or dev != self.flat.device:
            return None
        if self.side is None:
            self.side = _concurrent_stream(dev)
        self.side_used = True
        return self.side

    def _join_side(self):
        if self.side_used:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.side)

    def reprobe(self):
        """Forget which buckets may fire from hooks: the next step sends everything from finish() and agrees anew."""
        if self.dist is not None and self.world > 1:
            self.hook_limit = None

    def _arrived(self, name):
        """A parameter's gradient is complete in the flat buffer (hook or direct write)."""
        if name in self.seen:
            return
        self.seen.add(name)
        self.arrived[self.bucket_of[name]] += 1
        limit = self.hook_limit or 0
        # buckets go out strictly in index order, and only those that every rank completes from its hooks
        while self.next < limit and self.arrived[self.next] == len(self.members[self.next]):
            self._fire(self.next, "hook")

    def _make_hook(self, name, p):
        b = self.bucket_of[name]
        lo = sum(self.params[n].numel() for n in self.members[b][:self.members[b].index(name)]) + self.bounds[b][0]

        def hook(param):
            g = param.grad
            if g is None or g.data_ptr() != self.flat.data_ptr() + lo * self.flat.element_size():
                # something replaced the view (e.g. zero_grad(set_to_none=True)): copy in and re-attach
                if g is not None:
                    self.flat[lo:lo + param.numel()].copy_(g.reshape(-1))
                param.grad = self.flat[lo:lo + param.numel()].view_as(param)
                param._vx_grad_sink.view = param.grad
            self._arrived(name)
        return hook

    def _fire(self, b, why):
        assert b == self.next
        self.next += 1
        self.fired[b] = True
        self.log.append((b, why))
        if self.dist is not None:
            lo, hi = self.bounds[b]
            self._join_side()
            w = _all_reduce_sum(self.flat[lo:hi], self.dist, async_op=True)
            if w is not None:
                self.works.append(w)

    def zero(self):
        """Start of a step (instead of optimizer.zero_grad): one fill; the .grad views stay attached."""
        self.flat.zero_()
        self.arrived = [0] * len(self.bounds)
        self.seen = set()
        self.fired = [False] * len(self.bounds)
        self.works, self.log, self.next, self.direct = [], [], 0, 0
        self.side_used = False
        for p in self.params.values():
            p._vx_grad_sink.uses = 0

    def finish(self):
        """After backward: reduce the buckets that did not fire, wait for all of them, average."""
        self._join_side()
        if self.hook_limit is None:
            # first step under a process group: how many leading buckets did EVERY rank complete? (one tiny MIN
            # all-reduce, issued here by every rank -- after backward, hence after every SyncBatchNorm exchange)
            k = 0
            while k < len(self.bounds) and self.arrived[k] == len(self.members[k]):
                k += 1
            t = torch.tensor([k], dtype=torch.int64, device=self.flat.device if self.dist.get_backend() == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
            self.hook_limit = int(t.item())
        while self.next < len(self.bounds):
            self._fire(self.next, "finish")
        for w in self.works:
            w.wait()
        if self.dist is not None and self.world > 1:
            self.flat.div_(self.world)
        return self.nbytes


def wrap_sync_bn(module, world, dist=None):
    """SyncBatchNorm like the reference (train.py:301). Its collectives need the nccl (RCCL) backend
    for GPU tensors; under any other backend (CPU rehearsals) the per-rank BatchNorm is kept."""
    if world > 1 and dist is not None and dist.get_backend() == "nccl":
        return torch.nn.SyncBatchNorm.convert_sync_batchnorm(module)
    return module


def train_step(net, opt, mean, coords, batch, grid, dist=None, world=1, buckets=None):
    """fwd + loss = mean(dense_out^2) + bwd (libvx dgrad/wgrad) + gradient all-reduce + optimizer step.
    With `buckets` (a GradBuckets over net's parameters) the exchange overlaps backward; without, one flat
    all-reduce after backward (dist_utils.py:7-41)."""
    if buckets is not None:
        buckets.zero()
    else:
        opt.zero_grad(set_to_none=True)
    out = net(mean, coords, batch, grid)
    loss = (out * out).mean()
    loss.backward()
    if buckets is not None:
        buckets.finish()
    else:
        allreduce_grads_flat(list(net.parameters()), dist, world)
    opt.step()
    return loss.detach()


def bench_train(a, dev, rank, world, dist):
    """BASELINE.json configs[4]: backbone training step under DP, batch a.batch frames per GPU."""
    from . import models
    from .synth import GEOMETRIES, backbone_state_dict, grid_size, lidar_cloud
    from .voxel import VoxelGenerator, voxelize_batch
    rg, vs, T, mv = GEOMETRIES[a.geometry]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    net = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5),
                                       backbone_state_dict("SpMiddleResNetFHD")).to(dev).train()
    if getattr(a, "torch_bn", False):
        net = wrap_sync_bn(net, world, dist)          # stock torch BatchNorm1d / SyncBatchNorm + ReLU + add kernels
    else:
        from .fused_bn import fuse_for_training      # libvx BatchNorm(+add+ReLU) kernels; global statistics when world > 1
        net = fuse_for_training(net, sync=world > 1)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    buckets = None if getattr(a, "flat_allreduce", False) else GradBuckets(net.named_parameters(), 4, dist, world)
    frames = [torch.from_numpy(lidar_cloud(seed=rank * a.batch + b, sweeps=10)).to(dev) for b in range(a.batch)]

    def step():
        mean, coords, _ = voxelize_batch(gen, frames)
        return train_step(net, opt, mean, coords, len(frames), grid, dist, world, buckets)

    def barrier():
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[dev.index])
            else:
                dist.barrier()

    for _ in range(a.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    nparam = sum(p.numel() for p in net.parameters())
    return {
        "metric": "backbone training frames/sec (voxelize + mean VFE + SpMiddleResNetFHD fwd+bwd + grad all-reduce + Adam)",
        "value": world * a.batch * a.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.geometry} 10-sweep synthetic frames, batch {a.batch}/GPU, DP training step "
                               "(BASELINE.json configs[4]), loss = mean(out^2), SyncBN when >1 GPU, "
                               + ("torch BatchNorm/ReLU kernels" if getattr(a, "torch_bn", False) else
                                  "fused libvx BatchNorm+add+ReLU kernels")
                               + (", one flat gradient all-reduce after backward" if buckets is None else
                                  f", gradients in one flat buffer exchanged as {len(buckets.bounds)} buckets (deep stages "
                                  "first) overlapped with backward"
                                  + (", weight-gradient kernels on a second stream" if buckets.side_enabled else "")),
                   "frames_per_gpu": a.batch, "parallelism": f"dp{world}", "grad_bucket_bytes": 4 * nparam},
        "final_loss": float(loss.item()),
    }
