"""Inference engine for the sparse backbones: same results as running the nn.Module in eval mode,
with BatchNorm1d(eval) + bias + ReLU + residual folded into the conv kernel's epilogue
(scn.py:82-98 collapses from 2 GEMM sets + 4 elementwise passes to 2 launches per block) and all
rulebooks of a frame built before the feature pass (they depend on coordinates only).
"""
import contextlib
import gc
import os

import numpy as np
import torch
from torch import nn

import spconv.pytorch as spconv

from . import ops
from .models import SparseBasicBlock


def _fold_bn(bn, dev):
    g = bn.weight.detach().double()
    b = bn.bias.detach().double()
    inv = torch.rsqrt(bn.running_var.detach().double() + bn.eps)
    scale = g * inv
    shift = b - bn.running_mean.detach().double() * scale
    return scale.float().to(dev).contiguous(), shift.float().to(dev).contiguous()


class _Step:
    __slots__ = ("kind", "w", "w32", "wpack", "bias", "scale", "shift", "relu", "key", "ksize", "stride", "padding", "residual")


def _compile_sequential(seq, dev, dtype):
    """SparseSequential -> list of fused conv steps."""
    mods = list(seq._modules.values())
    steps = []
    i = 0

    def conv_step(conv, bn, relu, residual):
        s = _Step()
        s.kind = "subm" if conv.subm else "conv"
        s.w = conv.weight.detach().to(dev, dtype).contiguous()
        # fp32 copy for layers that may be fed fp32 features while the engine runs bf16 (the voxelizer's means)
        s.w32 = conv.weight.detach().to(dev, torch.float32).contiguous() if dtype != torch.float32 else s.w
        s.wpack = None   # ops.pack_weight(s.w) would select the streaming-W kernel (slower end to end)
        s.bias = conv.bias.detach().float().to(dev).contiguous() if conv.bias is not None else None
        s.scale, s.shift = _fold_bn(bn, dev) if bn is not None else (None, None)
        s.relu, s.key, s.residual = relu, conv.indice_key, residual
        s.ksize, s.stride, s.padding = conv.kernel_size, conv.stride, conv.padding
        return s

    while i < len(mods):
        m = mods[i]
        if isinstance(m, spconv.SparseConvolution):
            bn = relu = None
            j = i + 1
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d):
                bn = mods[j]
                j += 1
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                relu = True
                j += 1
            steps.append(conv_step(m, bn, bool(relu), None))
            i = j
        elif isinstance(m, SparseBasicBlock):
            if m.downsample is not None:
                raise NotImplementedError("downsample blocks are not used by the VISTA backbones")
            steps.append(conv_step(m.conv1, m.bn1, True, "save"))   # remembers its input as identity
            steps.append(conv_step(m.conv2, m.bn2, True, "add"))
            i += 1
        else:
            raise NotImplementedError(f"cannot fuse a bare {type(m).__name__} into the plan")
    return steps


class _State:
    def __init__(self, feat, idx, shape, batch, cache, rank=None):
        self.feat, self.idx, self.shape, self.batch, self.cache = feat, idx, shape, batch, cache
        self.rank = rank  # rank table of idx when idx came out of a strided conv


def _run(steps, st, geom_only=False, products=None):
    st = _State(st.feat, st.idx, st.shape, st.batch, st.cache, st.rank)
    identity = None
    for s in steps:
        if s.kind == "subm":
            rb = st.cache.get((s.key, id(st.idx)))
            if rb is None:
                rb = ops.rulebook_subm(st.idx, st.batch, st.shape, s.ksize, rank=st.rank)
                st.cache[(s.key, id(st.idx))] = rb
        else:
            rb = st.cache.get(("conv", id(st.idx), tuple(s.ksize), tuple(s.stride), tuple(s.padding)))
            if rb is None:
                rb = ops.rulebook_conv(st.idx, st.batch, st.shape, s.ksize, s.stride, s.padding)
                st.cache[("conv", id(st.idx), tuple(s.ksize), tuple(s.stride), tuple(s.padding))] = rb
        if s.residual == "save":
            identity = st.feat
        res = identity if s.residual == "add" else None
        if not geom_only:
            w = s.w
            if st.feat.dtype != w.dtype:
                # first layer: the voxelizer's fp32 means (Cin=5) run through the fp32 kernel
                w = s.w32
            st.feat = ops.conv_fwd(st.feat, w, rb, bias=s.bias, scale=s.scale, shift=s.shift,
                                   residual=res, relu=s.relu, wpack=s.wpack if w is s.w else None, products=products)
            if st.feat.dtype != s.w.dtype:
                st.feat = st.feat.to(s.w.dtype)
        if s.kind == "conv":
            st.idx, st.shape, st.rank = rb.out_indices, rb.out_shape, rb.out_rank
    return st


def _run_dn(steps, st, caps, counts, slot, err, geom_only=False, mark=None, products=None):
    """_run with device-resident row counts (vista_amd.ops_dn): no host sync anywhere. `caps[j]` / `counts[slot + j]`
    = output capacity / count slot of the j-th strided conv met (in call order over the whole plan).
    geom_only: build the rulebooks only (they depend on coordinates alone); `mark(rb)` is called on every rulebook
    built, and a rulebook that carries a `ready` event (built on another stream) is waited for before its first use."""
    from . import ops_dn
    st = _State(st.feat, st.idx, st.shape, st.batch, st.cache, st.rank)
    st.n = st.cache["n", id(st.idx)]
    identity = None
    for s in steps:
        if s.kind == "subm":
            key = (s.key, id(st.idx))
            rb = st.cache.get(key)
            if rb is None:
                rb = ops_dn.rulebook_subm_dn(st.idx, st.n, st.batch, st.shape, s.ksize, rank=st.rank)
                st.cache[key] = rb
                if mark is not None:
                    mark(rb)
        else:
            key = ("conv", id(st.idx), tuple(s.ksize), tuple(s.stride), tuple(s.padding))
            rb = st.cache.get(key)
            if rb is None:
                j = st.cache["next_conv"]
                st.cache["next_conv"] = j + 1
                rb = ops_dn.rulebook_conv_dn(st.idx, st.n, st.batch, st.shape, s.ksize, s.stride, s.padding,
                                             caps[j], counts[slot + j:slot + j + 1], err)
                st.cache[key] = rb
                st.cache["n", id(rb.out_indices)] = rb.n_out
                if mark is not None:
                    mark(rb)
        if not geom_only and getattr(rb, "ready", None) is not None:
            torch.cuda.current_stream(st.idx.device).wait_event(rb.ready)
            rb.ready = None
        if s.residual == "save":
            identity = st.feat
        res = identity if s.residual == "add" else None
        if not geom_only:
            w = s.w
            if st.feat.dtype != w.dtype:
                w = s.w32
            st.feat = ops_dn.conv_fwd_dn(st.feat, w, rb, bias=s.bias, scale=s.scale, shift=s.shift, residual=res,
                                         relu=s.relu, products=products)
            if st.feat.dtype != s.w.dtype:
                st.feat = st.feat.to(s.w.dtype)
        if s.kind == "conv":
            st.idx, st.shape, st.rank, st.n = rb.out_indices, rb.out_shape, rb.out_rank, rb.n_out
    return st


class _Compiled:
    """The fused step lists of one backbone. Shared by an engine, its eager twin and its plans, and holding no
    reference back to any of them: a plan never keeps its engine alive (round 2's FramePlan.engine <-> FusedBackbone._plans
    cycle left abandoned graphs to the cyclic collector, see _capture_guard)."""
    __slots__ = ("kind", "mode", "middle", "bev", "rv")

    def __init__(self, kind, mode, middle, bev=None, rv=None):
        self.kind, self.mode, self.middle, self.bev, self.rv = kind, mode, middle, bev, rv


# Stream capture and object teardown do not mix. On ROCm ~CUDAGraph() calls hipDeviceSynchronize() (and releases the
# graph's private pool); under an active capture that call is refused, the C++ destructor throws, and the process
# dies with SIGABRT. torch >= 2.9 no longer runs gc.collect() when a capture begins (torch.compiler.config
# .force_cudagraph_gc is off), so a graph that is only reachable through a reference cycle is destroyed whenever
# the cyclic collector happens to run -- in round 2 that was in the middle of the next engine's capture
# (GPUTEST_r02: abort inside FramePlan.capture of the second STRESS engine). Hence: plans are closed explicitly and
# deterministically (FramePlan.close), the collector runs BEFORE a capture begins and is switched off for its duration,
# and a close() requested while a capture is open is deferred until it has ended.
_capture_depth = 0
_deferred_close = []


@contextlib.contextmanager
def _capture_guard():
    global _capture_depth
    gc.collect()                       # anything holding a graph or a pool goes now, outside the capture
    was_enabled = gc.isenabled()
    gc.disable()
    _capture_depth += 1
    try:
        yield
    finally:
        _capture_depth -= 1
        if was_enabled:
            gc.enable()
        if _capture_depth == 0:
            while _deferred_close:
                _deferred_close.pop().close()


class FramePlan:
    """Voxelization + backbone for `n_frames` clouds without one host synchronisation inside the step.

    Every buffer has a fixed capacity and every row count stays on the device (the "_dn" entry points). The eager
    path waits for the GPU five times per step (the voxel count and the output row count of each strided conv) and
    the GPU idles while the host catches up after each wait; here the ~120 launches of a step are either
      "queued": issued back to back on the caller's stream (the host runs ahead of the GPU), or
      "graph" : recorded once in a HIP graph and replayed with a single graph launch (no Python dispatch at all).
    After the step the counts come back in one small copy; a count above its capacity means rows were dropped, and
    the caller (FusedBackbone.forward_frames) re-plans with larger capacities. Results are bitwise those of the eager
    path (tests/test_x_gpu_plan.py)."""

    COUNT_SLOTS = 16     # strided convs a plan may hold

    def __init__(self, compiled, gen, n_frames, grid, cap_points, cap_rows, caps, dev, products=None):
        self.products = products     # fp32 matrix arithmetic of every conv call of this plan (flags of vx_spconv_fwd_dn)
        self.engine, self.gen, self.n_frames, self.grid = compiled, gen, int(n_frames), list(grid)   # engine: a _Compiled
        self.cap_points, self.cap_rows, self.caps = int(cap_points), int(cap_rows), [int(c) for c in caps]
        if len(self.caps) > self.COUNT_SLOTS:
            raise ValueError("too many strided convolutions for one plan")
        self.dev = dev
        nf = self.n_frames
        self.slot = 2 + nf                                   # counts: [err, total voxels, per frame ..., conv outputs ...]
        self.counts = torch.zeros((self.slot + self.COUNT_SLOTS,), dtype=torch.int64, device=dev)
        self.h_counts = torch.zeros_like(self.counts, device="cpu").pin_memory()
        self.points = torch.zeros((self.cap_points, 5), dtype=torch.float32, device=dev)
        self.offsets = torch.zeros((nf + 1,), dtype=torch.int64, device=dev)
        self.h_offsets = torch.zeros((nf + 1,), dtype=torch.int64).pin_memory()
        self.graph = None
        self.graph_geo = self.graph_feat = None    # capture_pair(): geometry and feature pass as two graphs
        self._geo = None
        self.out = None
        self.closed = False
        self.limit = None          # diagnostic (tools/graph_probe.py): stop after this many steps of the middle stack
        # Rulebooks depend on coordinates only: they are built on a second stream (a parallel branch of the captured
        # graph) while the first stage's convolutions run, each handed over through an event. Same kernels on the same
        # inputs: outputs are unchanged. VX_GEOM_OVERLAP=0 keeps everything on one stream.
        self.geom_stream = torch.cuda.Stream(dev) if os.environ.get("VX_GEOM_OVERLAP", "1") != "0" else None
        self._events, self._next_event = [], 0

    def _geometry_ahead(self, st, args):
        """All rulebooks of the plan on the geometry stream, in the order the feature pass meets them."""
        e = self.engine
        main = torch.cuda.current_stream(self.dev)
        side = self.geom_stream
        side.wait_stream(main)                     # after the voxelizer

        def mark(rb):
            # events come from the plan's own pool (created before any capture, re-used by every pass)
            if self._next_event == len(self._events):
                self._events.append(torch.cuda.Event())
            rb.ready = self._events[self._next_event]
            self._next_event += 1
            rb.ready.record(side)
        self._next_event = 0
        with torch.cuda.stream(side):
            mid = _run_dn(e.middle, st, *args, geom_only=True, mark=mark)
            if e.kind != "SpMiddleResNetFHD":
                _run_dn(e.bev, mid, *args, geom_only=True, mark=mark)
                _run_dn(e.rv, mid, *args, geom_only=True, mark=mark)

    def _forward(self):
        from . import ops_dn
        e, g = self.engine, self.gen
        # the out-of-grid error word is part of the step: cleared here (a kernel node), so one bad frame does not
        # poison every later call on this plan
        ops.fill_(self.counts[0:1], 0)
        mean, coords, _ = ops_dn.voxelize_batch_dn(self.points, self.offsets, self.n_frames, g.voxel_size,
                                                   g.point_cloud_range, g.max_num_points_per_voxel, self.cap_rows,
                                                   self.counts[1:2 + self.n_frames])
        sparse_shape = [int(v) for v in (np.array(self.grid[::-1]) + [1, 0, 0])]
        cache = {"next_conv": 0, ("n", id(coords)): self.counts[1:2]}
        st = _State(mean, coords, sparse_shape, self.n_frames, cache)
        args = (self.caps, self.counts, self.slot, self.counts[0:1])
        from . import ops_dn as od
        if self.limit is not None:
            mid = _run_dn(e.middle[:self.limit], st, *args, products=self.products)
            return mid.feat, mid.idx
        if self.geom_stream is not None:
            self._geometry_ahead(st, args)
        try:
            return self._features(st, args, od)
        finally:
            if self.geom_stream is not None:       # join (every rulebook was waited for; this closes the branch)
                torch.cuda.current_stream(self.dev).wait_stream(self.geom_stream)

    def _features(self, st, args, od):
        e = self.engine
        pr = dict(products=self.products)
        mid = _run_dn(e.middle, st, *args, **pr)
        if e.kind == "SpMiddleResNetFHD":
            d = od.to_dense_dn(mid.feat, mid.idx, mid.n, mid.batch, mid.shape)
            n, c, dd, h, w = d.shape
            return d.view(n, c * dd, h, w) if e.mode == "bev" else d
        b = _run_dn(e.bev, mid, *args, **pr)
        bev = od.to_dense_dn(b.feat, b.idx, b.n, b.batch, b.shape)
        n, c, d, h, w = bev.shape
        bev = bev.view(n, c * d, h, w)
        r = _run_dn(e.rv, mid, *args, **pr)
        rv = od.to_dense_dn(r.feat, r.idx, r.n, r.batch, r.shape)
        n, c, d, h, w = rv.shape
        rv = rv.permute(0, 1, 4, 3, 2).contiguous().view(n, c * w, h, d)
        return bev, rv

    def _geometry(self):
        """Everything of a step that depends on the points alone, on the current stream: error word, voxelizer + mean
        VFE, every rulebook. -> (state, args) for _features, which then finds all its rulebooks in the cache."""
        from . import ops_dn
        e, g = self.engine, self.gen
        ops.fill_(self.counts[0:1], 0)
        mean, coords, _ = ops_dn.voxelize_batch_dn(self.points, self.offsets, self.n_frames, g.voxel_size,
                                                   g.point_cloud_range, g.max_num_points_per_voxel, self.cap_rows,
                                                   self.counts[1:2 + self.n_frames])
        sparse_shape = [int(v) for v in (np.array(self.grid[::-1]) + [1, 0, 0])]
        cache = {"next_conv": 0, ("n", id(coords)): self.counts[1:2]}
        st = _State(mean, coords, sparse_shape, self.n_frames, cache)
        args = (self.caps, self.counts, self.slot, self.counts[0:1])
        mid = _run_dn(e.middle, st, *args, geom_only=True)
        if e.kind != "SpMiddleResNetFHD":
            _run_dn(e.bev, mid, *args, geom_only=True)
            _run_dn(e.rv, mid, *args, geom_only=True)
        return st, args

    def capture_pair(self):
        """The step as TWO graphs sharing one memory pool: `graph_geo` (voxelizer + rulebooks) and `graph_feat` (the
        convolutions + dense()), for FramePipeline2, which runs the geometry of the next step beside the convolutions
        of this one. Same kernels on the same operands as capture()'s single graph: the outputs are the same bits."""
        from . import ops_dn as od
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side), torch.no_grad():
            st, args = self._geometry()
            self._features(st, args, od)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        del side, st, args
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with _capture_guard():
            ops.L.trace("capture (geometry) begin")
            with torch.cuda.graph(g1, capture_error_mode="thread_local"), torch.no_grad():
                self._geo = self._geometry()
            ops.L.trace("capture (features) begin")
            with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"), torch.no_grad():
                self.out = self._features(*self._geo, od)
            ops.L.trace("capture end")
        self.graph_geo, self.graph_feat = g1, g2
        return self

    def capture(self):
        """One un-captured pass (loads kernels, sets function attributes, fills the allocator), then the capture."""
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side), torch.no_grad():
            self._forward()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        del side
        graph = torch.cuda.CUDAGraph()
        with _capture_guard():
            ops.L.trace("capture begin")
            with torch.cuda.graph(graph, capture_error_mode="thread_local"), torch.no_grad():
                self.out = self._forward()
            ops.L.trace("capture end")
        self.graph = graph
        return self

    def close(self):
        """Destroy the captured graph and give the plan's buffers back -- NOW, never at collector time, and never
        inside a capture (then it is deferred to the capture's end). Idempotent; the plan cannot run afterwards."""
        if _capture_depth > 0:
            if self not in _deferred_close:
                _deferred_close.append(self)
            return
        graphs = [g for g in (self.graph_feat, self.graph_geo, self.graph) if g is not None]
        self.graph = self.graph_geo = self.graph_feat = None
        self.out = self._geo = None
        if graphs:
            torch.cuda.synchronize(self.dev)       # no replay of them in flight
            for graph in graphs:
                graph.reset()                      # hipGraphExecDestroy + release of its private pool
            del graphs
        self.points = self.offsets = self.counts = None
        self.closed = True

    def __del__(self):
        try:
            if not getattr(self, "closed", True):
                self.close()
        except Exception:
            pass

    def stage(self, frames):
        """Copy the clouds into the plan's input buffer and upload their point ranges (stream-ordered, no sync)."""
        off = 0
        self.h_offsets[0] = 0
        for i, f in enumerate(frames):
            n = f.shape[0]
            self.points[off:off + n].copy_(f, non_blocking=True)
            off += n
            self.h_offsets[i + 1] = off
        self.offsets.copy_(self.h_offsets, non_blocking=True)

    def replay(self):
        if self.closed:
            raise ops.L.VxError("FramePlan: replay() after close()")
        if self.graph is not None:
            self.graph.replay()
        else:                       # "queued": the same launch sequence straight onto the current stream
            with torch.no_grad():
                self.out = self._forward()
        self.h_counts.copy_(self.counts, non_blocking=True)

    def overflow(self):
        """After a stream synchronisation: None if every count fits, else the capacities this input needs."""
        h = self.h_counts.tolist()
        if h[0]:
            raise ops.L.VxError("a voxel coordinate lies outside batch x grid (vx_rulebook_conv_dn error word)")
        need_rows, need = h[1], h[self.slot:self.slot + len(self.caps)]
        if need_rows <= self.cap_rows and all(n <= c for n, c in zip(need, self.caps)):
            return None
        return need_rows, need


class FusedBackbone:
    """Eval-mode forward of SpMiddleResNetFHD / MultiViewBackbone with fused epilogues.

    __call__(voxel_features, coors, batch_size, input_shape) is the reference's backbone signature (scn.py:367,
    multiview.py:88) on per-layer launches. forward_frames(generator, clouds, grid) runs voxelization + backbone from
    raw clouds; unless the engine was built with eager=True it goes through a captured FramePlan."""

    HEADROOM = 1.125      # capacity = observed count x this, rounded up to 1024 rows

    def __init__(self, module, dtype=torch.float32, eager=False, dispatch=None, f32_products=None):
        dev = next(module.parameters()).device
        # fp32 matrix arithmetic of this engine's conv calls (0 / 6 / 9, include/vx_abi.h), carried in every call's
        # flags; None = whatever the library default is when a step is issued (then part of the plan key)
        if f32_products not in (None, 0, 6, 9):
            raise ValueError("f32_products must be None, 0, 6 or 9")
        self.f32_products = f32_products
        self.dev = dev
        self.kind = type(module).__name__
        self.dtype = dtype
        self.eager = bool(eager)
        # how forward_frames issues a step: "queued" (sync-free launches on the stream) or "graph" (captured HIP graph)
        self.dispatch = dispatch or os.environ.get("VX_DISPATCH", "graph")
        if self.dispatch not in ("queued", "graph"):
            raise ValueError("dispatch must be 'queued' or 'graph'")
        self.mode = getattr(module, "mode", "bev")
        self.middle = _compile_sequential(module.middle_conv, dev, dtype)
        self.bev = self.rv = None
        if self.kind == "MultiViewBackbone":
            self.bev = _compile_sequential(module.bev_conv, dev, dtype)
            self.rv = _compile_sequential(module.rv_conv, dev, dtype)
        elif self.kind != "SpMiddleResNetFHD":
            raise TypeError(f"unsupported backbone {self.kind}")
        self._compiled = _Compiled(self.kind, self.mode, self.middle, self.bev, self.rv)
        self._plans = {}
        self._twin = None

    def close(self):
        """Destroy every captured plan of this engine (graphs, private pools, buffers). Called when the engine is
        dropped; call it yourself to release the memory earlier."""
        plans, self._plans = list(getattr(self, "_plans", {}).values()), {}
        for p in plans:
            p.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dispatch_name(self):
        if self.eager:
            return "eager per-layer launches (5 host syncs per step)"
        return ("one captured HIP graph per step" if self.dispatch == "graph" else
                "sync-free queued launches") + " (device-resident row counts, one count check per step)"

    def eager_twin(self):
        """The same compiled steps on per-layer launches from the host (each C-ABI call can then be bracketed by
        events; also the reference the captured plan is tested against)."""
        if self.eager:
            return self
        if self._twin is None:
            t = object.__new__(FusedBackbone)
            t.__dict__.update(self.__dict__)
            t.eager, t._plans, t._twin = True, {}, None
            self._twin = t
        return self._twin

    # ---- captured path ------------------------------------------------------------------------------------
    @staticmethod
    def _cap(n, headroom):
        return max(1024, int(-(-int(n * headroom) // 1024) * 1024))

    def _probe(self, gen, frames, grid):
        """Eager geometry pass over these clouds -> (voxel count, [output rows of every strided conv in plan order])."""
        from .voxel import voxelize_batch
        mean, coords, _ = voxelize_batch(gen, frames)
        sparse_shape = [int(v) for v in (np.array(grid[::-1]) + [1, 0, 0])]
        st = _State(mean, coords.int().contiguous(), sparse_shape, len(frames), {})
        # _run visits strided convs in order and caches each under a ("conv", ...) key: dict order = plan order
        geo = _run(self.middle, st, geom_only=True)
        if self.kind == "MultiViewBackbone":
            _run(self.bev, geo, geom_only=True)
            _run(self.rv, geo, geom_only=True)
        need = [rb.n_out for key, rb in st.cache.items() if key[0] == "conv"]
        return int(coords.shape[0]), need

    def _products(self):
        return self.f32_products if self.f32_products is not None else int(ops.L.lib().vx_spconv_get_f32_products())

    def _build_plan(self, gen, frames, grid, old=None, products=None, pair=False, floors=()):
        """Capacities from an eager geometry pass over these clouds (exact counts, with its host syncs: paid once
        per plan), never below those of the plan being replaced (`old`, closed here) nor of the plans in `floors`."""
        need_rows, need = self._probe(gen, frames, grid)
        h = self.HEADROOM
        points = sum(int(f.shape[0]) for f in frames)
        cap_points, cap_rows, caps = self._cap(points, h), self._cap(need_rows, h), [self._cap(n, h) for n in need]
        for src in (old, *floors):
            if src is not None and len(src.caps) == len(caps):
                cap_points, cap_rows = max(cap_points, src.cap_points), max(cap_rows, src.cap_rows)
                caps = [max(a, b) for a, b in zip(caps, src.caps)]
        if old is not None:
            old.close()                # its graph and pool go before the new plan allocates and captures
        plan = FramePlan(self._compiled, gen, len(frames), grid, cap_points, cap_rows, caps, frames[0].device,
                         products=products)
        plan.stage(frames)
        if pair:
            return plan.capture_pair()
        return plan.capture() if self.dispatch == "graph" else plan

    def forward_frames(self, gen, frames, grid):
        """Voxelize + mean VFE + backbone for a list of CUDA clouds f32[N_i, 5]. The captured path returns tensors
        OWNED BY THE PLAN: they are overwritten by the next call with the same number of frames."""
        if self.eager:
            from .voxel import voxelize_batch
            mean, coords, _ = voxelize_batch(gen, frames)
            return self(mean, coords, len(frames), grid)
        if any(f.shape[0] > gen._max_voxels for f in frames) or not 1 <= len(frames) <= 64:
            return self.eager_twin().forward_frames(gen, frames, grid)     # max_voxels truncation: per-frame calls
        # the fp32 matrix arithmetic is an argument of every conv call of a plan (never read from the process default
        # at replay time), hence part of the key
        products = self._products()
        key = (id(gen), len(frames), tuple(grid), products)
        plan = self._plans.get(key)
        points = sum(int(f.shape[0]) for f in frames)
        if plan is None or points > plan.cap_points:
            plan = self._plans[key] = self._build_plan(gen, frames, grid, plan, products)
        stream = torch.cuda.current_stream(plan.dev)
        for attempt in range(2):
            plan.stage(frames)
            plan.replay()
            stream.synchronize()
            if plan.overflow() is None:
                return plan.out
            # some count exceeded its capacity (rows were dropped, so the counts after it are not reliable either):
            # size a new plan from an exact geometry pass over THIS input and run again
            plan = self._plans[key] = self._build_plan(gen, frames, grid, plan, products)
        raise ops.L.VxError("FramePlan: a plan sized from this very input overflowed")

    # ---- per-layer path ------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, voxel_features, coors, batch_size, input_shape, out=None):
        sparse_shape = [int(v) for v in (np.array(input_shape[::-1]) + [1, 0, 0])]
        st = _State(voxel_features, coors.int().contiguous(), sparse_shape, int(batch_size), {})
        # geometry pass first: every host sync (output row counts of the strided convs) happens before the
        # feature pass, which then queues its ~25 launches without waiting on the GPU
        geo = _run(self.middle, st, geom_only=True)
        if self.kind == "MultiViewBackbone":
            _run(self.bev, geo, geom_only=True)
            _run(self.rv, geo, geom_only=True)
        pr = dict(products=self._products())
        mid = _run(self.middle, st, **pr)
        if self.kind == "SpMiddleResNetFHD":
            d = ops.to_dense(mid.feat, mid.idx, mid.batch, mid.shape, out=out)
            n, c, dd, h, w = d.shape
            return d.view(n, c * dd, h, w) if self.mode == "bev" else d
        if out is not None:
            raise NotImplementedError("out= is only supported for SpMiddleResNetFHD")
        b = _run(self.bev, mid, **pr)
        bev = ops.to_dense(b.feat, b.idx, b.batch, b.shape)
        n, c, d, h, w = bev.shape
        bev = bev.view(n, c * d, h, w)
        r = _run(self.rv, mid, **pr)
        rv = ops.to_dense(r.feat, r.idx, r.batch, r.shape)
        n, c, d, h, w = rv.shape
        rv = rv.permute(0, 1, 4, 3, 2).contiguous().view(n, c * w, h, d)
        return bev, rv


class FramePipeline:
    """Voxelization + backbone for a list of frames, split into `groups` sub-batches that run on
    alternating HIP streams. Frames are independent, so the sub-batches share nothing; while one
    group's host syncs (voxel / output-row counts) and small rulebook kernels are in progress, the other
    group's sparse convs keep the CUs busy, and one kernel's last partial round of workgroups overlaps the
    next kernel's first. Results are bitwise identical to the single-batch call (every output row's
    accumulation order is independent of where its tile sits)."""

    def __init__(self, generator, engine, groups=2):
        self.gen, self.engine, self.groups = generator, engine, max(1, int(groups))
        self.streams = None

    def __call__(self, frames, grid, join=True):
        """-> dense BEV tensor [len(frames), C*D, H, W]. join=False leaves the side streams un-joined so
        that consecutive calls pipeline; the caller then synchronises (torch.cuda.synchronize()) before use."""
        from .voxel import voxelize_batch
        dev = frames[0].device
        if self.groups == 1 or len(frames) < 2:
            mean, coords, _ = voxelize_batch(self.gen, frames)
            return self.engine(mean, coords, len(frames), grid)
        if self.engine.kind != "SpMiddleResNetFHD":
            raise NotImplementedError("FramePipeline drives SpMiddleResNetFHD")
        if self.streams is None:
            # first call: single stream, learns the output geometry
            mean, coords, _ = voxelize_batch(self.gen, frames)
            first = self.engine(mean, coords, len(frames), grid)
            self._shape, self._dtype = tuple(first.shape[1:]), first.dtype
            self.streams = [torch.cuda.Stream(dev) for _ in range(self.groups)]
            return first
        main = torch.cuda.current_stream(dev)
        per = -(-len(frames) // self.groups)
        out = torch.empty((len(frames), *self._shape), dtype=self._dtype, device=dev)   # on the caller's stream
        used = []
        for g in range(self.groups):
            fr = frames[g * per:(g + 1) * per]
            if not fr:
                break
            s = self.streams[g]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                mean, coords, _ = voxelize_batch(self.gen, fr)
                self.engine(mean, coords, len(fr), grid, out=out[g * per:g * per + len(fr)])
            out.record_stream(s)
            used.append(s)
        if join:
            for s in used:
                main.wait_stream(s)
        return out


class FramePipeline2:
    """Throughput mode of FusedBackbone.forward_frames: two plans in flight. Frames are independent, and the first
    ~0.9 ms of a step -- voxelizer, hash build, the first rulebook -- is a chain of small kernels nothing else runs
    beside; so the GEOMETRY graph of step k + 1 (voxelizer + every rulebook, LDS-light kernels that fit beside a conv
    workgroup) is replayed on a second stream while the FEATURE graph of step k (the convolutions) runs on the first.
    Two slots (plan = buffers + graph pair) alternate; from batch 8 on the feature graphs are serialised on one stream
    -- two conv kernels side by side gain nothing there (tools/experiments/corun_probe.py) -- while small batches run them
    on two (see submit); a slot's geometry waits until the feature graph that last read the slot's rulebooks is done.

        pipe = FramePipeline2(engine, gen, grid)
        for clouds in stream_of_frame_lists:
            done = pipe.submit(clouds)                # queues this step, -> output of the PREVIOUS one (None at first)
        last = pipe.drain()                           # -> [output of the last step]

    Outputs are the plan's tensors -- valid until the next submit(), which re-uses their slot -- and bitwise those
    of forward_frames. A step whose counts exceed its slot's capacities is re-run through forward_frames (exact
    re-plan, the output a private copy) and the slots are rebuilt with at least those capacities before their next use."""

    TWO_STREAMS_UP_TO = 4      # frames per step up to which the two slots' feature graphs run side by side

    def __init__(self, engine, gen, grid):
        if engine.eager or engine.dispatch != "graph":
            raise ValueError("FramePipeline2 drives a graph-dispatch FusedBackbone")
        self.engine, self.gen, self.grid = engine, gen, list(grid)
        self.dev = engine.dev
        self.slots = [None, None]          # plans (buffers + geometry / feature graph)
        self.busy = [None, None]           # clouds of the step in flight on the slot
        self.done_ev = [torch.cuda.Event(), torch.cuda.Event()]
        self.geo_ev = [torch.cuda.Event(), torch.cuda.Event()]
        self.k = 0
        self.floor = None                  # a plan whose capacities every slot must reach (set by an overflow)
        self.feat_stream = self.geo_stream = None
        self.rebuilt = 0

    def _covers(self, plan, frames, products):
        f = self.floor
        return (plan is not None and not plan.closed and plan.n_frames == len(frames) and plan.products == products
                and sum(int(c.shape[0]) for c in frames) <= plan.cap_points
                and (f is None or len(f.caps) != len(plan.caps)
                     or (plan.cap_rows >= f.cap_rows and all(a >= b for a, b in zip(plan.caps, f.caps)))))

    def _plan(self, s, frames):
        e = self.engine
        products = e._products()
        plan = self.slots[s]
        if not self._covers(plan, frames, products):
            old = plan if plan is not None and not plan.closed else None
            if old is not None and old.n_frames != len(frames):
                old.close()
                old = None
            plan = self.slots[s] = e._build_plan(self.gen, frames, self.grid, old, products, pair=True,
                                                 floors=(self.slots[1 - s], self.floor))
            self.rebuilt += 1
        return plan

    def submit(self, frames):
        """Queue one step, then wait for the PREVIOUS one: -> its output (None on the first call). The step queued here
        is already running its geometry beside that step's convolutions while the host waits. The returned tensors
        belong to the other slot and stay untouched until the next submit()."""
        if self.feat_stream is None:
            from .train import _concurrent_stream
            # both are the pipeline's own: the caller's stream only hands the clouds over (were the feature graphs on it,
            # "wait for the clouds" would mean "wait for the previous step's convolutions")
            self.feat_stream = torch.cuda.Stream(self.dev)
            with torch.cuda.stream(self.feat_stream):
                self.geo_stream = _concurrent_stream(self.dev)   # one that really runs beside feat_stream (hardware queues)
            with torch.cuda.stream(self.feat_stream):
                self.feat_stream2 = _concurrent_stream(self.dev)
        s = self.k & 1
        assert self.busy[s] is None          # collected by the previous submit
        plan = self._plan(s, frames)
        # Small batches leave CUs idle in every kernel's last partial round of tiles (one FHD frame: 332 tiles of the
        # 128-channel layers on 256 CUs): there the two slots' feature graphs run on two streams and fill each other's
        # gaps (batch 1: 225 -> 248 frames/s, bf16 487 -> 567; batch 4 +2 %). From batch 8 on the rounds are full and
        # two conv kernels side by side gain nothing (-0.6 %): one feature stream. VX_PIPE_FEAT_STREAMS=1|2 overrides.
        two = os.environ.get("VX_PIPE_FEAT_STREAMS") or ("2" if len(frames) <= self.TWO_STREAMS_UP_TO else "1")
        fs, gs = (self.feat_stream2 if (s == 1 and two == "2") else self.feat_stream), self.geo_stream
        gs.wait_stream(torch.cuda.current_stream(self.dev))      # the caller's clouds
        gs.wait_event(self.done_ev[s])                           # the slot's rulebooks were last read by this graph
        with torch.cuda.stream(gs):
            plan.stage(frames)
            plan.graph_geo.replay()
            self.geo_ev[s].record(gs)
        for f in frames:
            f.record_stream(gs)
        fs.wait_event(self.geo_ev[s])
        with torch.cuda.stream(fs):
            plan.graph_feat.replay()
            plan.h_counts.copy_(plan.counts, non_blocking=True)
            self.done_ev[s].record(fs)
        self.busy[s] = frames
        self.k += 1
        return self._collect(1 - s)

    def _collect(self, s):
        frames, plan = self.busy[s], self.slots[s]
        if frames is None:
            return None
        self.busy[s] = None
        self.done_ev[s].synchronize()
        if plan.overflow() is None:
            return plan.out
        # rows were dropped: exact re-run of this step through the engine's own plan (which re-plans from this very
        # input); its capacities become the floor of both slots
        torch.cuda.synchronize(self.dev)
        e = self.engine
        out = e.forward_frames(self.gen, frames, self.grid)
        out = tuple(o.clone() for o in out) if isinstance(out, tuple) else out.clone()
        self.floor = e._plans[(id(self.gen), len(frames), tuple(self.grid), e._products())]
        return out

    def drain(self):
        """Wait for the step in flight; -> [its output] (or [])."""
        outs = []
        for s in ((self.k & 1), (self.k & 1) ^ 1):
            o = self._collect(s)
            if o is not None:
                outs.append(o)
        return outs

    def close(self):
        self.drain()
        for t in (0, 1):
            if self.slots[t] is not None:
                self.slots[t].close()
                self.slots[t] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
