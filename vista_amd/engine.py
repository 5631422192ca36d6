"""Inference engine for the sparse backbones: same results as running the nn.Module in eval mode,
with BatchNorm1d(eval) + bias + ReLU + residual folded into the conv kernel's epilogue
(scn.py:82-98 collapses from 2 GEMM sets + 4 elementwise passes to 2 launches per block) and all
rulebooks of a frame built before the feature pass (they depend on coordinates only).
"""
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
        s.w32 = conv.weight.detach().to(dev, torch.float32).contiguous() if conv.in_channels % 8 else None
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


def _run(steps, st, geom_only=False):
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
                                   residual=res, relu=s.relu, wpack=s.wpack if w is s.w else None)
            if st.feat.dtype != s.w.dtype:
                st.feat = st.feat.to(s.w.dtype)
        if s.kind == "conv":
            st.idx, st.shape, st.rank = rb.out_indices, rb.out_shape, rb.out_rank
    return st


class FusedBackbone:
    """Eval-mode forward of SpMiddleResNetFHD / MultiViewBackbone with fused epilogues."""

    def __init__(self, module, dtype=torch.float32, eager=True):
        dev = next(module.parameters()).device
        self.kind = type(module).__name__
        self.dtype = dtype
        self.eager = bool(eager)
        self.mode = getattr(module, "mode", "bev")
        self.middle = _compile_sequential(module.middle_conv, dev, dtype)
        if self.kind == "MultiViewBackbone":
            self.bev = _compile_sequential(module.bev_conv, dev, dtype)
            self.rv = _compile_sequential(module.rv_conv, dev, dtype)
        elif self.kind != "SpMiddleResNetFHD":
            raise TypeError(f"unsupported backbone {self.kind}")

    def dispatch_name(self):
        return "eager per-layer launches"

    def eager_twin(self):
        """The same plan with per-layer launches from the host (each C-ABI call can then be bracketed by events)."""
        return self

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
        mid = _run(self.middle, st)
        if self.kind == "SpMiddleResNetFHD":
            d = ops.to_dense(mid.feat, mid.idx, mid.batch, mid.shape, out=out)
            n, c, dd, h, w = d.shape
            return d.view(n, c * dd, h, w) if self.mode == "bev" else d
        if out is not None:
            raise NotImplementedError("out= is only supported for SpMiddleResNetFHD")
        b = _run(self.bev, mid)
        bev = ops.to_dense(b.feat, b.idx, b.batch, b.shape)
        n, c, d, h, w = bev.shape
        bev = bev.view(n, c * d, h, w)
        r = _run(self.rv, mid)
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
