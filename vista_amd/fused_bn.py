"""Training-time fusion of the BatchNorm1d + (residual add) + ReLU passes over sparse features.

The reference applies these as separate torch modules to `.features` (det3d/models/backbones/scn.py:82-98 for the
residual block, :333-364 for conv + BN + ReLU; BatchNorm1d built by det3d/models/utils/norm.py:99-103) and turns every
BatchNorm into SyncBatchNorm when it trains on more than one GPU (det3d/torchie/apis/train.py:301). Here one
`FusedBatchNorm1d` runs the libvx kernels of vista_amd/csrc/vx_bn.hip: forward = statistics + apply (residual and
ReLU folded in, one mask bit per element kept for backward), backward = reduce + apply. With `sync=True` the two
small double-precision sum vectors are all-reduced over the process group between the halves, which is everything
SyncBatchNorm exchanges (RCCL when the backend is "nccl").

`fuse_for_training(net)` rewrites a backbone in place; parameters, buffers and state_dict keys are unchanged (the fused
module sits at the BatchNorm's index, the ReLU's index holds an Identity).
"""
import torch
import torch.nn.functional as F
from torch import nn

import spconv.pytorch as spconv

from . import ops
from .models import SparseBasicBlock


def _all_reduce_sum(t, group):
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) != "nccl":
        # rehearsal backends (gloo) reduce host memory; RCCL reduces in place on the GPU
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, mod, sinks=None):
        x = x.contiguous()
        c = x.shape[1]
        sums = ops.bn_stats(x)
        if mod.sync:
            mod.reduce_sums(sums)
        mean, invstd = ops.bn_finalize(sums, c, mod.eps, mod.momentum, mod.running_mean, mod.running_var)
        y, mask = ops.bn_apply(x, mean, invstd, gamma, beta, residual, mod.relu)
        ctx.save_for_backward(x, gamma, mean, invstd, mask, sums[2 * c:])
        ctx.mod = mod
        ctx.has_res = residual is not None
        ctx.sinks = sinks
        if sinks is not None:
            sinks[0].uses += 1
            sinks[1].uses += 1
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd, mask, count = ctx.saved_tensors
        mod = ctx.mod
        c = x.shape[1]
        dy = dy.contiguous()
        direct = ctx.sinks is not None and ctx.sinks[0].uses == 1 and ctx.sinks[1].uses == 1 \
            and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        pg_out = ctx.sinks[1].gb.flat[ctx.sinks[1].lo:ctx.sinks[1].lo + 2 * c] if direct else None
        sums, pgrads = ops.bn_bwd_reduce(dy, x, mask, mean, invstd, pg_out=pg_out)
        if mod.sync:
            mod.reduce_sums(sums)
        want_dres = ctx.has_res and ctx.needs_input_grad[3]
        dx, dres = ops.bn_bwd_apply(dy, x, mask, mean, invstd, gamma, sums, count,
                                    want_dres=want_dres and mask is not None)
        if want_dres and mask is None:
            dres = dy            # no ReLU between the add and the output: the residual branch sees dy itself
        if direct:
            ctx.sinks[1].done()
            ctx.sinks[0].done()
            return dx, None, None, dres, None, None
        return dx, pgrads[c:], pgrads[:c], dres, None, None


class FusedBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters, buffers and eval behaviour) whose TRAINING forward on HIP tensors runs the
    fused kernels. relu=True applies the ReLU that followed the norm; forward(x, residual=r) adds r before it.
    sync=True makes the batch statistics global over `process_group` like SyncBatchNorm; `all_reduce` (a callable
    that sums a float64 tensor over ranks in place) replaces the torch.distributed call, e.g. for rehearsals."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, relu=False, sync=False, process_group=None,
                 all_reduce=None):
        super().__init__(num_features, eps=eps, momentum=momentum)
        self.relu, self.sync, self.process_group = bool(relu), bool(sync), process_group
        self.all_reduce = all_reduce

    def reduce_sums(self, t):
        if self.all_reduce is not None:
            self.all_reduce(t)
        else:
            _all_reduce_sum(t, self.process_group)

    @classmethod
    def from_bn(cls, bn, relu=False, sync=False, process_group=None):
        if not (bn.affine and bn.track_running_stats and bn.momentum is not None):
            raise ValueError("FusedBatchNorm1d needs an affine BatchNorm1d with running statistics and a momentum")
        new = cls(bn.num_features, bn.eps, bn.momentum, relu, sync, process_group)
        new.weight, new.bias = bn.weight, bn.bias
        new.running_mean, new.running_var = bn.running_mean, bn.running_var
        new.num_batches_tracked = bn.num_batches_tracked
        new.train(bn.training)
        return new

    def extra_repr(self):
        return super().extra_repr() + f", relu={self.relu}, sync={self.sync}"

    def forward(self, x, residual=None):
        fused = (self.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                 and ops.bn_supported(x.shape[1]))
        if not fused:
            if self.training and self.sync:
                raise ops.L.VxError("FusedBatchNorm1d(sync=True) trains on float32 HIP tensors [n, c] only")
            y = super().forward(x)
            if residual is not None:
                y = y + residual
            return F.relu(y) if self.relu else y
        self.num_batches_tracked.add_(1)
        # gradient sinks of (gamma, beta) (vista_amd.train.GradBuckets): when beta and gamma sit next to each other in
        # the flat gradient buffer, in that order, vx_bn_bwd_reduce writes its {d beta, d gamma} vector straight there
        sinks = None
        sg, sb = getattr(self.weight, "_vx_grad_sink", None), getattr(self.bias, "_vx_grad_sink", None)
        if (sg is not None and sb is not None and torch.is_grad_enabled()
                and sb.view.data_ptr() + 4 * x.shape[1] == sg.view.data_ptr()):
            sinks = (sg, sb)
        return _BatchNormFn.apply(x, self.weight, self.bias, residual, self, sinks)


class FusedSparseBasicBlock(spconv.SparseModule):
    """SparseBasicBlock (scn.py:54-98) with bn1+ReLU and bn2+add+ReLU as one fused op each; shares the block's
    submodules, so parameters and state_dict keys are the block's."""

    def __init__(self, blk, sync=False, process_group=None):
        super().__init__()
        self.conv1 = blk.conv1
        self.bn1 = FusedBatchNorm1d.from_bn(blk.bn1, True, sync, process_group)
        self.relu = blk.relu
        self.conv2 = blk.conv2
        self.bn2 = FusedBatchNorm1d.from_bn(blk.bn2, True, sync, process_group)
        self.downsample, self.stride = blk.downsample, blk.stride
        _norm_follows(self.conv1, self.bn1)
        _norm_follows(self.conv2, self.bn2)

    def forward(self, x):
        y = self.conv1(x)
        y = y.replace_feature(self.bn1(y.features))
        y = self.conv2(y)
        skip = x if self.downsample is None else self.downsample(x)
        return y.replace_feature(self.bn2(y.features, residual=skip.features))


def _norm_follows(conv, bn):
    """Tell a sparse conv that `bn` consumes its output directly: while bn is in training mode the conv's bias
    gradient is identically zero (d loss / d bias = column sums of the norm's input gradient, which the mean
    subtraction makes vanish), so its backward writes zeros instead of reducing grad_out over the rows. The bias
    keeps requires_grad and its place in the optimizer (weight decay still applies, as in the reference)."""
    if isinstance(conv, spconv.SparseConvolution) and conv.bias is not None:
        object.__setattr__(conv, "norm_after", bn)


def fuse_for_training(module, sync=False, process_group=None):
    """Rewrite `module` in place: [BatchNorm1d, ReLU] runs of every SparseSequential and every SparseBasicBlock
    become fused ops. Returns `module`. sync=True: statistics over all ranks (use when world_size > 1, as the
    reference converts to SyncBatchNorm)."""
    for name, child in list(module._modules.items()):
        if child is None:
            continue
        if type(child) is SparseBasicBlock:
            module._modules[name] = FusedSparseBasicBlock(child, sync, process_group)
        elif isinstance(child, spconv.SparseSequential):
            names = list(child._modules.keys())
            for i, nm in enumerate(names):
                m = child._modules[nm]
                if type(m) is nn.BatchNorm1d:
                    nxt = child._modules[names[i + 1]] if i + 1 < len(names) else None
                    relu = type(nxt) is nn.ReLU
                    child._modules[nm] = FusedBatchNorm1d.from_bn(m, relu, sync, process_group)
                    if i > 0:
                        _norm_follows(child._modules[names[i - 1]], child._modules[nm])
                    if relu:
                        child._modules[names[i + 1]] = nn.Identity()
            fuse_for_training(child, sync, process_group)
        else:
            fuse_for_training(child, sync, process_group)
    return module
