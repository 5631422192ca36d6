"""SubMConv3d / SparseConv3d: nn.Modules with the constructor surface the reference uses
(scn.py:30-38,333-361; multiview.py:34-82), backed by libvx rulebooks + MFMA implicit GEMM.

Weight layout: KRSC [Cout, kz, ky, kx, Cin] (spconv >= 2.2). `convert_weight`
maps the two older spconv layouts onto it.
"""
import math

import torch
from torch import nn
from torch.nn import init

from vista_amd import ops as vx

from .core import SparseConvTensor
from .modules import SparseModule


class _SparseConvFn(torch.autograd.Function):
    """out = conv(features, weight) + bias over a rulebook; dgrad/wgrad are the libvx kernels."""

    @staticmethod
    def forward(ctx, features, weight, bias, rb, bias_grad_zero=False, sink_w=None, sink_b=None):
        ctx.rb = rb
        ctx.has_bias = bias is not None
        ctx.bias_grad_zero = bias_grad_zero
        # gradient sinks (vista_amd.train.GradBuckets): the parameter's slice of the flat gradient buffer, written by
        # the wgrad kernel itself instead of through autograd's accumulate add. A parameter used more than once in a
        # forward pass falls back to autograd's accumulation.
        ctx.sink_w, ctx.sink_b = sink_w, sink_b
        for sk in (sink_w, sink_b):
            if sk is not None:
                sk.uses += 1
        ctx.save_for_backward(features, weight)
        return vx.conv_fwd(features, weight, rb, bias=bias)

    @staticmethod
    def backward(ctx, grad_out):
        features, weight = ctx.saved_tensors
        rb = ctx.rb
        grad_out = grad_out.contiguous()
        din = dw = db = None
        if ctx.needs_input_grad[1]:
            sk = ctx.sink_w
            if sk is not None and sk.uses == 1 and weight.dtype == torch.float32:
                side = sk.gb.wgrad_stream(features.device)
                if side is None:
                    vx.conv_wgrad(features, grad_out, rb, weight.shape, out=sk.view)
                else:
                    # wgrad and dgrad of a layer both start from grad_out and nothing downstream reads the weight
                    # gradient before the exchange: wgrad goes to the buckets' side stream (issued BEFORE dgrad, so it
                    # waits for grad_out only) and runs beside the dgrad / BatchNorm kernels of the main stream
                    held = (features, grad_out) + tuple(rb.pairs())   # pair list built once, on the main stream
                    side.wait_stream(torch.cuda.current_stream(features.device))
                    with torch.cuda.stream(side):
                        vx.conv_wgrad(features, grad_out, rb, weight.shape, out=sk.view)
                    for t in held:          # freed by autograd / with the rulebook while the side stream may still read them
                        t.record_stream(side)
                sk.done()
            else:
                dw = vx.conv_wgrad(features, grad_out, rb, weight.shape).to(weight.dtype)
        if ctx.needs_input_grad[0]:
            din = vx.conv_dgrad(grad_out, weight, rb)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # in front of a batch-statistics norm the column sums of grad_out are zero by construction (the norm
            # subtracts the batch mean, SyncBatchNorm the global one): write the zeros, skip the reduction
            sk = ctx.sink_b
            if sk is not None and sk.uses == 1:
                if not ctx.bias_grad_zero:
                    torch.sum(grad_out, 0, out=sk.view)
                sk.done()                              # (the flat buffer was zeroed at the start of the step)
            else:
                db = grad_out.new_zeros(grad_out.shape[1]) if ctx.bias_grad_zero else grad_out.sum(0)
        return din, dw, db, None, None, None, None


def convert_weight(w, layout):
    """Map a spconv checkpoint weight onto KRSC [Cout,kz,ky,kx,Cin].
    layout: 'KRSC' (spconv>=2.2), 'RSCK' ([kz,ky,kx,Cin,Cout], spconv 1.x), 'RSKC' ([kz,ky,kx,Cout,Cin], 2.0/2.1 native)."""
    if layout == "KRSC":
        return w
    if layout == "RSCK":
        return w.permute(4, 0, 1, 2, 3).contiguous()
    if layout == "RSKC":
        return w.permute(3, 0, 1, 2, 4).contiguous()
    raise ValueError(f"unknown weight layout {layout}")


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, subm=False, output_padding=0, transposed=False, inverse=False,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__()
        if ndim != 3:
            raise NotImplementedError("only 3-D sparse convolutions are on the VISTA path")
        if groups != 1 or transposed or inverse:
            raise NotImplementedError("grouped / transposed / inverse sparse convs are outside the VISTA path")
        self.ndim = ndim
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size = vx.triple(kernel_size)
        self.stride = vx.triple(stride)
        self.padding = vx.triple(padding)
        self.dilation = vx.triple(dilation)
        if self.dilation != [1, 1, 1]:
            raise NotImplementedError("dilated sparse convolutions are outside the VISTA path")
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.name = name
        self.weight = nn.Parameter(torch.empty(self.out_channels, *self.kernel_size, self.in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(self.out_channels))
        else:
            self.register_parameter("bias", None)
        object.__setattr__(self, "norm_after", None)    # set by vista_amd.fused_bn (not a submodule of the conv)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        s = (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
             f"padding={self.padding}, subm={self.subm}")
        if self.bias is None:
            s += ", bias=False"
        if self.indice_key is not None:
            s += f", indice_key={self.indice_key}"
        return s

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # det3d/torchie/trainer/checkpoint.py:42-92 silently skips shape mismatches; make a wrong
        # weight layout loud instead of loading nothing.
        key = prefix + "weight"
        if key in state_dict and tuple(state_dict[key].shape) != tuple(self.weight.shape):
            raise RuntimeError(
                f"{key}: checkpoint weight shape {tuple(state_dict[key].shape)} != KRSC "
                f"{tuple(self.weight.shape)}; convert with spconv.pytorch.conv.convert_weight")
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _rulebook(self, x):
        rb = x.find_indice_pair(self.indice_key)
        if rb is not None:
            if self.subm:
                if rb.ksize != self.kernel_size or rb.n_in != x.indices.shape[0]:
                    raise ValueError(f"indice_key {self.indice_key}: cached rulebook does not match this layer")
                return rb
            if (rb.ksize, rb.stride, rb.padding) == (self.kernel_size, self.stride, self.padding) \
                    and rb.n_in == x.indices.shape[0] and not rb.subm:
                return rb
            raise ValueError(f"indice_key {self.indice_key}: cached rulebook does not match this layer")
        if self.subm:
            rb = vx.rulebook_subm(x.indices, x.batch_size, x.spatial_shape, self.kernel_size,
                                  rank=getattr(x, "rank_table", None))
        else:
            rb = vx.rulebook_conv(x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.stride,
                                  self.padding, transpose=torch.is_grad_enabled())
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = rb
        return rb

    def forward(self, input):
        if not isinstance(input, SparseConvTensor):
            raise TypeError("sparse convolutions take a SparseConvTensor")
        feats = input.features
        if feats.shape[1] != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {feats.shape[1]}")
        rb = self._rulebook(input)
        weight = self.weight if self.weight.dtype == feats.dtype else self.weight.to(feats.dtype)
        norm = self.norm_after
        direct = weight is self.weight and torch.is_grad_enabled()
        out_feats = _SparseConvFn.apply(feats, weight, self.bias, rb, norm is not None and norm.training,
                                        getattr(self.weight, "_vx_grad_sink", None) if direct else None,
                                        getattr(self.bias, "_vx_grad_sink", None) if direct and self.bias is not None else None)
        out = SparseConvTensor(out_feats, input.indices if self.subm else rb.out_indices,
                               input.spatial_shape if self.subm else rb.out_shape, input.batch_size,
                               input.grid, input.voxel_num, input.indice_dict, input.benchmark)
        out.benchmark_record = input.benchmark_record
        out.rank_table = input.rank_table if self.subm else rb.out_rank
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, algo=None, fp32_accum=None, name=None):
        # submanifold: stride 1, padding forced to k//2 whatever is passed (scn.py:46-48 passes padding=1 with k=1)
        super().__init__(3, in_channels, out_channels, kernel_size, 1, 0, dilation, groups, bias, True,
                         indice_key=indice_key, algo=algo, fp32_accum=fp32_accum, name=name)
        self.padding = [k // 2 for k in self.kernel_size]
        if any(k % 2 == 0 for k in self.kernel_size):
            raise ValueError("submanifold convolution needs odd kernel sizes")


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         False, indice_key=indice_key, algo=algo, fp32_accum=fp32_accum, name=name)
