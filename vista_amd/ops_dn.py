"""Front-ends of the "_dn" entry points of libvx (include/vx_abi.h): the ops of vista_amd.ops with their
data-dependent row counts left on the device. Nothing here synchronises with the host, reads a count or sizes a
tensor from one: every buffer has a host-known CAPACITY, so a sequence of these calls can be queued without waiting
on the GPU and captured in a HIP graph (vista_amd.engine.FramePlan). The caller compares counts and capacities
after the pass.

A count is a 1-element int64 CUDA tensor (usually a view into one counts vector).
"""
import ctypes

import numpy as np
import torch

from . import _lib as L
from .ops import _DT, _need_gpu, _ptr, _stream, _ws, conv_out_shape, triple


def _count(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.int64 or t.numel() < 1:
        raise ValueError(f"{name} must be an int64 CUDA tensor (the device-resident row count)")
    return t


def voxelize_batch_dn(points, frame_offsets, n_frames, voxel_size, pc_range, max_points, cap_rows, num_voxels):
    """points f32[cap_points, F] (frames concatenated; rows past the last offset are ignored), frame_offsets int64
    CUDA [n_frames + 1], num_voxels int64 CUDA [n_frames + 1] (receives {total, per frame}; only [0] when
    n_frames == 1) -> (mean f32[cap_rows, F], coords int32[cap_rows, 4], num_points int32[cap_rows])."""
    _need_gpu(points, "points")
    if points.dtype != torch.float32 or points.dim() != 2 or not points.is_contiguous():
        raise ValueError("points must be a contiguous float32 [cap_points, F] tensor")
    _count(frame_offsets, "frame_offsets")
    _count(num_voxels, "num_voxels")
    if frame_offsets.numel() != n_frames + 1 or num_voxels.numel() < (n_frames + 1 if n_frames > 1 else 1):
        raise ValueError("frame_offsets needs n_frames + 1 entries and num_voxels room for {total, per frame}")
    capp, f = points.shape
    dev = points.device
    vs = np.asarray(voxel_size, dtype=np.float32)
    rg = np.asarray(pc_range, dtype=np.float32)
    coords = torch.empty((cap_rows, 4), dtype=torch.int32, device=dev)
    nump = torch.empty((cap_rows,), dtype=torch.int32, device=dev)
    mean = torch.empty((cap_rows, f), dtype=torch.float32, device=dev)
    lib = L.lib()
    wsb = lib.vx_voxelize_workspace_bytes(capp)
    ws = _ws(wsb, dev)
    with torch.cuda.device(dev):
        rc = lib.vx_voxelize_hard_batch_dn(_ptr(points), int(n_frames), _ptr(frame_offsets), capp, f,
                                           vs.ctypes.data_as(L.c_f32p), rg.ctypes.data_as(L.c_f32p), int(max_points),
                                           _ptr(coords), _ptr(nump), _ptr(mean), int(cap_rows), _ptr(num_voxels),
                                           _ptr(ws), wsb, _stream())
    L.check(rc, "vx_voxelize_hard_batch_dn")
    return mean, coords, nump


class RulebookDn:
    """Neighbour table nbr[k][o] with row stride `cap`; the row count lives in `n_out` (device)."""

    def __init__(self, nbr, n_out, cap, kvol, out_indices=None, out_shape=None, out_rank=None):
        self.nbr, self.n_out, self.cap, self.kvol = nbr, n_out, cap, kvol
        self.out_indices, self.out_shape, self.out_rank = out_indices, out_shape, out_rank


def rulebook_subm_dn(indices, n, batch_size, spatial_shape, ksize, rank=None):
    _need_gpu(indices, "indices")
    _count(n, "n")
    ks = triple(ksize)
    cap = indices.shape[0]
    kvol = ks[0] * ks[1] * ks[2]
    nbr = torch.empty((kvol, cap), dtype=torch.int32, device=indices.device)
    lib = L.lib()
    with torch.cuda.device(indices.device):
        if rank is not None:
            rc = lib.vx_rulebook_subm_ranked_dn(_ptr(indices), _ptr(n), cap, int(batch_size), L.i32x3(spatial_shape),
                                                L.i32x3(ks), _ptr(rank), _ptr(nbr), cap, _stream())
            L.check(rc, "vx_rulebook_subm_ranked_dn")
        else:
            wsb = lib.vx_rulebook_workspace_bytes(cap, int(batch_size), None)
            ws = _ws(wsb, indices.device)
            rc = lib.vx_rulebook_subm_dn(_ptr(indices), _ptr(n), cap, int(batch_size), L.i32x3(spatial_shape),
                                         L.i32x3(ks), _ptr(nbr), cap, _ptr(ws), wsb, _stream())
            L.check(rc, "vx_rulebook_subm_dn")
    return RulebookDn(nbr, n, cap, kvol)


def rulebook_conv_dn(indices, n_in, batch_size, spatial_shape, ksize, stride, padding, cap_out, n_out, err):
    """-> RulebookDn with out_indices int32[cap_out, 4], out_shape, out_rank; *n_out receives the output row count."""
    _need_gpu(indices, "indices")
    _count(n_in, "n_in")
    _count(n_out, "n_out")
    _count(err, "err")
    dev = indices.device
    ks, st, pd = triple(ksize), triple(stride), triple(padding)
    cap_in = indices.shape[0]
    kvol = ks[0] * ks[1] * ks[2]
    osh_host = conv_out_shape(spatial_shape, ks, st, pd)
    if min(osh_host) < 1:
        raise ValueError(f"sparse conv output shape {osh_host} is empty")
    lib = L.lib()
    osh = L.i32x3(osh_host)
    wsb = lib.vx_rulebook_workspace_bytes(cap_in, int(batch_size), osh)
    ws = _ws(wsb, dev)
    out_idx = torch.empty((cap_out, 4), dtype=torch.int32, device=dev)
    nbr = torch.empty((kvol, cap_out), dtype=torch.int32, device=dev)
    rank = torch.empty((max(lib.vx_rank_table_bytes(int(batch_size), osh), 8),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.vx_rulebook_conv_dn(_ptr(indices), _ptr(n_in), cap_in, int(batch_size), L.i32x3(spatial_shape),
                                     L.i32x3(ks), L.i32x3(st), L.i32x3(pd), osh, int(cap_out), _ptr(out_idx),
                                     _ptr(n_out), _ptr(nbr), int(cap_out), _ptr(rank), _ptr(err), _ptr(ws), wsb,
                                     _stream())
    L.check(rc, "vx_rulebook_conv_dn")
    return RulebookDn(nbr, n_out, int(cap_out), kvol, out_idx, list(osh), rank)


def conv_fwd_dn(features, weight, rb, bias=None, scale=None, shift=None, residual=None, relu=False, products=None):
    """features [cap_in, Cin] -> [rb.cap, Cout]; rows at and past the device count are left unwritten."""
    _need_gpu(features, "features")
    if features.dtype not in _DT or weight.dtype != features.dtype:
        raise ValueError("features / weight dtype")
    if not features.is_contiguous() or not weight.is_contiguous():
        raise ValueError("features and weight must be contiguous")
    cout, cin = weight.shape[0], weight.shape[-1]
    if features.dim() != 2 or features.shape[1] != cin or weight.numel() != cout * rb.kvol * cin:
        raise ValueError("features / weight / rulebook shapes do not match")
    out = torch.empty((rb.cap, cout), dtype=features.dtype, device=features.device)
    if residual is not None and (residual.shape != out.shape or residual.dtype != out.dtype
                                 or not residual.is_contiguous()):
        raise ValueError("residual must match the output")
    with torch.cuda.device(features.device):
        rc = L.lib().vx_spconv_fwd_dn(_ptr(features), features.shape[0], cin, _ptr(weight), rb.kvol, cout,
                                      _ptr(rb.nbr), rb.cap, _ptr(rb.n_out), rb.cap, _ptr(bias), _ptr(scale),
                                      _ptr(shift), _ptr(residual),
                                      (L.VX_EPI_RELU if relu else 0) | L.ARITH_FLAG[products], _ptr(out),
                                      _DT[features.dtype], _stream())
    L.check(rc, "vx_spconv_fwd_dn")
    return out


def to_dense_dn(features, indices, n, batch_size, spatial_shape):
    _need_gpu(features, "features")
    _count(n, "n")
    cap, c = features.shape
    sh = triple(spatial_shape)
    dense = torch.empty((int(batch_size), c, *sh), dtype=features.dtype, device=features.device)
    with torch.cuda.device(features.device):
        rc = L.lib().vx_to_dense_dn(_ptr(features), _ptr(indices), _ptr(n), cap, c, int(batch_size), L.i32x3(sh),
                                    _ptr(dense), _DT[features.dtype], _stream())
    L.check(rc, "vx_to_dense_dn")
    return dense
