"""Torch-tensor front-ends of the libvx C-ABI (device pointers + the current stream).

PyTorch is plumbing here: it owns device memory (caching allocator) and streams. Every op
requires CUDA(HIP) tensors and raises otherwise: there is no CPU path.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise L.VxError(f"{name} must be a CUDA/HIP tensor: this path has no CPU implementation")


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def fill_(t, value=0):
    """Stream-ordered byte fill of a contiguous CUDA tensor by a kernel (vx_fill): the capturable clear."""
    _need_gpu(t, "tensor")
    if not t.is_contiguous():
        raise ValueError("fill_: tensor must be contiguous")
    with torch.cuda.device(t.device):
        L.check(L.lib().vx_fill(_ptr(t), int(value), t.numel() * t.element_size(), _stream()), "vx_fill")
    return t


_DT = {torch.float32: L.VX_F32, torch.bfloat16: L.VX_BF16}

# bench.py sets this to a list to get HIP-event pairs (recorded on the launch stream) around every
# conv_fwd launch; None (the default) records nothing.
TIMING = None


class _Timed:
    """with _Timed("op", **meta) as rec: ... -- records a HIP-event pair on the current stream."""

    def __init__(self, op, **meta):
        self.rec = None
        if TIMING is not None:
            self.rec = dict(op=op, ev0=torch.cuda.Event(enable_timing=True), ev1=torch.cuda.Event(enable_timing=True),
                            **meta)

    def __enter__(self):
        if self.rec is not None:
            self.rec["ev0"].record()
        return self.rec

    def __exit__(self, *exc):
        if self.rec is not None:
            self.rec["ev1"].record()
            TIMING.append(self.rec)
        return False


def triple(v):
    if isinstance(v, (int, np.integer)):
        return [int(v)] * 3
    v = [int(x) for x in v]
    if len(v) != 3:
        raise ValueError(f"expected an int or a sequence of 3 ints, got {v}")
    return v


# ------------------------------------------------------------------------------------------------
# voxelization + mean VFE
# ------------------------------------------------------------------------------------------------
def voxelize(points, voxel_size, pc_range, max_points, max_voxels, batch_idx=0,
             with_voxels=False, with_mean=True):
    """points f32[N,F] (cuda) -> dict(coords int32[M,4] (b,z,y,x), num_points int32[M],
    voxels f32[M,T,F] | None, mean f32[M,F] | None). One host sync (reads M)."""
    _need_gpu(points, "points")
    if points.dtype != torch.float32 or points.dim() != 2:
        raise ValueError("points must be float32 [N, F]")
    points = points.contiguous()
    n, f = points.shape
    dev = points.device
    vs = np.asarray(voxel_size, dtype=np.float32)
    rg = np.asarray(pc_range, dtype=np.float32)
    if vs.shape != (3,) or rg.shape != (6,):
        raise ValueError("voxel_size must have 3 and pc_range 6 entries")
    cap = max(1, min(n, int(max_voxels)))
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    nump = torch.empty((cap,), dtype=torch.int32, device=dev)
    voxels = torch.empty((cap, max_points, f), dtype=torch.float32, device=dev) if with_voxels else None
    mean = torch.empty((cap, f), dtype=torch.float32, device=dev) if with_mean else None
    d_m = torch.zeros((1,), dtype=torch.int64, device=dev)
    h_m = ctypes.c_int64(0)
    lib = L.lib()
    wsb = lib.vx_voxelize_workspace_bytes(n)
    ws = _ws(wsb, dev)
    with torch.cuda.device(dev), _Timed("voxelize", n_points=n, n_feat=f) as rec:
        rc = lib.vx_voxelize_hard(_ptr(points), n, f, vs.ctypes.data_as(L.c_f32p), rg.ctypes.data_as(L.c_f32p),
                                  int(max_points), int(max_voxels), int(batch_idx), _ptr(coords), _ptr(nump),
                                  _ptr(voxels), _ptr(mean), cap, _ptr(d_m), ctypes.byref(h_m), _ptr(ws), wsb,
                                  _stream())
    L.check(rc, "vx_voxelize_hard")
    m = int(h_m.value)
    if rec is not None:
        rec["n_voxels"] = m
    return dict(coords=coords[:m], num_points=nump[:m],
                voxels=voxels[:m] if voxels is not None else None,
                mean=mean[:m] if mean is not None else None, num_voxels=m)


def voxelize_batch(points_list, voxel_size, pc_range, max_points, max_voxels, with_voxels=False, with_mean=True):
    """Whole batch in one launch sequence: list of f32[N_b,F] clouds -> dict like voxelize() with coords
    column 0 = frame index and `frame_voxels` = [M_0, ...]. Falls back to per-frame calls when a frame
    has more points than max_voxels (only then can the reference's truncation trigger)."""
    for p in points_list:
        _need_gpu(p, "points")
    b = len(points_list)
    if b == 0 or b > 64 or any(p.shape[0] > max_voxels for p in points_list):
        outs = [voxelize(p, voxel_size, pc_range, max_points, max_voxels, batch_idx=i, with_voxels=with_voxels,
                         with_mean=with_mean) for i, p in enumerate(points_list)]
        cat = lambda k: torch.cat([o[k] for o in outs]) if outs and outs[0][k] is not None else None  # noqa: E731
        return dict(coords=cat("coords"), num_points=cat("num_points"), voxels=cat("voxels"), mean=cat("mean"),
                    num_voxels=sum(o["num_voxels"] for o in outs), frame_voxels=[o["num_voxels"] for o in outs])
    pts = torch.cat([p.contiguous() for p in points_list]) if b > 1 else points_list[0].contiguous()
    if pts.dtype != torch.float32 or pts.dim() != 2:
        raise ValueError("points must be float32 [N, F]")
    n, f = pts.shape
    dev = pts.device
    vs = np.asarray(voxel_size, dtype=np.float32)
    rg = np.asarray(pc_range, dtype=np.float32)
    offs = np.zeros(b + 1, dtype=np.int64)
    np.cumsum([p.shape[0] for p in points_list], out=offs[1:])
    cap = max(1, n)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    nump = torch.empty((cap,), dtype=torch.int32, device=dev)
    voxels = torch.empty((cap, max_points, f), dtype=torch.float32, device=dev) if with_voxels else None
    mean = torch.empty((cap, f), dtype=torch.float32, device=dev) if with_mean else None
    d_m = torch.zeros((b + 1,), dtype=torch.int64, device=dev)
    h_m = (ctypes.c_int64 * (b + 1))()
    lib = L.lib()
    wsb = lib.vx_voxelize_workspace_bytes(n)
    ws = _ws(wsb, dev)
    with torch.cuda.device(dev), _Timed("voxelize", n_points=n, n_feat=f) as rec:
        rc = lib.vx_voxelize_hard_batch(_ptr(pts), b, offs.ctypes.data_as(L.c_i64p), f, vs.ctypes.data_as(L.c_f32p),
                                        rg.ctypes.data_as(L.c_f32p), int(max_points), int(max_voxels), _ptr(coords),
                                        _ptr(nump), _ptr(voxels), _ptr(mean), cap, _ptr(d_m), h_m, _ptr(ws), wsb,
                                        _stream())
    L.check(rc, "vx_voxelize_hard_batch")
    m = int(h_m[0])
    per = [int(h_m[i + 1]) for i in range(b)] if b > 1 else [m]
    if rec is not None:
        rec["n_voxels"] = m
    return dict(coords=coords[:m], num_points=nump[:m], voxels=voxels[:m] if voxels is not None else None,
                mean=mean[:m] if mean is not None else None, num_voxels=m, frame_voxels=per)


# ------------------------------------------------------------------------------------------------
# rulebooks
# ------------------------------------------------------------------------------------------------
class Rulebook:
    """Output-major neighbour table nbr[k][o] (+ optional transpose) of one sparse conv geometry."""

    def __init__(self, nbr, n_in, n_out, ksize, stride, padding, subm, in_shape, out_shape,
                 out_indices, nbr_t=None):
        self.nbr, self.nbr_t = nbr, nbr_t
        self.n_in, self.n_out = n_in, n_out
        self.ksize, self.stride, self.padding, self.subm = ksize, stride, padding, subm
        self.in_shape, self.out_shape = list(in_shape), list(out_shape)
        self.out_indices = out_indices
        self.kvol = ksize[0] * ksize[1] * ksize[2]
        self._pairs = None
        self._num_pairs = None
        self.out_rank = None  # rank table of the output site set (strided convs): see rulebook_subm
        self._sorted = None

    def pairs(self):
        """(indice_pairs int32[K,2,n_out], indice_pair_num int32[K]) -- spconv's pair-list view."""
        if self._pairs is None:
            self._pairs = nbr_to_pairs(self.nbr)
        return self._pairs

    def num_pairs(self):
        """Active (input, output, offset) triples of this rulebook (one host sync, cached)."""
        if self._num_pairs is None:
            self._num_pairs = int((self.nbr[:, :self.n_out] >= 0).sum().item()) if self.n_out else 0
        return self._num_pairs

    def sorted_table(self):
        """(nbr_sorted int32[K,n_out], perm int32[n_out]): mask-sorted row order for conv_fwd."""
        if self._sorted is None:
            nbr_s = torch.empty_like(self.nbr)
            perm = torch.empty((max(self.n_out, 1),), dtype=torch.int32, device=self.nbr.device)
            with torch.cuda.device(self.nbr.device):
                rc = L.lib().vx_rulebook_sort_rows(_ptr(self.nbr), max(self.n_out, 1), self.kvol, self.n_out,
                                                   _ptr(nbr_s), max(self.n_out, 1), _ptr(perm), _stream())
            L.check(rc, "vx_rulebook_sort_rows")
            self._sorted = (nbr_s, perm)
        return self._sorted

    @property
    def indice_pairs(self):
        return self.pairs()[0]

    @property
    def indice_pair_num(self):
        return self.pairs()[1]

    def transpose(self):
        """nbr_t[k][i] = output row fed by input row i through offset k (strided convs)."""
        if self.subm:
            return self.nbr
        if self.nbr_t is None:
            k, n_out = self.nbr.shape
            t = torch.full((k, max(self.n_in, 1)), -1, dtype=torch.int32, device=self.nbr.device)
            kk, oo = torch.nonzero(self.nbr >= 0, as_tuple=True)
            t[kk, self.nbr[kk, oo].long()] = oo.int()
            self.nbr_t = t
        return self.nbr_t


def rulebook_subm(indices, batch_size, spatial_shape, ksize, rank=None):
    """Submanifold rulebook. `rank` = rank table of this site set (Rulebook.out_rank of the strided
    conv that produced `indices`): lookups then need no hash table."""
    _need_gpu(indices, "indices")
    if indices.dtype != torch.int32 or indices.dim() != 2 or indices.shape[1] != 4:
        raise ValueError("indices must be int32 [N, 4] (batch, z, y, x)")
    indices = indices.contiguous()
    ks = triple(ksize)
    n = indices.shape[0]
    kvol = ks[0] * ks[1] * ks[2]
    nbr = torch.empty((kvol, n), dtype=torch.int32, device=indices.device)
    lib = L.lib()
    if rank is not None:
        rb = Rulebook(nbr, n, n, ks, [1, 1, 1], [k // 2 for k in ks], True, spatial_shape, spatial_shape, indices)
        with torch.cuda.device(indices.device), _Timed("rulebook_subm", n=n, kvol=kvol, rb=rb):
            rc = lib.vx_rulebook_subm_ranked(_ptr(indices), n, int(batch_size), L.i32x3(spatial_shape),
                                             L.i32x3(ks), _ptr(rank), _ptr(nbr), max(n, 1), _stream())
        L.check(rc, "vx_rulebook_subm_ranked")
        return rb
    wsb = lib.vx_rulebook_workspace_bytes(n, int(batch_size), None)
    ws = _ws(wsb, indices.device)
    rb = Rulebook(nbr, n, n, ks, [1, 1, 1], [k // 2 for k in ks], True, spatial_shape, spatial_shape, indices)
    with torch.cuda.device(indices.device), _Timed("rulebook_subm", n=n, kvol=kvol, rb=rb):
        rc = lib.vx_rulebook_subm(_ptr(indices), n, int(batch_size), L.i32x3(spatial_shape), L.i32x3(ks),
                                  _ptr(nbr), max(n, 1), _ptr(ws), wsb, _stream())
    L.check(rc, "vx_rulebook_subm")
    return rb


def conv_out_shape(in_shape, ksize, stride, padding):
    ks, st, pd = triple(ksize), triple(stride), triple(padding)
    return [(int(in_shape[d]) + 2 * pd[d] - ks[d]) // st[d] + 1 for d in range(3)]


def rulebook_conv(indices, batch_size, spatial_shape, ksize, stride, padding, transpose=False):
    _need_gpu(indices, "indices")
    if indices.dtype != torch.int32 or indices.dim() != 2 or indices.shape[1] != 4:
        raise ValueError("indices must be int32 [N, 4] (batch, z, y, x)")
    indices = indices.contiguous()
    dev = indices.device
    ks, st, pd = triple(ksize), triple(stride), triple(padding)
    n = indices.shape[0]
    kvol = ks[0] * ks[1] * ks[2]
    osh_host = conv_out_shape(spatial_shape, ks, st, pd)
    if min(osh_host) < 1:
        raise ValueError(f"sparse conv output shape {osh_host} is empty")
    lib = L.lib()
    osh = L.i32x3(osh_host)
    wsb = lib.vx_rulebook_workspace_bytes(n, int(batch_size), osh)
    ws = _ws(wsb, dev)
    h_n = ctypes.c_int64(0)
    args = (_ptr(indices), n, int(batch_size), L.i32x3(spatial_shape), L.i32x3(ks), L.i32x3(st), L.i32x3(pd))
    with torch.cuda.device(dev), _Timed("rulebook_conv", n=n, kvol=kvol) as rec:
        rc = lib.vx_rulebook_conv_count(*args, osh, ctypes.byref(h_n), _ptr(ws), wsb, _stream())
        L.check(rc, "vx_rulebook_conv_count")
        n_out = int(h_n.value)
        out_idx = torch.empty((n_out, 4), dtype=torch.int32, device=dev)
        nbr = torch.empty((kvol, n_out), dtype=torch.int32, device=dev)
        nbr_t = torch.empty((kvol, max(n, 1)), dtype=torch.int32, device=dev) if transpose else None
        rank = torch.empty((max(lib.vx_rank_table_bytes(int(batch_size), osh), 8),), dtype=torch.uint8, device=dev)
        rc = lib.vx_rulebook_conv_emit(*args, n_out, _ptr(out_idx), _ptr(nbr), max(n_out, 1), _ptr(nbr_t),
                                       max(n, 1), _ptr(rank), _ptr(ws), wsb, _stream())
        L.check(rc, "vx_rulebook_conv_emit")
    rb = Rulebook(nbr, n, n_out, ks, st, pd, False, spatial_shape, list(osh), out_idx, nbr_t)
    rb.out_rank = rank if n and n_out else None
    if rec is not None:
        rec["rb"] = rb
    return rb


def nbr_to_pairs(nbr):
    _need_gpu(nbr, "nbr")
    kvol, n_out = nbr.shape
    pairs = torch.empty((kvol, 2, n_out), dtype=torch.int32, device=nbr.device)
    num = torch.empty((kvol,), dtype=torch.int32, device=nbr.device)
    lib = L.lib()
    wsb = lib.vx_pairs_workspace_bytes(kvol, n_out)
    ws = _ws(wsb, nbr.device)
    with torch.cuda.device(nbr.device):
        rc = lib.vx_nbr_to_pairs(_ptr(nbr), max(n_out, 1), kvol, n_out, _ptr(pairs), _ptr(num), _ptr(ws), wsb,
                                 _stream())
    L.check(rc, "vx_nbr_to_pairs")
    return pairs, num


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
def pack_weight(weight):
    """fp32 KRSC weight -> MFMA-fragment-ordered copy for the streaming-W kernel (conv_fwd(wpack=...))."""
    _need_gpu(weight, "weight")
    if weight.dtype != torch.float32:
        raise ValueError("pack_weight: fp32 weights only")
    weight = weight.contiguous()
    cout, cin = weight.shape[0], weight.shape[-1]
    kvol = weight.numel() // (cout * cin)
    lib = L.lib()
    out = torch.empty((lib.vx_spconv_pack_weight_bytes(cin, cout, kvol),), dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib.vx_spconv_pack_weight(_ptr(weight), cin, cout, kvol, _ptr(out), _stream())
    L.check(rc, "vx_spconv_pack_weight")
    return out


def _arith(products):
    try:
        return L.ARITH_FLAG[products]
    except KeyError:
        raise ValueError("products must be None (library default), 0 (fp32 MFMA), 6 or 9 (bf16 plane products)")


def conv_fwd(features, weight, rb, bias=None, scale=None, shift=None, residual=None, relu=False,
             sort_rows=False, wpack=None, compact=True, split_tail=True, products=None):
    """features [n_in,Cin], weight [Cout,kz,ky,kx,Cin] (KRSC) -> [n_out,Cout]; fused epilogue
    relu?((acc + bias) * scale + shift + residual). sort_rows: run on the rulebook's mask-sorted
    row order (same bits out); wpack: packed weight copy -> barrier-free streaming-W kernel (fp32).
    Both are OFF by default: measured on MI355X they only pay together, on the
    sparse strided layers, and the sort costs what they save. compact=False forces the row-stationary
    kernels where the library would pick the per-offset compaction kernel (fp32, Cin <= 64); split_tail=False
    launches the row-stationary kernel's plain grid (bitwise the same output, for A/B timing and tests).
    products: fp32 matrix arithmetic of THIS call (0 / 6 / 9, include/vx_abi.h); None = the library default."""
    _need_gpu(features, "features")
    if features.dtype not in _DT:
        raise ValueError(f"unsupported feature dtype {features.dtype}")
    if weight.dtype != features.dtype:
        raise ValueError("weight and features must share a dtype")
    features = features.contiguous()
    weight = weight.contiguous()
    cout, cin = weight.shape[0], weight.shape[-1]
    if features.dim() != 2 or features.shape[1] != cin or features.shape[0] != rb.n_in:
        raise ValueError(f"features {tuple(features.shape)} do not match rulebook/weight ({rb.n_in}, {cin})")
    if weight.numel() != cout * rb.kvol * cin:
        raise ValueError("weight shape does not match the rulebook's kernel size")
    out = torch.empty((rb.n_out, cout), dtype=features.dtype, device=features.device)
    for t in (bias, scale, shift):
        if t is not None and (t.dtype != torch.float32 or t.numel() != cout or not t.is_cuda):
            raise ValueError("bias/scale/shift must be float32 CUDA tensors of length Cout")
    if residual is not None:
        residual = residual.contiguous()
        if residual.shape != out.shape or residual.dtype != out.dtype:
            raise ValueError("residual must match the output shape and dtype")
    nbr, perm = rb.nbr, None
    if wpack is not None or sort_rows:
        compact = False
    if sort_rows and rb.n_out >= 4096 and rb.kvol > 1:
        nbr, perm = rb.sorted_table()
    with torch.cuda.device(features.device), _Timed("conv", cin=cin, cout=cout, kvol=rb.kvol, rb=rb,
                                                   esize=features.element_size()):
        rc = L.lib().vx_spconv_fwd(_ptr(features), rb.n_in, cin, _ptr(weight), rb.kvol, cout, _ptr(nbr),
                                   max(rb.n_out, 1), rb.n_out, _ptr(bias), _ptr(scale), _ptr(shift),
                                   _ptr(residual), (L.VX_EPI_RELU if relu else 0) | (0 if compact else L.VX_CONV_NO_COMPACT)
                                   | (0 if split_tail else L.VX_CONV_NO_SPLIT_TAIL) | _arith(products),
                                   _ptr(perm), _ptr(wpack),
                                   _ptr(out), _DT[features.dtype], _stream())
    L.check(rc, "vx_spconv_fwd")
    return out


def conv_dgrad(grad_out, weight, rb, products=None):
    _need_gpu(grad_out, "grad_out")
    grad_out = grad_out.contiguous()
    cout, cin = weight.shape[0], weight.shape[-1]
    weight = weight.permute(4, 1, 2, 3, 0).contiguous()  # [Cin, kz, ky, kx, Cout]
    din = torch.empty((rb.n_in, cin), dtype=grad_out.dtype, device=grad_out.device)
    tbl = rb.transpose()
    with torch.cuda.device(grad_out.device):
        rc = L.lib().vx_spconv_dgrad(_ptr(grad_out), rb.n_out, cout, _ptr(weight), rb.kvol, cin, _ptr(tbl),
                                     max(rb.n_in, 1), rb.n_in, 1 if rb.subm else 0, _arith(products), _ptr(din),
                                     _DT[grad_out.dtype], _stream())
    L.check(rc, "vx_spconv_dgrad")
    return din


def to_dense(features, indices, batch_size, spatial_shape, out=None):
    """-> [B, C, D, H, W] (channels first, contiguous); `out` = a contiguous tensor of that many elements to fill
    instead of allocating (e.g. a slice of a larger batch)."""
    _need_gpu(features, "features")
    features = features.contiguous()
    indices = indices.contiguous()
    n, c = features.shape
    sh = triple(spatial_shape)
    shape = (int(batch_size), c, *sh)
    if out is None:
        dense = torch.empty(shape, dtype=features.dtype, device=features.device)
    else:
        if (not out.is_contiguous() or out.dtype != features.dtype or out.device != features.device
                or out.numel() != int(np.prod(shape))):
            raise ValueError(f"out must be a contiguous {features.dtype} tensor of {shape} elements on {features.device}")
        dense = out.view(shape)
    with torch.cuda.device(features.device), _Timed("dense", n=n, c=c, numel=dense.numel(),
                                                   esize=features.element_size()):
        rc = L.lib().vx_to_dense(_ptr(features), _ptr(indices), n, c, int(batch_size), L.i32x3(sh), _ptr(dense),
                                 _DT[features.dtype], _stream())
    L.check(rc, "vx_to_dense")
    return dense


def set_f32_products(products):
    """Matrix arithmetic of the fp32 convolutions with Cin >= 64 (include/vx_abi.h): 0 = fp32 MFMA, 9 = exact
    split-bf16 plane products, 6 = the six largest of them. Returns the previous setting."""
    lib = L.lib()
    old = lib.vx_spconv_get_f32_products()
    L.check(lib.vx_spconv_set_f32_products(int(products)), "vx_spconv_set_f32_products")
    return old


def from_dense(dense, indices, spatial_shape):
    """The gradient of to_dense: dense [B, C, D, H, W] contiguous -> [n, C] rows gathered at `indices`."""
    _need_gpu(dense, "dense")
    dense = dense.contiguous()
    indices = indices.contiguous()
    sh = triple(spatial_shape)
    if dense.dim() != 5 or list(dense.shape[2:]) != sh or dense.dtype not in _DT:
        raise ValueError(f"dense must be [B, C, {sh[0]}, {sh[1]}, {sh[2]}] in float32 or bfloat16")
    n, c = indices.shape[0], dense.shape[1]
    out = torch.empty((n, c), dtype=dense.dtype, device=dense.device)
    with torch.cuda.device(dense.device):
        rc = L.lib().vx_from_dense(_ptr(dense), _ptr(indices), n, c, dense.shape[0], L.i32x3(sh), _ptr(out),
                                   _DT[dense.dtype], _stream())
    L.check(rc, "vx_from_dense")
    return out


def conv_wgrad(features, grad_out, rb, weight_shape, products=None, out=None):
    """-> dW float32 with the KRSC shape of the weight. out: a contiguous float32 tensor of that shape to OVERWRITE
    instead of allocating (e.g. the parameter's view into a flat gradient buffer)."""
    _need_gpu(features, "features")
    features = features.contiguous()
    grad_out = grad_out.contiguous()
    cout, cin = int(weight_shape[0]), int(weight_shape[-1])
    pairs, num = rb.pairs()
    if out is None:
        dw = torch.empty(tuple(weight_shape), dtype=torch.float32, device=features.device)
    else:
        if (out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape) != tuple(weight_shape)
                or out.device != features.device):
            raise ValueError("conv_wgrad: out must be a contiguous float32 tensor of the weight's shape on the same device")
        dw = out
    lib = L.lib()
    wsb = lib.vx_spconv_wgrad_workspace_bytes(cin, cout, rb.kvol, rb.n_out)
    ws = _ws(wsb, features.device)
    with torch.cuda.device(features.device):
        rc = lib.vx_spconv_wgrad(_ptr(features), rb.n_in, cin, _ptr(grad_out), rb.n_out, cout, _ptr(pairs),
                                 _ptr(num), rb.kvol, _ptr(dw), _DT[features.dtype], _arith(products), _ptr(ws), wsb,
                                 _stream())
    L.check(rc, "vx_spconv_wgrad")
    return dw


# ------------------------------------------------------------------------------------------------
# train-mode BatchNorm1d (+ residual + ReLU) over sparse features
# ------------------------------------------------------------------------------------------------
def bn_supported(c):
    """Widths the fused BatchNorm kernels take: a power of two in [8, 512]."""
    return 8 <= c <= 512 and (c & (c - 1)) == 0


def _bn_check(x, name):
    _need_gpu(x, name)
    if x.dtype != torch.float32 or x.dim() != 2 or not bn_supported(x.shape[1]):
        raise ValueError(f"{name} must be float32 [n, c] with c a power of two in [8, 512], got {tuple(x.shape)} {x.dtype}")
    return x.contiguous()


def bn_stats(x):
    """-> f64[2c+1] = (sum x, sum x^2 per channel, n). Under SyncBN the caller all-reduces this tensor."""
    x = _bn_check(x, "x")
    n, c = x.shape
    sums = torch.empty((2 * c + 1,), dtype=torch.float64, device=x.device)
    lib = L.lib()
    wsb = lib.vx_bn_workspace_bytes(c)
    ws = _ws(wsb, x.device)
    with torch.cuda.device(x.device):
        rc = lib.vx_bn_stats(_ptr(x), n, c, _ptr(sums), _ptr(ws), wsb, _stream())
    L.check(rc, "vx_bn_stats")
    return sums


def bn_finalize(sums, c, eps, momentum, running_mean=None, running_var=None):
    """sums f64[2c+1] -> (mean, invstd) f32[c]; updates the running statistics in place when given."""
    _need_gpu(sums, "sums")
    if sums.dtype != torch.float64 or sums.numel() != 2 * c + 1 or not sums.is_contiguous():
        raise ValueError("sums must be a contiguous float64 tensor of 2c+1 elements")
    for t in (running_mean, running_var):
        if t is not None and (t.dtype != torch.float32 or t.numel() != c or not t.is_cuda or not t.is_contiguous()):
            raise ValueError("running statistics must be contiguous float32 CUDA tensors of length c")
    mean = torch.empty((c,), dtype=torch.float32, device=sums.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=sums.device)
    with torch.cuda.device(sums.device):
        rc = L.lib().vx_bn_finalize(_ptr(sums), c, float(eps), float(momentum), _ptr(running_mean), _ptr(running_var),
                                    _ptr(mean), _ptr(invstd), _stream())
    L.check(rc, "vx_bn_finalize")
    return mean, invstd


def _bn_vec(t, c, name):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or t.numel() != c:
        raise ValueError(f"{name} must be a float32 CUDA tensor of length c")
    return t.contiguous()


def bn_apply(x, mean, invstd, gamma=None, beta=None, residual=None, relu=False):
    """-> (y, mask): y = relu?((x - mean) * invstd * gamma + beta + residual?); mask uint8[n*c/8] (one bit per
    element, y > 0) when relu else None."""
    x = _bn_check(x, "x")
    n, c = x.shape
    mean, invstd = _bn_vec(mean, c, "mean"), _bn_vec(invstd, c, "invstd")
    gamma, beta = _bn_vec(gamma, c, "gamma"), _bn_vec(beta, c, "beta")
    if residual is not None:
        residual = _bn_check(residual, "residual")
        if residual.shape != x.shape:
            raise ValueError("residual must match x")
    y = torch.empty_like(x)
    mask = torch.empty((n * c // 8,), dtype=torch.uint8, device=x.device) if relu else None
    with torch.cuda.device(x.device):
        rc = L.lib().vx_bn_apply(_ptr(x), n, c, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(residual),
                                 1 if relu else 0, _ptr(y), _ptr(mask), _stream())
    L.check(rc, "vx_bn_apply")
    return y, mask


def bn_bwd_reduce(dy, x, mask, mean, invstd, pg_out=None):
    """-> (sums f64[2c] = (sum g, sum g*xhat), param_grads f32[2c] = (d beta, d gamma) of this rank). pg_out: a contiguous
    float32[2c] tensor to OVERWRITE with the parameter gradients instead of allocating one."""
    dy, x = _bn_check(dy, "dy"), _bn_check(x, "x")
    if dy.shape != x.shape:
        raise ValueError("dy must match x")
    n, c = x.shape
    if mask is not None and (mask.dtype != torch.uint8 or mask.numel() != n * c // 8 or not mask.is_cuda):
        raise ValueError("mask must be the uint8[n*c/8] tensor bn_apply returned")
    sums = torch.empty((2 * c,), dtype=torch.float64, device=x.device)
    if pg_out is None:
        pg = torch.empty((2 * c,), dtype=torch.float32, device=x.device)
    else:
        if pg_out.dtype != torch.float32 or pg_out.numel() != 2 * c or not pg_out.is_contiguous() or pg_out.device != x.device:
            raise ValueError("pg_out must be a contiguous float32[2c] tensor on x's device")
        pg = pg_out
    lib = L.lib()
    wsb = lib.vx_bn_workspace_bytes(c)
    ws = _ws(wsb, x.device)
    mean, invstd = _bn_vec(mean, c, "mean"), _bn_vec(invstd, c, "invstd")
    with torch.cuda.device(x.device):
        rc = lib.vx_bn_bwd_reduce(_ptr(dy), _ptr(x), _ptr(mask), n, c, _ptr(mean), _ptr(invstd), _ptr(sums), _ptr(pg),
                                  _ptr(ws), wsb, _stream())
    L.check(rc, "vx_bn_bwd_reduce")
    return sums, pg


def bn_bwd_apply(dy, x, mask, mean, invstd, gamma, sums, count, want_dres=False):
    """-> (dx, dres | None). sums f64[2c] (all-reduced under SyncBN), count f64[1] = total rows."""
    dy, x = _bn_check(dy, "dy"), _bn_check(x, "x")
    n, c = x.shape
    if sums.dtype != torch.float64 or sums.numel() != 2 * c or count.dtype != torch.float64 or count.numel() != 1:
        raise ValueError("sums must be float64[2c] and count float64[1]")
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    mean, invstd, gamma = _bn_vec(mean, c, "mean"), _bn_vec(invstd, c, "invstd"), _bn_vec(gamma, c, "gamma")
    sums, count = sums.contiguous(), count.contiguous()
    with torch.cuda.device(x.device):
        rc = L.lib().vx_bn_bwd_apply(_ptr(dy), _ptr(x), _ptr(mask), n, c, _ptr(mean), _ptr(invstd), _ptr(gamma),
                                     _ptr(sums), _ptr(count), _ptr(dx), _ptr(dres), _stream())
    L.check(rc, "vx_bn_bwd_apply")
    return dx, dres
