"""ORACLE (test infrastructure): CPU restatement of the reference's sparse backbones
on top of oracle/spconv_oracle.c.

Follows the layer lists of det3d/models/backbones/scn.py:332-364 (SpMiddleResNetFHD),
det3d/models/backbones/multiview.py:33-86 (MultiViewBackbone), the residual block of
scn.py:82-98 (conv1 -> bn1 -> relu -> conv2 -> bn2 -> +identity -> relu; both convs carry a bias,
scn.py:68-72), the forward glue of scn.py:367-381 / multiview.py:88-106, and BatchNorm1d(eps=1e-5)
(det3d/models/utils/norm.py:62,99-103). Parity for the spconv arithmetic is unpinned by the
reference (no fixtures); see oracle/spconv_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import numpy as np

from . import spconv_oracle as so

EPS = 1e-5


def _bn(x, sd, name, train):
    g, b = sd[name + ".weight"], sd[name + ".bias"]
    if train:
        mean = x.mean(0, dtype=np.float64)
        var = x.var(0, dtype=np.float64)  # biased, as BatchNorm uses for normalisation
    else:
        mean, var = sd[name + ".running_mean"].astype(np.float64), sd[name + ".running_var"].astype(np.float64)
    y = (x.astype(np.float64) - mean) / np.sqrt(var + EPS) * g + b
    return y.astype(np.float32)


def _relu(x):
    return np.maximum(x, 0.0)


class Tensor:
    def __init__(self, feat, coords, shape, batch):
        self.feat, self.coords, self.shape, self.batch = feat, coords, [int(s) for s in shape], batch
        self.cache = {}


def _block(t, sd, p, key, train, stats):
    if key not in t.cache:
        t.cache[key] = so.rulebook_subm(t.coords, t.shape, 3)
    nbr = t.cache[key]
    idt = t.feat
    x = so.conv_fwd(t.feat, sd[p + ".conv1.weight"], nbr, sd[p + ".conv1.bias"])
    x = _relu(_bn(x, sd, p + ".bn1", train))
    x = so.conv_fwd(x, sd[p + ".conv2.weight"], nbr, sd[p + ".conv2.bias"])
    x = _bn(x, sd, p + ".bn2", train)
    t.feat = _relu(x + idt)
    if stats is not None:
        stats.append((p, int((nbr >= 0).sum()) * 2, t.feat.shape[0]))
    return t


# (type, args) mirroring the SparseSequential indices of the reference
def _middle(with_last):
    L = [("subm", "res0"), ("bn",), ("relu",), ("block", "res0"), ("block", "res0"),
         ("conv", 3, 2, 1), ("bn",), ("relu",), ("block", "res1"), ("block", "res1"),
         ("conv", 3, 2, 1), ("bn",), ("relu",), ("block", "res2"), ("block", "res2"),
         ("conv", 3, 2, [0, 1, 1]), ("bn",), ("relu",), ("block", "res3"), ("block", "res3")]
    if with_last:
        L += [("conv", (3, 1, 1), (2, 1, 1), 0), ("bn",), ("relu",)]
    return L


_BEV = [("conv", (3, 1, 1), (2, 1, 1), 0), ("bn",), ("relu",)]
_RV = [("conv", (1, 1, 3), (1, 1, 2), (0, 0, 1)), ("bn",), ("relu",), ("block", "res4"),
       ("conv", (1, 1, 3), (1, 1, 2), (0, 0, 1)), ("bn",), ("relu",), ("block", "res5"),
       ("conv", (1, 1, 3), (1, 1, 2), (0, 0, 1)), ("bn",), ("relu",)]


def run_sequential(t, sd, prefix, layers, train=False, stats=None):
    """SparseSequential semantics: sparse modules take the tensor, BN/ReLU take .features."""
    new = Tensor(t.feat, t.coords, t.shape, t.batch)
    new.cache = t.cache
    t = new
    for i, l in enumerate(layers):
        p = f"{prefix}.{i}"
        if l[0] == "subm":
            if l[1] not in t.cache:
                t.cache[l[1]] = so.rulebook_subm(t.coords, t.shape, 3)
            nbr = t.cache[l[1]]
            t.feat = so.conv_fwd(t.feat, sd[p + ".weight"], nbr, sd.get(p + ".bias"))
            if stats is not None:
                stats.append((p, int((nbr >= 0).sum()), t.feat.shape[0]))
        elif l[0] == "conv":
            oc, nbr, osh = so.rulebook_conv(t.coords, t.shape, l[1], l[2], l[3])
            feat = so.conv_fwd(t.feat, sd[p + ".weight"], nbr, sd.get(p + ".bias"))
            t = Tensor(feat, oc, osh, t.batch)
            if stats is not None:
                stats.append((p, int((nbr >= 0).sum()), feat.shape[0]))
        elif l[0] == "bn":
            if t.feat.shape[0]:
                t.feat = _bn(t.feat, sd, p, train)
        elif l[0] == "relu":
            t.feat = _relu(t.feat)
        elif l[0] == "block":
            t = _block(t, sd, p, l[1], train, stats)
    return t


def sp_middle_resnet_fhd(voxel_features, coors, batch_size, input_shape, sd, train=False, stats=None):
    """scn.py:367-381. input_shape = grid [X,Y,Z]; returns f32[B, C*D, H, W]."""
    sparse_shape = (np.array(input_shape[::-1]) + [1, 0, 0]).tolist()
    t = Tensor(np.asarray(voxel_features, np.float32), np.asarray(coors, np.int32), sparse_shape, batch_size)
    t = run_sequential(t, sd, "middle_conv", _middle(True), train, stats)
    d = so.to_dense(t.feat, t.coords, batch_size, t.shape)
    n, c, dd, h, w = d.shape
    return d.reshape(n, c * dd, h, w), t


def multi_view_backbone(voxel_features, coors, batch_size, input_shape, sd, train=False, stats=None,
                        return_tensors=False):
    """multiview.py:88-106. returns (bev f32[B,C*D,H,W], rv f32[B,C*W,H,D]) (+ the two sparse tensors they were
    densified from when return_tensors is set: tests compare active sets with them)."""
    sparse_shape = (np.array(input_shape[::-1]) + [1, 0, 0]).tolist()
    t = Tensor(np.asarray(voxel_features, np.float32), np.asarray(coors, np.int32), sparse_shape, batch_size)
    t = run_sequential(t, sd, "middle_conv", _middle(False), train, stats)
    tb = run_sequential(t, sd, "bev_conv", _BEV, train, stats)
    bev = so.to_dense(tb.feat, tb.coords, batch_size, tb.shape)
    n, c, d, h, w = bev.shape
    bev = bev.reshape(n, c * d, h, w)
    tr = run_sequential(t, sd, "rv_conv", _RV, train, stats)
    rv = so.to_dense(tr.feat, tr.coords, batch_size, tr.shape)
    n, c, d, h, w = rv.shape
    rv = np.ascontiguousarray(rv.transpose(0, 1, 4, 3, 2)).reshape(n, c * w, h, d)
    if return_tensors:
        return bev, rv, tb, tr
    return bev, rv


def active_mask(t, layout="bev"):
    """bool array shaped like the densified output of sparse tensor `t`: True on every element of an active site."""
    ones = np.ones((t.feat.shape[0], t.feat.shape[1]), dtype=np.float32)
    d = so.to_dense(ones, t.coords, t.batch, t.shape)
    n, c, dd, h, w = d.shape
    if layout == "bev":
        return d.reshape(n, c * dd, h, w) != 0
    return np.ascontiguousarray(d.transpose(0, 1, 4, 3, 2)).reshape(n, c * w, h, dd) != 0
