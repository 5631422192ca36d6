"""ORACLE (test infrastructure): ctypes front-end of oracle/spconv_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    src = os.path.join(_HERE, "spconv_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.orc_rulebook_conv.restype = ctypes.c_int64
        L.orc_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _triple(v):
    if isinstance(v, (int, np.integer)):
        return [int(v)] * 3
    v = [int(x) for x in v]
    assert len(v) == 3
    return v


def out_shape(in_shape, ksize, stride, pad):
    k, s, p = _triple(ksize), _triple(stride), _triple(pad)
    return [(int(in_shape[d]) + 2 * p[d] - k[d]) // s[d] + 1 for d in range(3)]


def rulebook_subm(coords, shape, ksize):
    """-> nbr int32[K, N] (input row feeding output row o at offset k, or -1)."""
    coords = _i32(coords)
    k = _triple(ksize)
    K = k[0] * k[1] * k[2]
    n = coords.shape[0]
    nbr = np.full((K, n), -1, np.int32)
    if n:
        rc = lib().orc_rulebook_subm(_p(coords, c_i32p), ctypes.c_int64(n), _p(_i32(shape), c_i32p),
                                     _p(_i32(k), c_i32p), _p(nbr, c_i32p))
        assert rc == 0
    return nbr


def rulebook_conv(coords, in_shape, ksize, stride, pad):
    """-> (out_coords int32[N_out,4] sorted by linear index, nbr int32[K,N_out], out_shape)."""
    coords = _i32(coords)
    k, s, p = _triple(ksize), _triple(stride), _triple(pad)
    K = k[0] * k[1] * k[2]
    n = coords.shape[0]
    osh = out_shape(in_shape, k, s, p)
    if min(osh) < 1:
        raise ValueError(f"sparse conv output shape {osh} is empty")
    if n == 0:
        return np.zeros((0, 4), np.int32), np.zeros((K, 0), np.int32), osh
    args = (_p(coords, c_i32p), ctypes.c_int64(n), _p(_i32(in_shape), c_i32p), _p(_i32(k), c_i32p),
            _p(_i32(s), c_i32p), _p(_i32(p), c_i32p))
    n_out = lib().orc_rulebook_conv(*args, None, None, ctypes.c_int64(0))
    assert n_out >= 0
    oc = np.zeros((n_out, 4), np.int32)
    nbr = np.full((K, n_out), -1, np.int32)
    rc = lib().orc_rulebook_conv(*args, _p(oc, c_i32p), _p(nbr, c_i32p), ctypes.c_int64(n_out))
    assert rc == n_out, rc
    return oc, nbr, osh


def nbr_to_pairs(nbr):
    """-> pairs int32[K,2,N_out] (in,out; -1 padded), pair_num int32[K]."""
    nbr = _i32(nbr)
    K, n_out = nbr.shape
    pairs = np.full((K, 2, n_out), -1, np.int32)
    num = np.zeros((K,), np.int32)
    if n_out:
        lib().orc_nbr_to_pairs(_p(nbr, c_i32p), ctypes.c_int(K), ctypes.c_int64(n_out),
                               _p(pairs, c_i32p), _p(num, c_i32p))
    return pairs, num


def _real(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _kind(dtype):
    """(numpy dtype, ctypes pointer, symbol suffix) of a precision: float32 = the oracle proper, float64 = the
    truth tests use to separate the GPU's rounding error from the fp32 oracle's own."""
    dt = np.dtype(dtype)
    if dt == np.float32:
        return np.float32, c_f32p, ""
    if dt == np.float64:
        return np.float64, c_f64p, "_f64"
    raise ValueError(f"oracle precisions are float32 and float64, not {dt}")


def conv_fwd(feat, weight, nbr, bias=None, dtype=np.float32):
    """feat [N_in,Cin], weight [Cout,kz,ky,kx,Cin] (KRSC), nbr int32[K,N_out] -> [N_out,Cout] in `dtype`."""
    dt, ptr, sfx = _kind(dtype)
    feat, weight, nbr = _real(feat, dt), _real(weight, dt), _i32(nbr)
    cout, cin = weight.shape[0], weight.shape[-1]
    K, n_out = nbr.shape
    assert weight.size == cout * K * cin and feat.shape[1] == cin
    out = np.zeros((n_out, cout), dt)
    b = _real(bias, dt) if bias is not None else None
    if n_out:
        rc = getattr(lib(), "orc_conv_fwd" + sfx)(_p(feat, ptr), ctypes.c_int64(feat.shape[0]), ctypes.c_int(cin),
                                                  _p(weight, ptr), ctypes.c_int(K), ctypes.c_int(cout),
                                                  _p(nbr, c_i32p), ctypes.c_int64(n_out), _p(b, ptr), _p(out, ptr))
        assert rc == 0
    return out


def conv_dgrad(dout, weight, nbr, n_in, dtype=np.float32):
    dt, ptr, sfx = _kind(dtype)
    dout, weight, nbr = _real(dout, dt), _real(weight, dt), _i32(nbr)
    cout, cin = weight.shape[0], weight.shape[-1]
    K, n_out = nbr.shape
    din = np.zeros((n_in, cin), dt)
    if n_in and n_out:
        getattr(lib(), "orc_conv_dgrad" + sfx)(_p(dout, ptr), ctypes.c_int64(n_out), ctypes.c_int(cout),
                                               _p(weight, ptr), ctypes.c_int(K), ctypes.c_int(cin),
                                               _p(nbr, c_i32p), ctypes.c_int64(n_in), _p(din, ptr))
    return din


def conv_wgrad(feat, dout, nbr, wshape, with_bias=False, dtype=np.float32):
    dt, ptr, sfx = _kind(dtype)
    feat, dout, nbr = _real(feat, dt), _real(dout, dt), _i32(nbr)
    cout, cin = wshape[0], wshape[-1]
    K, n_out = nbr.shape
    dw = np.zeros(wshape, dt)
    db = np.zeros((cout,), dt) if with_bias else None
    getattr(lib(), "orc_conv_wgrad" + sfx)(_p(feat, ptr), ctypes.c_int64(feat.shape[0]), ctypes.c_int(cin),
                                           _p(dout, ptr), ctypes.c_int64(n_out), ctypes.c_int(cout),
                                           _p(nbr, c_i32p), ctypes.c_int(K), _p(dw, ptr), _p(db, ptr))
    return (dw, db) if with_bias else dw


def to_dense(feat, coords, batch, shape):
    feat, coords = _f32(feat), _i32(coords)
    c = feat.shape[1]
    dense = np.zeros((batch, c, int(shape[0]), int(shape[1]), int(shape[2])), np.float32)
    lib().orc_to_dense(_p(feat, c_f32p), _p(coords, c_i32p), ctypes.c_int64(feat.shape[0]),
                       ctypes.c_int(c), ctypes.c_int(batch), _p(_i32(shape), c_i32p), _p(dense, c_f32p))
    return dense


def set_num_threads(n):
    lib().orc_set_num_threads(ctypes.c_int(int(n)))


def num_threads():
    return int(lib().orc_num_threads())
