"""CPU: the C sparse-conv oracle against an independent dense formulation
(torch.nn.functional.conv3d + autograd). The reference holds no fixture for the
spconv boundary ("parity unpinned"); this is the mathematical pin.

Geometries are the ones the backbones use (scn.py:333-361, multiview.py:68-86)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import spconv_oracle as so

GEOMS = [
    # (ksize, stride, pad)
    ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
    ((3, 1, 1), (2, 1, 1), (0, 0, 0)),
    ((1, 1, 3), (1, 1, 2), (0, 0, 1)),
    ((3, 3, 3), (1, 1, 1), (1, 1, 1)),
]


def random_sites(rng, batch, shape, n):
    vol = int(np.prod(shape))
    rows = []
    for b in range(batch):
        lin = rng.choice(vol, size=min(n, vol), replace=False)
        z, r = np.divmod(lin, shape[1] * shape[2])
        y, x = np.divmod(r, shape[2])
        rows.append(np.stack([np.full_like(z, b), z, y, x], 1))
    c = np.concatenate(rows).astype(np.int32)
    return c[rng.permutation(c.shape[0])]


def densify(feat, coords, batch, shape):
    d = torch.zeros(batch, feat.shape[1], *shape, dtype=torch.float64)
    c = torch.as_tensor(coords).long()
    d[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = torch.as_tensor(feat, dtype=torch.float64)
    return d


def w_dense(w):  # KRSC [Cout,kz,ky,kx,Cin] -> torch [Cout,Cin,kz,ky,kx]
    return torch.as_tensor(w, dtype=torch.float64).permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("ks,st,pd", GEOMS)
def test_sparse_conv_equals_dense(ks, st, pd):
    rng = np.random.default_rng(0)
    batch, shape, cin, cout = 2, [9, 14, 12], 5, 7
    coords = random_sites(rng, batch, shape, 150)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    w = rng.normal(size=(cout, *ks, cin)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    oc, nbr, osh = so.rulebook_conv(coords, shape, ks, st, pd)
    out = so.conv_fwd(feat, w, nbr, bias)
    # canonical order: strictly increasing linear index
    lin = ((oc[:, 0].astype(np.int64) * osh[0] + oc[:, 1]) * osh[1] + oc[:, 2]) * osh[2] + oc[:, 3]
    assert np.all(np.diff(lin) > 0)
    dense_in = densify(feat, coords, batch, shape)
    ref = F.conv3d(dense_in, w_dense(w), None, stride=st, padding=pd)
    assert list(ref.shape[2:]) == osh
    # active output set == sites whose receptive field holds an active input
    occ = F.conv3d((densify(np.ones((coords.shape[0], 1), np.float32), coords, batch, shape)),
                   torch.ones(1, 1, *ks, dtype=torch.float64), None, stride=st, padding=pd)
    act = torch.nonzero(occ[:, 0] > 0).numpy()
    np.testing.assert_array_equal(act, oc)  # nonzero() is also in linear-index order
    got = so.to_dense(out - bias[None], oc, batch, osh)
    np.testing.assert_allclose(got, ref.numpy(), rtol=1e-5, atol=1e-5)
    # outside the active set the dense conv is exactly zero (no input in the window)
    mask = so.to_dense(np.ones((oc.shape[0], 1), np.float32), oc, batch, osh)[:, 0] > 0
    assert float(ref.abs().amax(1)[~torch.as_tensor(mask)].max()) == 0.0


def test_subm_equals_masked_dense():
    rng = np.random.default_rng(1)
    batch, shape, cin, cout = 2, [8, 11, 13], 6, 4
    coords = random_sites(rng, batch, shape, 200)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    w = rng.normal(size=(cout, 3, 3, 3, cin)).astype(np.float32)
    nbr = so.rulebook_subm(coords, shape, 3)
    out = so.conv_fwd(feat, w, nbr, None)
    ref = F.conv3d(densify(feat, coords, batch, shape), w_dense(w), None, padding=1)
    c = torch.as_tensor(coords).long()
    ref_rows = ref[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]].numpy()
    np.testing.assert_allclose(out, ref_rows, rtol=1e-5, atol=1e-5)
    # symmetry of the submanifold rulebook: nbr[k][o]==i  <=>  nbr[K-1-k][i]==o
    K = nbr.shape[0]
    for k in range(K):
        o = np.flatnonzero(nbr[k] >= 0)
        np.testing.assert_array_equal(nbr[K - 1 - k][nbr[k][o]], o)
    # centre offset is the identity
    np.testing.assert_array_equal(nbr[K // 2], np.arange(coords.shape[0]))


@pytest.mark.parametrize("ks,st,pd", GEOMS)
def test_grads_equal_dense_autograd(ks, st, pd):
    rng = np.random.default_rng(2)
    batch, shape, cin, cout = 1, [7, 10, 9], 4, 5
    coords = random_sites(rng, batch, shape, 120)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    w = rng.normal(size=(cout, *ks, cin)).astype(np.float32)
    oc, nbr, osh = so.rulebook_conv(coords, shape, ks, st, pd)
    dout = rng.normal(size=(oc.shape[0], cout)).astype(np.float32)
    din = so.conv_dgrad(dout, w, nbr, coords.shape[0])
    dw, db = so.conv_wgrad(feat, dout, nbr, w.shape, with_bias=True)
    x = densify(feat, coords, batch, shape).requires_grad_(True)
    wt = w_dense(w).requires_grad_(True)
    y = F.conv3d(x, wt, None, stride=st, padding=pd)
    y.backward(densify(dout, oc, batch, osh))
    c = torch.as_tensor(coords).long()
    np.testing.assert_allclose(din, x.grad[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]].numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dw, wt.grad.permute(0, 2, 3, 4, 1).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(db, dout.sum(0), rtol=1e-5, atol=1e-5)


def test_pairs_view_and_empty():
    rng = np.random.default_rng(3)
    coords = random_sites(rng, 1, [5, 6, 7], 40)
    oc, nbr, _ = so.rulebook_conv(coords, [5, 6, 7], 3, 2, 1)
    pairs, num = so.nbr_to_pairs(nbr)
    assert num.sum() == (nbr >= 0).sum()
    for k in range(nbr.shape[0]):
        o = pairs[k, 1, :num[k]]
        assert np.all(np.diff(o) > 0)
        np.testing.assert_array_equal(nbr[k][o], pairs[k, 0, :num[k]])
    # every input feeds at least one output for k3 s2 p1 and appears in >=1 pair
    assert set(np.unique(nbr[nbr >= 0])) == set(range(coords.shape[0]))
    oc0, nbr0, _ = so.rulebook_conv(np.zeros((0, 4), np.int32), [5, 6, 7], 3, 2, 1)
    assert oc0.shape == (0, 4) and nbr0.shape == (27, 0)


@pytest.mark.parametrize("ks,st,pd", [GEOMS[0], GEOMS[4]])
def test_fp64_oracle_variant_against_dense_autograd(ks, st, pd):
    """The _f64 routines (the truth GPU gradient tests measure rounding error against) agree with dense float64
    conv3d + autograd to float64 rounding, and the fp32 oracle agrees with them to fp32 rounding."""
    rng = np.random.default_rng(4)
    batch, shape, cin, cout = 1, [7, 10, 9], 4, 5
    coords = random_sites(rng, batch, shape, 120)
    feat = rng.normal(size=(coords.shape[0], cin))
    w = rng.normal(size=(cout, *ks, cin))
    bias = rng.normal(size=(cout,))
    oc, nbr, osh = so.rulebook_conv(coords, shape, ks, st, pd)
    dout = rng.normal(size=(oc.shape[0], cout))
    out = so.conv_fwd(feat, w, nbr, bias, dtype=np.float64)
    din = so.conv_dgrad(dout, w, nbr, coords.shape[0], dtype=np.float64)
    dw, db = so.conv_wgrad(feat, dout, nbr, w.shape, with_bias=True, dtype=np.float64)
    assert out.dtype == din.dtype == dw.dtype == db.dtype == np.float64
    x = densify(feat, coords, batch, shape).requires_grad_(True)
    wt = w_dense(w).requires_grad_(True)
    y = F.conv3d(x, wt, torch.as_tensor(bias), stride=st, padding=pd)
    y.backward(densify(dout, oc, batch, osh))
    c, o = torch.as_tensor(coords).long(), torch.as_tensor(oc).long()
    np.testing.assert_allclose(out, y.detach()[o[:, 0], :, o[:, 1], o[:, 2], o[:, 3]].numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(din, x.grad[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]].numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(dw, wt.grad.permute(0, 2, 3, 4, 1).numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(db, dout.sum(0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(so.conv_fwd(feat, w, nbr, bias), out, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(so.conv_dgrad(dout, w, nbr, coords.shape[0]), din, rtol=2e-5, atol=2e-5)
    with pytest.raises(ValueError):
        so.conv_fwd(feat, w, nbr, bias, dtype=np.float16)
