"""The three matrix arithmetics of the fp32 row-stationary convolution (include/vx_abi.h, vx_spconv_set_f32_products):
0 = fp32 MFMA, 9 = every fp32 operand split exactly into three bf16 planes and all nine plane products accumulated in
fp32, 6 = the six largest of them (library default). Bar: the path's fp32 tolerance (1e-4 relative, BASELINE.json
north_star) against the fp32 oracle, and -- the reason the split forms may stand in for the fp32 MFMA -- an error
against the fp64 oracle that is no larger than the fp32 MFMA's own on the same inputs."""
import numpy as np
import pytest
import torch

from test_gpu_a_parity import random_sites

pytestmark = pytest.mark.gpu
MODES = [0, 6, 9]


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def restore_mode():
    from vista_amd import ops
    old = ops.set_f32_products(6)
    ops.set_f32_products(old)
    yield
    ops.set_f32_products(old)


def test_default_mode_and_argument_check():
    from vista_amd import ops
    from vista_amd._lib import VxError, lib
    import os
    want = int(os.environ.get("VX_F32_PRODUCTS", "6"))
    assert lib().vx_spconv_get_f32_products() == want
    with pytest.raises(VxError, match="0 .fp32 MFMA., 6 or 9"):
        ops.set_f32_products(7)
    assert lib().vx_spconv_get_f32_products() == want


def _wide(rng, shape):
    """values over ~12 binades with both signs: every plane of the split carries weight"""
    return (rng.normal(size=shape) * np.exp(rng.normal(size=shape) * 2.0)).astype(np.float32)


@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 128), (128, 128), (128, 64), (128, 32), (64, 24), (128, 100)])
def test_modes_vs_fp32_and_fp64_oracle(cin, cout, dev):
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(cin * 131 + cout)
    shape = [9, 40, 40]
    coords = random_sites(rng, 2, shape, 2600)
    n = coords.shape[0]
    feat = _wide(rng, (n, cin))
    w = _wide(rng, (cout, 3, 3, 3, cin)) / np.float32(np.sqrt(27 * cin))
    bias = rng.normal(size=(cout,)).astype(np.float32)
    nbr = so.rulebook_subm(coords, shape, 3)
    ref32 = so.conv_fwd(feat, w, nbr, bias)
    ref64 = so.conv_fwd(feat, w, nbr, bias, dtype=np.float64)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    fd, wd, bd = (torch.from_numpy(a).to(dev) for a in (feat, w, bias))
    err = {}
    for mode in MODES:
        ops.set_f32_products(mode)
        got = ops.conv_fwd(fd, wd, rb, bias=bd, compact=False).cpu().numpy()        # compact=False: the row-stationary kernel
        assert np.abs(got - ref32).max() <= 1e-4 * np.abs(ref32).max(), mode
        e = np.abs(got.astype(np.float64) - ref64)
        err[mode] = (e.max(), np.sqrt((e ** 2).mean()))
        again = ops.conv_fwd(fd, wd, rb, bias=bd, compact=False).cpu().numpy()
        assert np.array_equal(got, again)                                            # run-to-run reproducible
    # the split forms are fp32 arithmetic: no worse against the truth than the fp32 MFMA (10 % slack on the maximum,
    # which is one element's luck; the rms is the stable figure)
    for mode in (6, 9):
        assert err[mode][1] <= 1.05 * err[0][1], (mode, err)
        assert err[mode][0] <= 1.5 * err[0][0], (mode, err)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("ks,st,pd", [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0)),
                                      ((1, 1, 1), (1, 1, 1), (0, 0, 0))])
def test_modes_strided_epilogue_and_dgrad(ks, st, pd, mode, dev):
    """strided rulebooks (few neighbours per row: most chunks skipped), the folded-BN / residual / ReLU epilogue, and the
    input gradient (the same kernel on the transposed weights, mirrored table for submanifold layers)."""
    from oracle import spconv_oracle as so
    from vista_amd import ops
    ops.set_f32_products(mode)
    rng = np.random.default_rng(17)
    shape = [11, 30, 34]
    cin, cout = 128, 64
    coords = random_sites(rng, 2, shape, 3100)
    feat = _wide(rng, (coords.shape[0], cin))
    w = _wide(rng, (cout, *ks, cin)) / np.float32(np.sqrt(np.prod(ks) * cin))
    oc, nbr, osh = so.rulebook_conv(coords, shape, ks, st, pd)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(oc.shape[0], cout)).astype(np.float32)
    ref = np.maximum(so.conv_fwd(feat, w, nbr, None) * sc + sh + res, 0)
    rb = ops.rulebook_conv(torch.from_numpy(coords).to(dev), 2, shape, ks, st, pd, transpose=True)
    fd, wd = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
    got = ops.conv_fwd(fd, wd, rb, scale=torch.from_numpy(sc).to(dev), shift=torch.from_numpy(sh).to(dev),
                       residual=torch.from_numpy(res).to(dev), relu=True, compact=False).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
    dout = _wide(rng, (oc.shape[0], cout))
    ref_din = so.conv_dgrad(dout, w, nbr, coords.shape[0])
    din = ops.conv_dgrad(torch.from_numpy(dout).to(dev), wd, rb).cpu().numpy()
    assert np.abs(din - ref_din).max() <= 1e-4 * np.abs(ref_din).max()
    # weight gradient (reduction over the pairs, 16 per MFMA in the split forms): every mode against fp32 and fp64
    ref_dw = so.conv_wgrad(feat, dout, nbr, w.shape)
    ref_dw64 = so.conv_wgrad(feat, dout, nbr, w.shape, dtype=np.float64)
    dw = ops.conv_wgrad(fd, torch.from_numpy(dout).to(dev), rb, w.shape).cpu().numpy()
    assert np.abs(dw - ref_dw).max() <= 1e-4 * np.abs(ref_dw).max()
    assert np.abs(dw - ref_dw64).max() <= 2e-5 * np.abs(ref_dw64).max()
    # submanifold dgrad (mirrored table)
    nbr_s = so.rulebook_subm(coords, shape, 3)
    ws = _wide(rng, (cin, 3, 3, 3, cin)) / np.float32(np.sqrt(27 * cin))
    dos = _wide(rng, (coords.shape[0], cin))
    rbs = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    din_s = ops.conv_dgrad(torch.from_numpy(dos).to(dev), torch.from_numpy(ws).to(dev), rbs).cpu().numpy()
    ref_s = so.conv_dgrad(dos, ws, nbr_s, coords.shape[0])
    assert np.abs(din_s - ref_s).max() <= 1e-4 * np.abs(ref_s).max()


@pytest.mark.parametrize("mode", [6, 9])
@pytest.mark.parametrize("n_sites", [1, 31, 300, 66900, 76000])
def test_split_forms_edge_sizes_tails_and_device_counts(n_sites, mode, dev):
    """one row, a partial wave, partial tiles, the split last round (66900 / 76000 rows: 262 / 297 tiles on 256 CUs) --
    bitwise equal with and without the tail split and with the row count on the device; 1e-4 against the oracle."""
    from oracle import spconv_oracle as so
    from vista_amd import ops, ops_dn
    ops.set_f32_products(mode)
    rng = np.random.default_rng(n_sites + mode)
    shape = [11, 96, 96]
    coords = random_sites(rng, 2, shape, n_sites)
    n = coords.shape[0]
    cin = cout = 128
    feat = _wide(rng, (n, cin))
    w = _wide(rng, (cout, 3, 3, 3, cin)) / np.float32(np.sqrt(27 * cin))
    cd, fd, wd = (torch.from_numpy(a).to(dev) for a in (coords, feat, w))
    rb = ops.rulebook_subm(cd, 2, shape, 3)
    got = ops.conv_fwd(fd, wd, rb, relu=True)
    plain = ops.conv_fwd(fd, wd, rb, relu=True, split_tail=False)
    assert torch.equal(got, plain)
    sub = min(n, 3000)
    nbr = so.rulebook_subm(coords, shape, 3)[:, :sub]
    ref = np.maximum(so.conv_fwd(feat, w, nbr), 0)
    assert np.abs(got[:sub].cpu().numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-30)
    cap = n + 555
    counts = torch.tensor([n], dtype=torch.int64, device=dev)
    coords_c = torch.full((cap, 4), 12345, dtype=torch.int32, device=dev)
    coords_c[:n] = cd
    feat_c = torch.full((cap, cin), float("nan"), dtype=torch.float32, device=dev)
    feat_c[:n] = fd
    rb_d = ops_dn.rulebook_subm_dn(coords_c, counts, 2, shape, 3)
    out_d = ops_dn.conv_fwd_dn(feat_c, wd, rb_d, relu=True)
    assert torch.equal(out_d[:n], got)


@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 32), (64, 64), (64, 128), (128, 64), (128, 128), (16, 32), (48, 64)])
def test_wgrad_modes_vs_fp64(cin, cout, dev):
    """weight gradient over a rulebook with uneven pair counts per offset (one chunk boundary inside): the split forms
    are no further from the fp64 oracle than the fp32 MFMA; widths that do not fill 32-channel tiles keep the fp32 form."""
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(cin + 7 * cout)
    shape = [9, 60, 60]
    coords = random_sites(rng, 2, shape, 9000)
    n = coords.shape[0]
    feat, dout = _wide(rng, (n, cin)), _wide(rng, (n, cout))
    nbr = so.rulebook_subm(coords, shape, 3)
    wshape = (cout, 3, 3, 3, cin)
    ref64 = so.conv_wgrad(feat, dout, nbr, wshape, dtype=np.float64)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    fd, dd = torch.from_numpy(feat).to(dev), torch.from_numpy(dout).to(dev)
    rms = {}
    for mode in MODES:
        ops.set_f32_products(mode)
        dw = ops.conv_wgrad(fd, dd, rb, wshape)
        assert torch.equal(dw, ops.conv_wgrad(fd, dd, rb, wshape))
        e = dw.cpu().numpy().astype(np.float64) - ref64
        assert np.abs(e).max() <= 2e-5 * np.abs(ref64).max(), mode
        rms[mode] = np.sqrt((e ** 2).mean())
    for mode in (6, 9):
        assert rms[mode] <= 1.05 * rms[0], rms


@pytest.mark.parametrize("mode", [0, 9])
def test_backbone_forward_in_the_other_modes(mode, dev):
    """SpMiddleResNetFHD end to end against the oracle in the two non-default modes (the default runs in
    test_gpu_a_parity.test_backbone_forward_vs_oracle and the full-size tests)."""
    from test_gpu_a_parity import test_backbone_forward_vs_oracle
    from vista_amd import ops
    ops.set_f32_products(mode)
    test_backbone_forward_vs_oracle("SpMiddleResNetFHD", dev)


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (32, 32)])
def test_per_call_arithmetic_flag_overrides_the_process_default(cin, cout, dev):
    """VX_CONV_F32_* in a call's flags (vx_spconv_fwd / _dn / dgrad / wgrad) selects that call's arithmetic whatever the
    process default is: bitwise the result of the same call under the matching default."""
    from vista_amd import ops, ops_dn
    rng = np.random.default_rng(cin + 7 * cout)
    shape = [9, 40, 40]
    coords = torch.from_numpy(random_sites(rng, 2, shape, 2600)).to(dev)
    n = coords.shape[0]
    feat = torch.from_numpy(_wide(rng, (n, cin))).to(dev)
    w = torch.from_numpy(_wide(rng, (cout, 3, 3, 3, cin)) / np.float32(np.sqrt(27 * cin))).to(dev)
    gout = torch.from_numpy(_wide(rng, (n, cout))).to(dev)
    rb = ops.rulebook_subm(coords, 2, shape, 3)
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    rbd = ops_dn.rulebook_subm_dn(coords, cnt, 2, shape, 3)
    want = {}
    for mode in MODES:
        ops.set_f32_products(mode)
        want[mode] = (ops.conv_fwd(feat, w, rb), ops.conv_dgrad(gout, w, rb), ops.conv_wgrad(feat, gout, rb, w.shape))
    for default in MODES:
        ops.set_f32_products(default)
        for mode in MODES:
            got = (ops.conv_fwd(feat, w, rb, products=mode), ops.conv_dgrad(gout, w, rb, products=mode),
                   ops.conv_wgrad(feat, gout, rb, w.shape, products=mode))
            for g, r in zip(got, want[mode]):
                assert torch.equal(g, r), (default, mode)
            assert torch.equal(ops_dn.conv_fwd_dn(feat, w, rbd, products=mode), want[mode][0])
    if cin >= 64:      # the three arithmetics do differ at these widths (otherwise the test above proves nothing)
        assert not torch.equal(want[0][0], want[6][0])
    with pytest.raises(ValueError):
        ops.conv_fwd(feat, w, rb, products=7)
    # wgrad into a caller-provided destination (a slice of a flat gradient buffer): same bits, destination overwritten
    flat = torch.full((w.numel() + 64,), float("nan"), device=dev)
    dst = flat[32:32 + w.numel()].view_as(w)
    out = ops.conv_wgrad(feat, gout, rb, w.shape, products=6, out=dst)
    assert out.data_ptr() == dst.data_ptr() and torch.equal(dst, want[6][2])
    assert bool(torch.isnan(flat[:32]).all()) and bool(torch.isnan(flat[32 + w.numel():]).all())


def test_engine_arithmetic_is_an_engine_property(dev):
    """Two engines with different f32_products in one process, interleaved: each keeps its own arithmetic (captured
    plans included) while the process default is changed under them."""
    from vista_amd import models, ops
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import GEOMETRIES, backbone_state_dict, grid_size, lidar_cloud
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES["SMALL"]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    net = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5),
                                       backbone_state_dict("SpMiddleResNetFHD")).to(dev).eval()
    fr = [torch.from_numpy(lidar_cloud(seed=41, sweeps=3)).to(dev)]
    ref = {}
    for mode in (0, 6):
        ops.set_f32_products(mode)
        ref[mode] = FusedBackbone(net, eager=True).forward_frames(gen, fr, grid).clone()
    assert not torch.equal(ref[0], ref[6])
    e0 = FusedBackbone(net, dispatch="graph", f32_products=0)
    e6 = FusedBackbone(net, dispatch="graph", f32_products=6)
    for default in (6, 0, 9, 6):
        ops.set_f32_products(default)
        assert torch.equal(e0.forward_frames(gen, fr, grid), ref[0])
        assert torch.equal(e6.forward_frames(gen, fr, grid), ref[6])
    assert len(e0._plans) == 1 and len(e6._plans) == 1
