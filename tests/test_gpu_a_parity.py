"""GPU parity tests proper: every call goes through the C-ABI (libvx.so via ctypes) and is checked
against the CPU oracle (oracle/) and the committed golden vectors (tests/golden/)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
MAN = json.load(open(os.path.join(G, "voxel_manifest.json")))
FULL = [k for k in MAN if not k.startswith("_") and os.path.exists(os.path.join(G, f"voxel_{k}.npz"))]
BIG = [k for k in MAN if not k.startswith("_") and os.path.exists(os.path.join(G, f"voxel_{k}_sample.npz"))]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


# ---------------------------------------------------------------------------- voxelization
@pytest.mark.parametrize("name", FULL)
def test_voxelize_golden_full(name, dev):
    from vista_amd import ops
    m = MAN[name]
    d = np.load(os.path.join(G, f"voxel_{name}.npz"))
    pts = torch.from_numpy(d["points"]).to(dev)
    r = ops.voxelize(pts, m["voxel_size"], m["range"], m["max_points"], m["max_voxels"], batch_idx=3,
                     with_voxels=True, with_mean=True)
    assert r["num_voxels"] == d["coords"].shape[0]
    c = r["coords"].cpu().numpy()
    np.testing.assert_array_equal(c[:, 1:], d["coords"])            # bit-exact coordinates, reference order
    assert np.all(c[:, 0] == 3)
    np.testing.assert_array_equal(r["num_points"].cpu().numpy(), d["num_points"])
    np.testing.assert_array_equal(r["voxels"].cpu().numpy(), d["voxels"])
    if c.shape[0]:
        # fp32 sums of <= T values: tolerance only covers summation order (voxel_encoder.py:206-211)
        np.testing.assert_allclose(r["mean"].cpu().numpy(), d["mean"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", BIG)
def test_voxelize_golden_big(name, dev):
    from vista_amd import ops
    from vista_amd.synth import lidar_cloud
    m = MAN[name]
    pts_np = lidar_cloud(seed=m["seed"], sweeps=m["sweeps"])
    assert sha(pts_np) == MAN["_points_sha256"][f"seed{m['seed']}_sweeps{m['sweeps']}"]
    r = ops.voxelize(torch.from_numpy(pts_np).to(dev), m["voxel_size"], m["range"], m["max_points"],
                     m["max_voxels"], with_voxels=True, with_mean=True)
    assert r["num_voxels"] == m["M"]
    assert sha(r["coords"][:, 1:].contiguous().cpu().numpy()) == m["coords_sha256"]
    assert sha(r["num_points"].cpu().numpy()) == m["num_points_sha256"]
    assert sha(r["voxels"].cpu().numpy()) == m["voxels_sha256"]


def test_voxelize_vs_oracle_random_and_determinism(dev):
    from oracle import voxelize_np as ov
    from vista_amd import ops
    rng = np.random.default_rng(5)
    pts = rng.uniform(-3, 3, (50000, 5)).astype(np.float32)
    vs, rg = [0.25, 0.25, 0.5], [-2.0, -2.0, -1.0, 2.0, 2.0, 1.0]
    v, c, n = ov.voxelize(pts, vs, rg, 4, 100000)
    t = torch.from_numpy(pts).to(dev)
    r1 = ops.voxelize(t, vs, rg, 4, 100000, with_voxels=True)
    r2 = ops.voxelize(t, vs, rg, 4, 100000, with_voxels=True)
    np.testing.assert_array_equal(r1["coords"][:, 1:].cpu().numpy(), c)
    np.testing.assert_array_equal(r1["num_points"].cpu().numpy(), n)
    np.testing.assert_array_equal(r1["voxels"].cpu().numpy(), v)
    np.testing.assert_allclose(r1["mean"].cpu().numpy(), ov.mean_vfe(v, n), rtol=1e-6, atol=1e-6)
    for k in ("coords", "num_points", "voxels", "mean"):
        assert torch.equal(r1[k], r2[k])  # run-to-run bitwise identical


def test_voxel_generator_api(dev):
    from vista_amd.voxel import VoxelGenerator, Voxelization, collate_voxels
    d = np.load(os.path.join(G, "voxel_tiny_1sweep.npz"))
    m = MAN["tiny_1sweep"]
    gen = VoxelGenerator(m["voxel_size"], m["range"], m["max_points"], max_voxels=m["max_voxels"])
    assert gen.grid_size.dtype == np.int64 and list(gen.grid_size) == m["grid_size"]
    v, c, n = gen.generate(torch.from_numpy(d["points"]).to(dev))
    np.testing.assert_array_equal(c.cpu().numpy(), d["coords"])
    stage = Voxelization(cfg=dict(range=m["range"], voxel_size=m["voxel_size"],
                                  max_points_in_voxel=m["max_points"], max_voxel_num=m["max_voxels"]))
    res, _ = stage(dict(lidar=dict(points=torch.from_numpy(d["points"]).to(dev)), mode="val"), None)
    vd = res["lidar"]["voxels"]
    assert set(vd) == {"voxels", "coordinates", "num_points", "num_voxels", "shape"}
    b = collate_voxels([vd, vd])
    cc = b["coordinates"].cpu().numpy()
    assert cc.shape == (2 * d["coords"].shape[0], 4)
    np.testing.assert_array_equal(cc[d["coords"].shape[0]:, 0], 1)
    np.testing.assert_array_equal(cc[:d["coords"].shape[0], 1:], d["coords"])


def test_voxelization_double_flip_stage(dev):
    """preprocess.py:210-254: three more voxelizations of the flipped clouds (test_aug.py:53-77 flips);
    each dict must equal the oracle's voxelization of that cloud, bit for bit."""
    from oracle import voxelize_np as ov
    from vista_amd.voxel import Voxelization
    d = np.load(os.path.join(G, "voxel_tiny_1sweep.npz"))
    m = MAN["tiny_1sweep"]
    pts = d["points"]
    clouds = {"points": pts}
    for key, sx, sy in (("yflip", 1, -1), ("xflip", -1, 1), ("double_flip", -1, -1)):
        q = pts.copy()
        q[:, 0] *= sx
        q[:, 1] *= sy
        clouds[f"{key}_points"] = q
    stage = Voxelization(cfg=dict(range=m["range"], voxel_size=m["voxel_size"], max_points_in_voxel=m["max_points"],
                                  max_voxel_num=m["max_voxels"], double_flip=True))
    from vista_amd.voxel import DoubleFlip
    res, _ = DoubleFlip()(dict(lidar=dict(points=torch.from_numpy(pts).to(dev)), mode="val"), None)   # GPU flips
    for k, v in clouds.items():
        np.testing.assert_array_equal(res["lidar"][k].cpu().numpy(), v)
    res, _ = stage(res, None)
    for key in ("", "yflip_", "xflip_", "double_flip_"):
        vd = res["lidar"][f"{key}voxels"]
        v, c, n = ov.voxelize(clouds[f"{key}points"], m["voxel_size"], m["range"], m["max_points"], m["max_voxels"])
        np.testing.assert_array_equal(vd["coordinates"].cpu().numpy(), c)
        np.testing.assert_array_equal(vd["num_points"].cpu().numpy(), n)
        np.testing.assert_array_equal(vd["voxels"].cpu().numpy(), v)
        assert vd["num_voxels"].dtype == np.int64 and vd["num_voxels"][0] == c.shape[0]
        assert ("range" in vd) == bool(key) and ("size" in vd) == bool(key)
    # training mode never takes the flip branch (preprocess.py:181)
    res, _ = stage(dict(lidar=dict(points=torch.from_numpy(pts).to(dev)), mode="train"), None)
    assert "yflip_voxels" not in res["lidar"]


# ---------------------------------------------------------------------------- rulebooks
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


GEOMS = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
         ((3, 1, 1), (2, 1, 1), (0, 0, 0)), ((1, 1, 3), (1, 1, 2), (0, 0, 1))]


@pytest.mark.parametrize("ks,st,pd", GEOMS)
def test_rulebook_conv_bit_exact(ks, st, pd, dev):
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(0)
    shape = [21, 64, 72]
    coords = random_sites(rng, 3, shape, 5000)
    oc, nbr, osh = so.rulebook_conv(coords, shape, ks, st, pd)
    rb = ops.rulebook_conv(torch.from_numpy(coords).to(dev), 3, shape, ks, st, pd, transpose=True)
    assert rb.out_shape == osh and rb.n_out == oc.shape[0]
    np.testing.assert_array_equal(rb.out_indices.cpu().numpy(), oc)
    np.testing.assert_array_equal(rb.nbr.cpu().numpy(), nbr)
    # pair-list view, bit-exact against the oracle's
    pairs, num = so.nbr_to_pairs(nbr)
    gp, gn = rb.pairs()
    np.testing.assert_array_equal(gn.cpu().numpy(), num)
    np.testing.assert_array_equal(gp.cpu().numpy(), pairs)
    # transpose table
    nt = rb.nbr_t.cpu().numpy()
    kk, oo = np.nonzero(nbr >= 0)
    ref_t = np.full_like(nt, -1)
    ref_t[kk, nbr[kk, oo]] = oo
    np.testing.assert_array_equal(nt, ref_t)


def test_rulebook_subm_bit_exact(dev):
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(1)
    shape = [11, 50, 60]
    coords = random_sites(rng, 2, shape, 6000)
    nbr = so.rulebook_subm(coords, shape, 3)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    np.testing.assert_array_equal(rb.nbr.cpu().numpy(), nbr)
    pairs, num = so.nbr_to_pairs(nbr)
    gp, gn = rb.pairs()
    np.testing.assert_array_equal(gn.cpu().numpy(), num)
    np.testing.assert_array_equal(gp.cpu().numpy(), pairs)


def test_rulebook_empty(dev):
    from vista_amd import ops
    e = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    rb = ops.rulebook_conv(e, 1, [5, 6, 7], 3, 2, 1)
    assert rb.n_out == 0 and rb.nbr.shape == (27, 0)
    rb = ops.rulebook_subm(e, 1, [5, 6, 7], 3)
    assert rb.nbr.shape == (27, 0)


# ---------------------------------------------------------------------------- convolution
CH = [(5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 32), (7, 9)]


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("cin,cout", CH)
def test_conv_fwd_subm_vs_oracle(cin, cout, compact, dev):
    """fp32 tolerance: 1e-4 relative (north_star); observed error is summation-order only. compact=True is the
    library's default choice (per-offset compaction kernel where eligible), False forces the row-stationary kernels."""
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(cin * 1000 + cout)
    shape = [9, 40, 40]
    coords = random_sites(rng, 2, shape, 1500)
    n = coords.shape[0]
    feat = rng.normal(size=(n, cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    nbr = so.rulebook_subm(coords, shape, 3)
    ref = so.conv_fwd(feat, w, nbr, bias)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    got = ops.conv_fwd(torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev), rb,
                       bias=torch.from_numpy(bias).to(dev), compact=compact).cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-4 * scale, np.abs(got - ref).max() / scale


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("ks,st,pd", GEOMS)
def test_conv_fwd_strided_and_epilogue(ks, st, pd, compact, dev):
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(3)
    shape = [11, 30, 34]
    cin, cout = 32, 64
    coords = random_sites(rng, 2, shape, 2500)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    w = (rng.normal(size=(cout, *ks, cin)) / np.sqrt(np.prod(ks) * cin)).astype(np.float32)
    oc, nbr, osh = so.rulebook_conv(coords, shape, ks, st, pd)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(oc.shape[0], cout)).astype(np.float32)
    ref = np.maximum(so.conv_fwd(feat, w, nbr, None) * sc + sh + res, 0)
    rb = ops.rulebook_conv(torch.from_numpy(coords).to(dev), 2, shape, ks, st, pd)
    got = ops.conv_fwd(torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev), rb,
                       scale=torch.from_numpy(sc).to(dev), shift=torch.from_numpy(sh).to(dev),
                       residual=torch.from_numpy(res).to(dev), relu=True, compact=compact).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("subm", [True, False])
@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 32), (32, 16), (64, 64), (128, 64), (128, 128), (128, 32)])
def test_conv_grads_vs_oracle(subm, cin, cout, dev):
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(11)
    shape = [9, 36, 36]
    coords = random_sites(rng, 2, shape, 3000)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    if subm:
        nbr = so.rulebook_subm(coords, shape, 3)
        rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    else:
        oc, nbr, osh = so.rulebook_conv(coords, shape, 3, 2, 1)
        rb = ops.rulebook_conv(torch.from_numpy(coords).to(dev), 2, shape, 3, 2, 1, transpose=True)
    dout = rng.normal(size=(nbr.shape[1], cout)).astype(np.float32)
    din_ref = so.conv_dgrad(dout, w, nbr, coords.shape[0])
    dw_ref = so.conv_wgrad(feat, dout, nbr, w.shape)
    din = ops.conv_dgrad(torch.from_numpy(dout).to(dev), torch.from_numpy(w).to(dev), rb).cpu().numpy()
    dw = ops.conv_wgrad(torch.from_numpy(feat).to(dev), torch.from_numpy(dout).to(dev), rb, w.shape).cpu().numpy()
    assert np.abs(din - din_ref).max() <= 1e-4 * np.abs(din_ref).max()
    assert np.abs(dw - dw_ref).max() <= 1e-4 * np.abs(dw_ref).max()


def test_to_dense(dev):
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(4)
    shape = [2, 45, 50]
    coords = random_sites(rng, 3, shape, 700)
    lin = ((coords[:, 0].astype(np.int64) * 2 + coords[:, 1]) * 45 + coords[:, 2]) * 50 + coords[:, 3]
    coords = coords[np.argsort(lin)]
    feat = rng.normal(size=(coords.shape[0], 128)).astype(np.float32)
    ref = so.to_dense(feat, coords, 3, shape)
    got = ops.to_dense(torch.from_numpy(feat).to(dev), torch.from_numpy(coords).to(dev), 3, shape).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_ops_reject_cpu_tensors():
    from vista_amd import ops
    from vista_amd._lib import VxError
    with pytest.raises(VxError):
        ops.voxelize(torch.zeros(4, 5), [1, 1, 1], [0, 0, 0, 1, 1, 1], 3, 10)


# ---------------------------------------------------------------------------- whole backbone
def _tiny_frames(dev, batch):
    from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
    from vista_amd.voxel import VoxelGenerator, voxelize_batch
    rg, vs, T, mv = GEOMETRIES["SMALL"]
    gen = VoxelGenerator(vs, rg, T, mv)
    clouds = [lidar_cloud(seed=20 + b, sweeps=2) for b in range(batch)]
    mean, coords, nump = voxelize_batch(gen, [torch.from_numpy(c).to(dev) for c in clouds])
    return clouds, mean, coords, grid_size(rg, vs).tolist(), (rg, vs, T, mv)


@pytest.mark.parametrize("kind", ["SpMiddleResNetFHD", "MultiViewBackbone"])
def test_backbone_forward_vs_oracle(kind, dev):
    """Voxelize + mean VFE + backbone on 2 small synthetic frames: module path (spconv API, torch
    BN/ReLU) and fused engine, both against the CPU oracle. fp32 tolerance 1e-4 relative."""
    from oracle import backbone_oracle as bo
    from oracle import voxelize_np as ov
    from vista_amd import models
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import backbone_state_dict
    clouds, mean, coords, grid, (rg, vs, T, mv) = _tiny_frames(dev, 2)
    # oracle front-end
    ovx = [ov.voxelize(c, vs, rg, T, mv) for c in clouds]
    o_feat = np.concatenate([ov.mean_vfe(v, n) for v, c, n in ovx])
    o_coords = ov.collate_coords([c for v, c, n in ovx])
    np.testing.assert_array_equal(coords.cpu().numpy(), o_coords)
    np.testing.assert_allclose(mean.cpu().numpy(), o_feat, rtol=1e-6, atol=1e-6)
    sd = backbone_state_dict(kind)
    net = getattr(models, kind)(num_input_features=5)
    models.load_numpy_state_dict(net, sd).to(dev).eval()
    if kind == "SpMiddleResNetFHD":
        r0, t0 = bo.sp_middle_resnet_fhd(o_feat, o_coords, 2, grid, sd)
        ref, masks = [r0], [bo.active_mask(t0)]
    else:
        bev, rv, tb, tr = bo.multi_view_backbone(o_feat, o_coords, 2, grid, sd, return_tensors=True)
        ref, masks = [bev, rv], [bo.active_mask(tb), bo.active_mask(tr, "rv")]
    with torch.no_grad():
        out_mod = net(mean, coords, 2, grid)
    out_fused = FusedBackbone(net)(mean, coords, 2, grid)
    if kind == "SpMiddleResNetFHD":
        out_mod, out_fused = [out_mod], [out_fused]
    for r, act, a, b in zip(ref, masks, out_mod, out_fused):
        assert tuple(a.shape) == r.shape == tuple(b.shape) == act.shape
        scale = np.abs(r).max()
        for got in (a.cpu().numpy(), b.cpu().numpy()):
            assert np.abs(got - r).max() <= 1e-4 * scale
            # identical active set: exact zeros everywhere outside the oracle's active sites ...
            assert not (got[~act] != 0).any()
            # ... and inside them the non-zero masks may differ only where BOTH values sit at the ReLU boundary
            # (a true zero in one, rounding noise in the other)
            diff = (got != 0) != (r != 0)
            assert np.abs(got[diff]).max(initial=0.0) <= 1e-4 * scale and np.abs(r[diff]).max(initial=0.0) <= 1e-4 * scale


def _oracle_train_step(sd, mean, coords, grid, numel, dtype, gates=None, impose=None, slack=None):
    """fwd + loss = sum(x^2)/numel + bwd of SpMiddleResNetFHD (train-mode BatchNorm) in torch CPU autograd whose
    sparse convolutions are the ORACLE's conv fwd/dgrad/wgrad in `dtype` (torch.float32: the reference's precision;
    torch.float64: the truth). -> (loss, {param name: grad ndarray}). `gates` (a list) receives, per ReLU in call
    order, (names of the conv weights whose output it gates, bool mask of open gates). `impose` (bool masks in the
    same order, e.g. the GPU run's) replaces every ReLU by a multiplication with its mask: the step becomes a smooth
    function and its gradients comparable to rounding accuracy at every depth; `slack` then receives, per ReLU, the
    largest |pre-activation| at a position where the mask disagrees with the sign, relative to the largest in the
    layer (a disagreement is legitimate only within rounding of zero)."""
    from oracle import spconv_oracle as so
    from vista_amd import models
    npdt = np.float32 if dtype == torch.float32 else np.float64

    class OConv(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, nbr):
            ctx.nbr, ctx.n_in = nbr, x.shape[0]
            ctx.save_for_backward(x, w)
            return torch.from_numpy(so.conv_fwd(x.numpy(), w.numpy(), nbr, dtype=npdt))

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            g = g.contiguous().numpy()
            return (torch.from_numpy(so.conv_dgrad(g, w.numpy(), ctx.nbr, ctx.n_in, dtype=npdt)),
                    torch.from_numpy(so.conv_wgrad(x.numpy(), g, ctx.nbr, tuple(w.shape), dtype=npdt)), None)

    ref = models.SpMiddleResNetFHD(num_input_features=5)
    models.load_numpy_state_dict(ref, sd).train().to(dtype)
    x = mean.cpu().to(dtype)
    c = coords.cpu().numpy()
    shape = [int(v) for v in (np.array(grid[::-1]) + [1, 0, 0])]
    cache = {}

    def run_conv(conv, x, c, shape):
        if conv.subm:
            key = (conv.indice_key, c.shape[0])
            if key not in cache:
                cache[key] = so.rulebook_subm(c, shape, conv.kernel_size)
            y = OConv.apply(x, conv.weight, cache[key])
            oc, osh = c, shape
        else:
            oc, nbr, osh = so.rulebook_conv(c, shape, conv.kernel_size, conv.stride, conv.padding)
            y = OConv.apply(x, conv.weight, nbr)
        if conv.bias is not None:
            y = y + conv.bias
        return y, oc, osh

    def relu(t, name):
        if impose is not None:
            mask = torch.from_numpy(impose[len(slack)])
            td = t.detach()
            wrong = (td > 0) != mask
            slack.append(float(td[wrong].abs().max() / td.abs().max()) if bool(wrong.any()) else 0.0)
            return t * mask.to(t.dtype)
        y = torch.relu(t)
        if gates is not None:
            gates.append((name, (y > 0).numpy()))
        return y

    pending = None
    for i, m in enumerate(ref.middle_conv):
        if isinstance(m, models.SparseBasicBlock):
            idt = x
            y, _, _ = run_conv(m.conv1, x, c, shape)
            y = relu(m.bn1(y), f"middle_conv.{i}.conv1.weight")
            y, _, _ = run_conv(m.conv2, y, c, shape)
            x = relu(m.bn2(y) + idt, f"middle_conv.{i}.conv2.weight")
        elif hasattr(m, "subm"):
            x, c, shape = run_conv(m, x, c, shape)
            pending = f"middle_conv.{i}.weight"
        elif isinstance(m, torch.nn.ReLU):
            x = relu(x, pending)
        else:
            x = m(x)
    # loss over the dense tensor == sum over active rows / numel(dense)
    loss = (x ** 2).sum() / numel
    loss.backward()
    return float(loss.item()), {n: p.grad.numpy().astype(np.float64) for n, p in ref.named_parameters()}


def _gpu_train_step(sd, mean, coords, grid, dev, fused=False):
    """The same step on the GPU (module path; fused=True: libvx BatchNorm kernels) -> (net, out, loss, gates)."""
    from vista_amd import models
    from vista_amd.fused_bn import FusedBatchNorm1d, fuse_for_training
    net = models.SpMiddleResNetFHD(num_input_features=5)
    models.load_numpy_state_dict(net, sd).to(dev).train()
    if fused:
        fuse_for_training(net)
    gates = []
    hooks = []
    for m in net.modules():
        if isinstance(m, torch.nn.ReLU) or (isinstance(m, FusedBatchNorm1d) and m.relu):
            hooks.append(m.register_forward_hook(lambda mod, inp, out: gates.append((out > 0).cpu().numpy())))
    out = net(mean, coords, 2, grid)
    loss = (out ** 2).mean()
    loss.backward()
    for h in hooks:
        h.remove()
    return net, out, loss, gates


def _check_grads_against_fp64(net, gates, gates64, g32, g64, gates32=None):
    """A weight gradient depends on the ReLU gates of its own layer and of every layer above it. Where the GPU run's
    gates all agree with the fp64 run's, the gradient must match fp64 to 1e-4 (fp32 rounding only). A gate that
    flips (a pre-activation within rounding of zero: fp32 and fp64 land on different sides) is a discontinuity, not
    an error: below the first flip only direction and size are checked -- the fp32 CPU oracle shows the same
    step-like behaviour at ITS first flip (tools/grad_error_table.py). The deep stages are compared with the fp32
    oracle itself as well (1e-3), under the same rule: only where the GPU's gates agree with the fp32 oracle's."""
    assert len(gates) == len(gates64) == 21

    def flips(ref_gates):              # conv weight name -> gates flipped at its ReLU or any later one
        out, later = {}, 0
        for (name, mref), m in zip(ref_gates[::-1], gates[::-1]):
            later += int((mref != m).sum())
            out[name] = later
        return out
    flips_from = flips(gates64)
    flips32 = flips(gates32) if gates32 is not None else None
    tight = 0
    for n, p in net.named_parameters():
        if not (n.endswith("weight") and p.dim() == 5):
            continue
        g = p.grad.cpu().numpy().astype(np.float64)
        scale = np.abs(g64[n]).max()
        err = np.abs(g - g64[n]).max() / scale
        cos = float(g.ravel() @ g64[n].ravel() / (np.linalg.norm(g) * np.linalg.norm(g64[n])))
        if flips_from[n] == 0:
            assert err <= 1e-4, (n, err)
            tight += 1
        else:
            assert cos >= 0.9995 and err <= 5e-2, (n, cos, err, flips_from[n])
        if int(n.split(".")[1]) >= 15 and flips32 is not None and flips32[n] == 0:
            assert np.abs(g - g32[n]).max() / np.abs(g32[n]).max() <= 1e-3, n
    return tight


def _check_grads_with_gpu_gates_imposed(net, gates, sd, mean, coords, grid, numel):
    """The strong form: the fp64 oracle step with the GPU run's own ReLU gate pattern imposed. Every conv weight
    gradient must then match to 1e-4 at EVERY depth, and every gate on which the GPU and fp64 disagree must sit on a
    pre-activation within 1e-5 of zero (relative to its layer): the disagreements are rounding, nothing else."""
    slack = []
    _, g = _oracle_train_step(sd, mean, coords, grid, numel, torch.float64, impose=gates, slack=slack)
    assert len(slack) == 21 and max(slack) <= 1e-5, slack
    worst = 0.0
    for n, p in net.named_parameters():
        if n.endswith("weight") and p.dim() == 5:
            err = np.abs(p.grad.cpu().numpy().astype(np.float64) - g[n]).max() / np.abs(g[n]).max()
            assert err <= 1e-4, (n, err)
            worst = max(worst, err)
    return worst


def test_backbone_train_step_grads_vs_oracle(dev):
    """One fwd+bwd of SpMiddleResNetFHD (train-mode BN, loss = mean(out^2)) through the autograd wiring of the libvx
    dgrad/wgrad kernels, against torch autograd over the ORACLE's convs in fp32 (the oracle proper) and in fp64 (the
    truth). See _check_grads_against_fp64 for the bar: 1e-4 against fp64 wherever the ReLU gate pattern agrees."""
    from vista_amd.synth import backbone_state_dict
    clouds, mean, coords, grid, _ = _tiny_frames(dev, 2)
    sd = backbone_state_dict("SpMiddleResNetFHD")
    net, out, loss, gates = _gpu_train_step(sd, mean, coords, grid, dev)
    numel = out.numel()
    gates64, gates32 = [], []
    loss32, g32 = _oracle_train_step(sd, mean, coords, grid, numel, torch.float32, gates32)
    loss64, g64 = _oracle_train_step(sd, mean, coords, grid, numel, torch.float64, gates64)
    assert abs(loss.item() - loss32) <= 1e-4 * abs(loss32)
    assert abs(loss.item() - loss64) <= 1e-4 * abs(loss64)
    tight = _check_grads_against_fp64(net, gates, gates64, g32, g64, gates32)
    assert tight >= 1, tight             # the last conv at least; how deep the agreement reaches is rounding luck
    _check_grads_with_gpu_gates_imposed(net, gates, sd, mean, coords, grid, numel)


# ---------------------------------------------------------------------------- bf16
@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 64), (64, 64), (128, 128), (128, 32)])
def test_conv_fwd_bf16_vs_oracle(cin, cout, dev):
    """bf16 in/weights/out, fp32 accumulate. Oracle fed the SAME bf16-rounded operands, so the only
    differences are accumulation order and the final bf16 rounding of the output (<= 2^-8 relative)."""
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(cin + cout)
    shape = [9, 40, 40]
    coords = random_sites(rng, 2, shape, 1500)
    feat = torch.from_numpy(rng.normal(size=(coords.shape[0], cin)).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rng.normal(size=(cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)).bfloat16()
    bias = rng.normal(size=(cout,)).astype(np.float32)
    nbr = so.rulebook_subm(coords, shape, 3)
    ref = so.conv_fwd(feat.float().numpy(), w.float().numpy(), nbr, bias)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    got = ops.conv_fwd(feat.to(dev), w.to(dev), rb, bias=torch.from_numpy(bias).to(dev))
    assert got.dtype == torch.bfloat16
    err = np.abs(got.float().cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err <= 5e-3, err


def test_backbone_forward_bf16_vs_oracle(dev):
    """north_star tolerance for bf16: 2e-2 relative against the fp32 CPU oracle."""
    from oracle import backbone_oracle as bo
    from oracle import voxelize_np as ov
    from vista_amd import models
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import backbone_state_dict
    clouds, mean, coords, grid, (rg, vs, T, mv) = _tiny_frames(dev, 2)
    sd = backbone_state_dict("SpMiddleResNetFHD")
    net = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5), sd).to(dev).eval()
    ovx = [ov.voxelize(c, vs, rg, T, mv) for c in clouds]
    ref, _ = bo.sp_middle_resnet_fhd(np.concatenate([ov.mean_vfe(v, n) for v, c, n in ovx]),
                                     ov.collate_coords([c for v, c, n in ovx]), 2, grid, sd)
    out = FusedBackbone(net, dtype=torch.bfloat16)(mean, coords, 2, grid)
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == ref.shape
    err = np.abs(out.float().cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err <= 2e-2, err


def test_mask_sorted_rows_bitwise_identical(dev):
    """vx_rulebook_sort_rows only re-groups rows into MFMA tiles: perm is a permutation that stays
    inside 16384-row windows, the sorted table is the permuted table, and conv_fwd gives the same
    bits with and without it."""
    from vista_amd import ops
    rng = np.random.default_rng(9)
    shape = [11, 120, 120]
    coords = random_sites(rng, 2, shape, 20000)
    n = coords.shape[0]
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    nbr_s, perm = rb.sorted_table()
    p = perm.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(p), np.arange(n))
    assert np.all(p // 16384 == np.arange(n) // 16384)
    nbr = rb.nbr.cpu().numpy()
    np.testing.assert_array_equal(nbr_s.cpu().numpy(), nbr[:, p])
    masks = ((nbr[:, p] >= 0).astype(np.int64) << np.arange(27)[:, None]).sum(0)
    for w0 in range(0, n, 16384):
        assert np.all(np.diff(masks[w0:w0 + 16384]) >= 0)      # sorted by mask inside each window
    for cin, cout, dt in [(64, 64, torch.float32), (128, 128, torch.float32), (32, 64, torch.bfloat16)]:
        x = torch.from_numpy(rng.normal(size=(n, cin)).astype(np.float32)).to(dev).to(dt)
        w = torch.from_numpy((rng.normal(size=(cout, 3, 3, 3, cin)) / 30).astype(np.float32)).to(dev).to(dt)
        res = torch.from_numpy(rng.normal(size=(n, cout)).astype(np.float32)).to(dev).to(dt)
        a = ops.conv_fwd(x, w, rb, residual=res, relu=True, sort_rows=True, compact=False)
        b = ops.conv_fwd(x, w, rb, residual=res, relu=True, sort_rows=False, compact=False)
        assert torch.equal(a, b)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 64), (64, 64), (128, 128), (128, 32)])
def test_conv_fwd_streaming_w_kernel(cin, cout, dev):
    """fp32 streaming-W kernel (packed weights): same summation order as the LDS-staged fp32-MFMA kernel ->
    bitwise equal to it (vx_spconv_set_f32_products(0) selects that arithmetic), and within 1e-4 of the oracle."""
    from oracle import spconv_oracle as so
    from vista_amd import ops
    old = ops.set_f32_products(0)
    try:
        _streaming_w_body(cin, cout, dev, so, ops)
    finally:
        ops.set_f32_products(old)


def _streaming_w_body(cin, cout, dev, so, ops):
    rng = np.random.default_rng(cin * 7 + cout)
    shape = [9, 60, 60]
    coords = random_sites(rng, 2, shape, 5000)
    n = coords.shape[0]
    feat = rng.normal(size=(n, cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    nbr = so.rulebook_subm(coords, shape, 3)
    ref = so.conv_fwd(feat, w, nbr, bias)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    x, wt, b = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(bias).to(dev)
    wp = ops.pack_weight(wt)
    got = ops.conv_fwd(x, wt, rb, bias=b, relu=False, wpack=wp, sort_rows=True)
    base = ops.conv_fwd(x, wt, rb, bias=b, relu=False, compact=False)
    assert np.abs(got.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    assert torch.equal(got, base)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cin,n_sites", [(128, 76000), (128, 9000), (64, 70000)])
def test_conv_fwd_split_tail_bitwise(cin, n_sites, dtype, dev):
    """spconv_w8's split-tail launch (last partial round of 256-row tiles taken by two workgroups per tile, half of
    the output columns each) sums in the same order as the plain grid: bitwise equal. 76000 / 70000 rows = one full
    round of 256 tiles + a tail; 9000 rows = tail only."""
    from vista_amd import ops
    rng = np.random.default_rng(n_sites + cin)
    shape = [5, 200, 200]
    coords = random_sites(rng, 1, shape, n_sites)
    n = coords.shape[0]
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    x = torch.from_numpy(rng.normal(size=(n, cin)).astype(np.float32)).to(dev).to(td)
    wt = torch.from_numpy((rng.normal(size=(128, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev).to(td)
    b = torch.from_numpy(rng.normal(size=(128,)).astype(np.float32)).to(dev)
    res = torch.from_numpy(rng.normal(size=(n, 128)).astype(np.float32)).to(dev).to(td)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 1, shape, 3)
    got = ops.conv_fwd(x, wt, rb, bias=b, residual=res, relu=True, compact=False)
    base = ops.conv_fwd(x, wt, rb, bias=b, residual=res, relu=True, compact=False, split_tail=False)
    assert torch.equal(got, base)
    assert float(got.float().abs().max()) > 0


@pytest.mark.parametrize("n_sites", [66000, 70000, 90000])
def test_compaction_kernel_half_height_tail_bitwise(n_sites, dev):
    """spconv_cb16's tail launch (the last partial round of 256-row tiles re-tiled at 128 rows in a second launch) is
    bitwise equal to the single launch: 66000 / 70000 / 90000 rows = one full round of 256 tiles + 2 / 18 / 96 tail
    tiles (the last one ragged). Forward with the full epilogue, and dgrad."""
    from vista_amd import ops
    rng = np.random.default_rng(n_sites)
    shape = [5, 200, 200]
    coords = random_sites(rng, 1, shape, n_sites)
    n = coords.shape[0]
    x = torch.from_numpy(rng.normal(size=(n, 64)).astype(np.float32)).to(dev)
    wt = torch.from_numpy((rng.normal(size=(64, 3, 3, 3, 64)) / np.sqrt(27 * 64)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.normal(size=(64,)).astype(np.float32)).to(dev)
    res = torch.from_numpy(rng.normal(size=(n, 64)).astype(np.float32)).to(dev)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 1, shape, 3)
    got = ops.conv_fwd(x, wt, rb, bias=b, residual=res, relu=True)
    base = ops.conv_fwd(x, wt, rb, bias=b, residual=res, relu=True, split_tail=False)
    assert torch.equal(got, base)
    assert float(got.abs().max()) > 0
    # against the row-stationary kernel family: same values to fp32 rounding
    other = ops.conv_fwd(x, wt, rb, bias=b, residual=res, relu=True, compact=False)
    assert float((got - other).abs().max()) <= 1e-4 * float(other.abs().max())


def test_voxelize_batch_equals_per_frame(dev):
    """One batched launch sequence == per-frame calls + collate (bit-exact, incl. the mean)."""
    from vista_amd import ops
    from vista_amd.synth import GEOMETRIES, lidar_cloud
    rg, vs, T, mv = GEOMETRIES["SMALL"]
    clouds = [torch.from_numpy(lidar_cloud(seed=40 + b, sweeps=1 + b)).to(dev) for b in range(3)]
    clouds.insert(1, clouds[0][:0])                      # an empty frame in the middle
    per = [ops.voxelize(c, vs, rg, T, mv, batch_idx=i, with_voxels=True) for i, c in enumerate(clouds)]
    bat = ops.voxelize_batch(clouds, vs, rg, T, mv, with_voxels=True)
    assert bat["frame_voxels"] == [p["num_voxels"] for p in per] and bat["frame_voxels"][1] == 0
    for k in ("coords", "num_points", "voxels", "mean"):
        assert torch.equal(bat[k], torch.cat([p[k] for p in per])), k
    # a frame with more points than max_voxels takes the per-frame path (truncation semantics kept)
    small = ops.voxelize_batch(clouds, vs, rg, T, 100, with_voxels=False)
    assert small["frame_voxels"] == [100, 0, 100, 100]


def test_frame_pipeline_bitwise_equals_single_batch(dev):
    """Sub-batches on alternating streams (engine.FramePipeline) == one batch, bit for bit."""
    from vista_amd import models
    from vista_amd.engine import FramePipeline, FusedBackbone
    from vista_amd.synth import GEOMETRIES, backbone_state_dict, grid_size, lidar_cloud
    from vista_amd.voxel import VoxelGenerator, voxelize_batch
    rg, vs, T, mv = GEOMETRIES["SMALL"]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    net = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5),
                                       backbone_state_dict("SpMiddleResNetFHD")).to(dev).eval()
    eng = FusedBackbone(net)
    frames = [torch.from_numpy(lidar_cloud(seed=s, sweeps=1)).to(dev) for s in range(5)]
    mean, coords, _ = voxelize_batch(gen, frames)
    ref = eng(mean, coords, len(frames), grid)
    pipe = FramePipeline(gen, eng, groups=2)
    for _ in range(3):                       # call 1 learns the shape, calls 2-3 run on two streams
        out = pipe(frames, grid)
        torch.cuda.synchronize()
        assert out.shape == ref.shape and torch.equal(out, ref)
    a = pipe(frames, grid, join=False)       # un-joined calls pipeline; synchronise before use
    b = pipe(frames[:3], grid, join=False)
    torch.cuda.synchronize()
    assert torch.equal(a, ref) and torch.equal(b, ref[:3])


@pytest.mark.parametrize("n_sites,ksize,cin,cout", [(7, 3, 16, 16), (33, 1, 32, 32), (129, 3, 64, 64), (300, 5, 16, 32),
                                                     (257, 3, 64, 128), (1, 3, 32, 64),
                                                     # the 16-pair kernel (fp32 Cin = 64, Cout > 32): one offset only, 125
                                                     # offsets, a single row, ragged multi-tile grids with Cout tiled over y
                                                     (5, 1, 64, 64), (40, 5, 64, 64), (1, 3, 64, 128), (700, 3, 64, 64),
                                                     (513, 1, 64, 128)])
def test_compaction_kernel_edge_shapes(n_sites, ksize, cin, cout, dev):
    """Per-offset compaction kernels (spconv_cb and, for fp32 Cin = 64 with Cout > 32, spconv_cb16) on tiny / ragged tiles,
    1x1x1 and 5x5x5 kernels (125 offsets), Cout tiled over blockIdx.y, forward and dgrad, against the oracle and bitwise
    against itself on a second run."""
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(n_sites * 31 + ksize)
    shape = [6, 12, 12]
    coords = random_sites(rng, 1, shape, n_sites)
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    w = (rng.normal(size=(cout, ksize, ksize, ksize, cin)) / np.sqrt(ksize ** 3 * cin)).astype(np.float32)
    nbr = so.rulebook_subm(coords, shape, ksize)
    ref = so.conv_fwd(feat, w, nbr, None)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 1, shape, ksize)
    x, wt = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
    got = ops.conv_fwd(x, wt, rb, compact=True)
    again = ops.conv_fwd(x, wt, rb, compact=True)
    assert torch.equal(got, again)
    assert np.abs(got.cpu().numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6)
    dout = rng.normal(size=(coords.shape[0], cout)).astype(np.float32)
    din_ref = so.conv_dgrad(dout, w, nbr, coords.shape[0])
    din = ops.conv_dgrad(torch.from_numpy(dout).to(dev), wt, rb).cpu().numpy()
    assert np.abs(din - din_ref).max() <= 1e-4 * max(np.abs(din_ref).max(), 1e-6)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (64, 64), (64, 128), (128, 128)])
def test_conv_on_fully_dense_grid(cin, cout, dev):
    """Every site active: tiles whose pair count per offset reaches the tile size (all waves busy, no padding
    slots) and, at the grid faces, drops to partial groups. Forward of both kernel families vs the oracle."""
    from oracle import spconv_oracle as so
    from vista_amd import ops
    rng = np.random.default_rng(cin + cout)
    shape = [4, 16, 16]
    zz, yy, xx = np.meshgrid(np.arange(4), np.arange(16), np.arange(16), indexing="ij")
    one = np.stack([np.zeros(zz.size, dtype=np.int64), zz.ravel(), yy.ravel(), xx.ravel()], 1)
    coords = np.concatenate([one, one + [1, 0, 0, 0]]).astype(np.int32)          # 2 frames x 1024 sites, sorted
    feat = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    nbr = so.rulebook_subm(coords, shape, 3)
    ref = so.conv_fwd(feat, w, nbr, None)
    rb = ops.rulebook_subm(torch.from_numpy(coords).to(dev), 2, shape, 3)
    np.testing.assert_array_equal(rb.nbr.cpu().numpy(), nbr)
    x, wt = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
    for compact in (True, False):
        got = ops.conv_fwd(x, wt, rb, compact=compact).cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), compact
