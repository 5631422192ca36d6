"""GPU parity at BASELINE.json's FULL sizes (FHD [41,1440,1440], ~247k points per frame; STRESS 0.05 m,
~1M points): one frame end to end against the CPU oracle (it finishes in seconds on the box's cores), and
size-independent properties where the oracle would be too slow: coordinate consistency of every rulebook
entry, sortedness / uniqueness of output sites, subm symmetry, pair-count conservation, batch invariance
(bitwise), linearity of a conv layer. All calls go through the C-ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _setup(geometry, seeds, sweeps, dev):
    from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES[geometry]
    gen = VoxelGenerator(vs, rg, T, mv)
    clouds = [lidar_cloud(seed=s, sweeps=sweeps) for s in seeds]
    grid = grid_size(rg, vs).tolist()
    return gen, clouds, grid, (rg, vs, T, mv)


def _lin(c, shape):
    c = c.long()
    return ((c[:, 0] * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]


def _check_rulebook(rb, in_idx, subm):
    """Every entry nbr[k][o] = i must satisfy in_coord(i) == out_coord(o) * stride - pad + k (same frame);
    every (i, k) whose target output exists must be present (conservation: counted through the transpose)."""
    ks, st, pd = rb.ksize, rb.stride, rb.padding
    out_idx = rb.out_indices
    lin = _lin(out_idx, rb.out_shape)
    if not subm:
        assert bool((lin[1:] > lin[:-1]).all()), "strided-conv output sites must be sorted and unique"
    total = 0
    k = 0
    for kz in range(ks[0]):
        for ky in range(ks[1]):
            for kx in range(ks[2]):
                row = rb.nbr[k]
                o = torch.nonzero(row >= 0, as_tuple=True)[0]
                i = row[o].long()
                total += int(o.numel())
                oc, ic = out_idx[o].long(), in_idx[i].long()
                assert torch.equal(ic[:, 0], oc[:, 0])
                for d, kk in ((1, kz), (2, ky), (3, kx)):
                    assert torch.equal(ic[:, d], oc[:, d] * st[d - 1] - pd[d - 1] + kk), (k, d)
                k += 1
    return total


def _expected_pairs(in_idx, rb):
    """Number of (input, offset) combinations whose output coordinate is on the output grid: every one of
    them must appear in the rulebook of a regular sparse conv (all such outputs are active)."""
    ks, st, pd = rb.ksize, rb.stride, rb.padding
    c = in_idx.long()
    cnt = torch.ones(c.shape[0], dtype=torch.long, device=c.device)
    for d in range(3):
        per = torch.zeros_like(cnt)
        for kk in range(ks[d]):
            t = c[:, d + 1] + pd[d] - kk
            ok = (t >= 0) & (t % st[d] == 0) & (t // st[d] < rb.out_shape[d])
            per += ok.long()
        cnt *= per
    return int(cnt.sum().item())


def test_fhd_frame_end_to_end_vs_oracle(dev):
    """BASELINE configs[0]/[1] at full size, one frame: voxels and every stage's site set bit-exact, dense
    output within 1e-4 relative (fp32), module path and fused engine."""
    from oracle import backbone_oracle as bo
    from oracle import spconv_oracle as so
    from oracle import voxelize_np as ov
    from vista_amd import models, ops
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import backbone_state_dict
    from vista_amd.voxel import voxelize_batch
    gen, clouds, grid, (rg, vs, T, mv) = _setup("FHD", [0], 10, dev)
    mean, coords, nump = voxelize_batch(gen, [torch.from_numpy(clouds[0]).to(dev)])
    v, c, n = ov.voxelize(clouds[0], vs, rg, T, mv)
    o_coords = ov.collate_coords([c])
    np.testing.assert_array_equal(coords.cpu().numpy(), o_coords)
    np.testing.assert_array_equal(nump.cpu().numpy(), n)
    o_feat = ov.mean_vfe(v, n)
    np.testing.assert_allclose(mean.cpu().numpy(), o_feat, rtol=1e-6, atol=1e-6)
    # site sets of the three strided stages + last conv, bit-exact (canonical order = sorted linear index)
    shape = [grid[2] + 1, grid[1], grid[0]]
    g_idx, g_shape, o_idx, o_shape = coords, shape, o_coords, shape
    for ks, st, pd in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
                       ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))):
        rb = ops.rulebook_conv(g_idx, 1, g_shape, ks, st, pd)
        oc, onbr, osh = so.rulebook_conv(o_idx, o_shape, ks, st, pd)
        np.testing.assert_array_equal(rb.out_indices.cpu().numpy(), oc)
        np.testing.assert_array_equal(rb.nbr.cpu().numpy(), onbr)
        g_idx, g_shape, o_idx, o_shape = rb.out_indices, rb.out_shape, oc, list(osh)
    sd = backbone_state_dict("SpMiddleResNetFHD")
    net = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5), sd).to(dev).eval()
    ref, _ = bo.sp_middle_resnet_fhd(o_feat, o_coords, 1, grid, sd)
    with torch.no_grad():
        out_mod = net(mean, coords, 1, grid).cpu().numpy()
    out_fused = FusedBackbone(net)(mean, coords, 1, grid).cpu().numpy()
    assert out_mod.shape == ref.shape == out_fused.shape == (1, 256, 180, 180)
    scale = np.abs(ref).max()
    assert np.abs(out_mod - ref).max() <= 1e-4 * scale      # north_star: 1e-4 relative in fp32
    assert np.abs(out_fused - ref).max() <= 1e-4 * scale
    out_bf16 = FusedBackbone(net, dtype=torch.bfloat16)(mean, coords, 1, grid).float().cpu().numpy()
    assert np.abs(out_bf16 - ref).max() <= 2e-2 * scale     # north_star: 2e-2 in bf16


@pytest.mark.parametrize("geometry,seeds,sweeps", [("FHD", [0, 1, 2, 3], 10), ("STRESS", [7], 40)])
def test_fullsize_rulebook_properties(geometry, seeds, sweeps, dev):
    from vista_amd import ops
    from vista_amd.voxel import voxelize_batch
    gen, clouds, grid, _ = _setup(geometry, seeds, sweeps, dev)
    batch = len(seeds)
    mean, coords, nump = voxelize_batch(gen, [torch.from_numpy(c).to(dev) for c in clouds])
    assert int(nump.min()) >= 1 and mean.shape[0] == coords.shape[0]
    shape = [grid[2] + 1, grid[1], grid[0]]
    # voxels are unique per frame
    lin = _lin(coords, shape)
    assert torch.unique(lin).numel() == lin.numel()
    idx, rank = coords, None
    for ks, st, pd in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
                       ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))):
        # submanifold rulebook of the current site set (hash path at stage 0, rank-table path after)
        rs = ops.rulebook_subm(idx, batch, shape, 3, rank=rank)
        n_pairs = _check_rulebook(rs, idx, subm=True)
        assert torch.equal(rs.nbr[13], torch.arange(idx.shape[0], dtype=torch.int32, device=dev))  # centre offset
        # symmetry: nbr[26-k][j] == i  <=>  nbr[k][i] == j
        for k in (0, 5, 12):
            o = torch.nonzero(rs.nbr[k] >= 0, as_tuple=True)[0]
            i = rs.nbr[k][o].long()
            assert torch.equal(rs.nbr[26 - k][i].long(), o)
        assert n_pairs == int((rs.nbr >= 0).sum())
        # strided rulebook to the next stage
        rb = ops.rulebook_conv(idx, batch, shape, ks, st, pd, transpose=True)
        got = _check_rulebook(rb, idx, subm=False)
        assert got == _expected_pairs(idx, rb), "a strided conv must hold every (input, offset) on the output grid"
        assert got == int((rb.nbr_t >= 0).sum())
        kk, ii = torch.nonzero(rb.nbr_t >= 0, as_tuple=True)
        assert torch.equal(rb.nbr[kk, rb.nbr_t[kk, ii].long()].long(), ii)      # transpose table is the inverse
        pairs, num = rb.pairs()
        assert int(num.sum()) == got
        idx, shape, rank = rb.out_indices, rb.out_shape, rb.out_rank


def test_fhd_batch_invariance_and_linearity(dev):
    """A frame's output does not depend on what else is in the batch (bitwise), and a conv layer is linear."""
    from vista_amd import models, ops
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import backbone_state_dict
    from vista_amd.voxel import voxelize_batch
    gen, clouds, grid, _ = _setup("FHD", [0, 1, 2], 10, dev)
    frames = [torch.from_numpy(c).to(dev) for c in clouds]
    net = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5),
                                       backbone_state_dict("SpMiddleResNetFHD")).to(dev).eval()
    eng = FusedBackbone(net)
    mean, coords, _ = voxelize_batch(gen, frames)
    full = eng(mean, coords, 3, grid)
    for b in (0, 2):
        m1, c1, _ = voxelize_batch(gen, [frames[b]])
        one = eng(m1, c1, 1, grid)
        assert torch.equal(one[0], full[b])
    # linearity of one full-size SubM layer (C=64 -> 64), no bias / activation
    shape = [grid[2] + 1, grid[1], grid[0]]
    rb1 = ops.rulebook_conv(coords, 3, shape, 3, 2, 1)
    rb2 = ops.rulebook_conv(rb1.out_indices, 3, rb1.out_shape, 3, 2, 1)
    rs = ops.rulebook_subm(rb2.out_indices, 3, rb2.out_shape, 3, rank=rb2.out_rank)
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(rs.n_in, 64, generator=g).to(dev)
    y = torch.randn(rs.n_in, 64, generator=g).to(dev)
    w = (torch.randn(64, 3, 3, 3, 64, generator=g) / np.sqrt(27 * 64)).to(dev)
    fx, fy = ops.conv_fwd(x, w, rs), ops.conv_fwd(y, w, rs)
    fxy = ops.conv_fwd(2.0 * x - 0.5 * y, w, rs)
    err = (fxy - (2.0 * fx - 0.5 * fy)).abs().max()
    assert float(err) <= 1e-5 * float(fxy.abs().max())
    # dgrad is the adjoint of fwd: <conv(x), g> == <x, dgrad(g)>
    gy = torch.randn(rs.n_out, 64, generator=g).to(dev)
    lhs = (fx.double() * gy.double()).sum()
    rhs = (x.double() * ops.conv_dgrad(gy, w, rs).double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-5 * abs(float(lhs))


def test_vista_full_net_forward(dev):
    """BASELINE configs[2] shape check at full size: one VISTA frame through MultiViewBackbone + dense tail; the
    dense tail on the GPU (MIOpen / rocBLAS) against the same modules on the CPU fed the same backbone maps."""
    from vista_amd import models
    from vista_amd.dense_tail import OHS_Multiview, VISTA_TASKS
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import backbone_state_dict
    from vista_amd.voxel import voxelize_batch
    gen, clouds, grid, _ = _setup("VISTA", [0], 10, dev)
    torch.manual_seed(5)
    det = OHS_Multiview().eval()
    models.load_numpy_state_dict(det.backbone, backbone_state_dict("MultiViewBackbone"))
    # non-trivial BatchNorm statistics in the dense tail
    with torch.no_grad():
        for m in det.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    cpu_state = {k: v.clone() for k, v in det.state_dict().items()}
    det = det.to(dev)
    mean, coords, nump = voxelize_batch(gen, [torch.from_numpy(clouds[0]).to(dev)])
    bev, rv = FusedBackbone(det.backbone)(mean, coords, 1, grid)
    assert bev.shape == (1, 640, 160, 160) and rv.shape == (1, 640, 160, 12)     # multiview.py:101-105, ohs.py:44-47
    with torch.no_grad():
        preds = det.dense_tail(bev, rv)
        # the module path (reader + backbone modules + tail) gives the same predictions as engine + tail
        ex = dict(voxels=mean[:, None, :], num_points=torch.ones_like(nump), coordinates=coords,
                  num_voxels=np.array([coords.shape[0]]), shape=[np.array(grid)])
        preds_mod = det(ex)
    ref = OHS_Multiview().eval()
    ref.load_state_dict(cpu_state)
    with torch.no_grad():
        preds_cpu = ref.dense_tail(bev.cpu(), rv.cpu())
    assert len(preds) == len(VISTA_TASKS)
    for p, q, c, t in zip(preds, preds_mod, preds_cpu, VISTA_TASKS):
        assert p["cls_preds"].shape == (1, 160, 160, t["num_class"]) and p["box_preds"].shape == (1, 160, 160, 10)
        for k in ("cls_preds", "box_preds"):
            assert bool(torch.isfinite(p[k]).all())
            s = float(c[k].abs().max())
            assert float((p[k].cpu() - c[k]).abs().max()) <= 2e-3 * s       # fp32 conv algorithms differ (MIOpen vs CPU)
            assert float((q[k] - p[k]).abs().max()) <= 2e-3 * s


def test_vista_frame_multiview_backbone_vs_oracle(dev):
    """BASELINE configs[2], sparse part, at FULL size, per-layer dispatch (the captured plan runs the same case in
    tests/test_x_gpu_plan.py, collected after every other parity test)."""
    from _fullsize_cases import vista_multiview_case
    vista_multiview_case(dev, dispatch="eager")


def test_stress_frame_end_to_end_vs_oracle(dev):
    """BASELINE configs[3] at FULL size, per-layer dispatch (captured: tests/test_x_gpu_plan.py)."""
    from _fullsize_cases import stress_case
    stress_case(dev, dispatch="eager")


def test_fhd_train_step_vs_oracle(dev):
    """BASELINE configs[4] at FULL size, one frame: loss of the training forward (train-mode BatchNorm, fused
    kernels) and the weight gradients of the deep stages (middle_conv.15 and up: the conv 64->128, both 128-channel
    blocks, the last conv) against torch CPU autograd over the ORACLE's fp32 conv forward / dgrad / wgrad."""
    from oracle import backbone_oracle as bo  # noqa: F401  (oracle package import check)
    from oracle import spconv_oracle as so
    from vista_amd import models
    from vista_amd.fused_bn import fuse_for_training
    from vista_amd.synth import backbone_state_dict
    from vista_amd.voxel import voxelize_batch
    gen, clouds, grid, _ = _setup("FHD", [0], 10, dev)
    mean, coords, _ = voxelize_batch(gen, [torch.from_numpy(clouds[0]).to(dev)])
    sd = backbone_state_dict("SpMiddleResNetFHD")
    net = fuse_for_training(models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5), sd).to(dev).train())
    out = net(mean, coords, 1, grid)
    loss = (out ** 2).mean()
    loss.backward()
    numel = out.numel()

    class OConv(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, nbr):
            ctx.nbr, ctx.n_in = nbr, x.shape[0]
            ctx.save_for_backward(x, w)
            return torch.from_numpy(so.conv_fwd(x.numpy(), w.numpy(), nbr))

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            g = g.contiguous().numpy()
            din = torch.from_numpy(so.conv_dgrad(g, w.numpy(), ctx.nbr, ctx.n_in)) if ctx.needs_input_grad[0] else None
            dw = torch.from_numpy(so.conv_wgrad(x.numpy(), g, ctx.nbr, tuple(w.shape))) if ctx.needs_input_grad[1] else None
            return din, dw, None

    ref = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5), sd).train()
    deep = {n: p for n, p in ref.named_parameters() if int(n.split(".")[1]) >= 15}
    for n, p in ref.named_parameters():
        p.requires_grad_(n in deep)               # autograd then stops below middle_conv.15
    x, c = mean.cpu(), coords.cpu().numpy()
    shape = [grid[2] + 1, grid[1], grid[0]]
    cache = {}

    def run_conv(conv, x, c, shape):
        if conv.subm:
            key = (conv.indice_key, c.shape[0])
            if key not in cache:
                cache[key] = so.rulebook_subm(c, shape, conv.kernel_size)
            y, oc, osh = OConv.apply(x, conv.weight, cache[key]), c, shape
        else:
            oc, nbr, osh = so.rulebook_conv(c, shape, conv.kernel_size, conv.stride, conv.padding)
            y = OConv.apply(x, conv.weight, nbr)
        return (y + conv.bias if conv.bias is not None else y), oc, osh

    for m in ref.middle_conv:
        if isinstance(m, models.SparseBasicBlock):
            idt = x
            y, _, _ = run_conv(m.conv1, x, c, shape)
            y = torch.relu(m.bn1(y))
            y, _, _ = run_conv(m.conv2, y, c, shape)
            x = torch.relu(m.bn2(y) + idt)
        elif hasattr(m, "subm"):
            x, c, shape = run_conv(m, x, c, shape)
        else:
            x = m(x)
    loss_ref = (x ** 2).sum() / numel
    loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    got = dict(net.named_parameters())
    checked = 0
    for n, p in deep.items():
        if not (n.endswith("weight") and p.dim() == 5):
            continue
        g, r = got[n].grad.cpu().numpy().ravel().astype(np.float64), p.grad.numpy().ravel().astype(np.float64)
        assert np.abs(g - r).max() <= 1e-3 * np.abs(r).max(), (n, np.abs(g - r).max() / np.abs(r).max())
        assert float(g @ r / (np.linalg.norm(g) * np.linalg.norm(r))) >= 0.99999, n
        checked += 1
    assert checked == 6     # conv 64->128, 2 blocks x 2 convs at 128 channels, the last conv
