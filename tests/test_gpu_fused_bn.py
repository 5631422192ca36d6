"""GPU tests of the fused train-mode BatchNorm1d (+ residual + ReLU) kernels (vx_bn.hip, vista_amd/fused_bn.py)
against stock torch BatchNorm1d / add / ReLU, which is what the reference applies to `.features`
(det3d/models/backbones/scn.py:82-98, :333-364). fp32 kernel -> torch reference of the same op; tolerances below."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _reference(x, gamma, beta, res, relu, dy, eps=1e-5, momentum=0.1, relu_mask=None):
    """float64 torch reference: BatchNorm1d(train) -> (+res) -> ReLU, gradients of sum(y * dy). relu_mask (0/1) stands
    in for the ReLU's own decision, so that elements within fp32 rounding of zero take the same side as the kernel."""
    c = x.shape[1]
    bn = torch.nn.BatchNorm1d(c, eps=eps, momentum=momentum).to(x.device).double().train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    xd = x.double().requires_grad_(True)
    rd = res.double().requires_grad_(True) if res is not None else None
    y = bn(xd)
    if rd is not None:
        y = y + rd
    pre = y.detach()
    if relu:
        y = torch.relu(y) if relu_mask is None else y * relu_mask.double()
    (y * dy.double()).sum().backward()
    return dict(pre=pre, y=y.detach(), dx=xd.grad, dgamma=bn.weight.grad, dbeta=bn.bias.grad,
                dres=rd.grad if rd is not None else None, rm=bn.running_mean, rv=bn.running_var)


def _close(a, b, tol, what, where=None):
    a, b = a.double(), b.double()
    scale = max(float(b.abs().max()), 1e-12)
    d = (a - b).abs()
    if where is not None:
        d = d * where
    err = float(d.max()) / scale
    assert err <= tol, (what, err)


@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("n,c", [(5, 16), (777, 16), (4099, 32), (100003, 64), (30001, 128), (513, 8), (1000, 256)])
def test_fused_bn_vs_torch(n, c, relu, with_res, dev):
    from vista_amd.fused_bn import FusedBatchNorm1d
    g = torch.Generator(device="cpu").manual_seed(n * 131 + c)
    x = (torch.randn(n, c, generator=g) * (0.5 + torch.rand(c, generator=g)) + torch.randn(c, generator=g) * 2).to(dev)
    if n < 16:
        # two or three random rows can land arbitrarily close: (x - mean) * invstd is then ill-conditioned in any fp32
        # implementation (the mean itself is stored in fp32); use rows that are spread out
        x = ((torch.arange(n).view(n, 1) - (n - 1) / 2) * (1 + torch.arange(c).view(1, c) / c) + 0.1).float().to(dev)
    res = torch.randn(n, c, generator=g).to(dev) if with_res else None
    dy = torch.randn(n, c, generator=g).to(dev)
    gamma = (0.5 + torch.rand(c, generator=g)).to(dev)
    beta = (torch.randn(c, generator=g) * 0.3).to(dev)

    bn = FusedBatchNorm1d(c, relu=relu).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if res is not None else None
    y = bn(xr, residual=rr)
    (y * dy).sum().backward()
    ref = _reference(x, gamma, beta, res, relu, dy, relu_mask=(y.detach() > 0) if relu else None)
    if relu:
        # the kernel's ReLU decisions are the reference's wherever the pre-activation is not within rounding of zero
        clear = ref["pre"].abs() > 1e-5
        assert float(clear.double().mean()) > 0.999
        assert torch.equal((y.detach() > 0)[clear], (ref["pre"] > 0)[clear])
    _close(y.detach(), ref["y"], 2e-6, "y")
    assert int(bn.num_batches_tracked) == 1
    _close(bn.running_mean, ref["rm"], 1e-6, "running_mean")
    _close(bn.running_var, ref["rv"], 1e-6, "running_var")
    # backward sums cancel (dx sums to ~0 per channel): tolerance relative to the largest entry
    _close(xr.grad, ref["dx"], 2e-5, "dx")
    _close(bn.weight.grad, ref["dgamma"], 1e-5, "dgamma")
    _close(bn.bias.grad, ref["dbeta"], 1e-5, "dbeta")
    if rr is not None:
        _close(rr.grad, ref["dres"], 1e-6, "dres")
    # ordered reductions: a second run is bitwise identical
    bn2 = FusedBatchNorm1d(c, relu=relu).to(dev).train()
    with torch.no_grad():
        bn2.weight.copy_(gamma)
        bn2.bias.copy_(beta)
    x2 = x.clone().requires_grad_(True)
    y2 = bn2(x2, residual=res)
    (y2 * dy).sum().backward()
    assert torch.equal(y2, y) and torch.equal(x2.grad, xr.grad) and torch.equal(bn2.weight.grad, bn.weight.grad)


def test_fused_bn_eval_and_unsupported_width_use_torch(dev):
    """Eval mode and widths outside the kernels' range keep nn.BatchNorm1d's own forward (same module API)."""
    from vista_amd.fused_bn import FusedBatchNorm1d
    torch.manual_seed(0)
    for c, train in ((16, False), (24, True)):
        ref = torch.nn.BatchNorm1d(c).to(dev).train(train)
        bn = FusedBatchNorm1d.from_bn(ref, relu=True)
        x = torch.randn(100, c, device=dev)
        ref2 = torch.nn.BatchNorm1d(c).to(dev).train(train)
        assert torch.allclose(bn(x), torch.relu(ref2(x)), atol=1e-6)


def test_fuse_for_training_matches_module_path(dev):
    """SpMiddleResNetFHD with fuse_for_training vs the plain module path (torch BatchNorm1d / ReLU / add) on the
    same frames: same state_dict keys, same loss, same running statistics, same gradients (to the conditioning
    of ~20 stacked train-mode BatchNorm backward passes, see test_backbone_train_step_grads_vs_oracle)."""
    from test_gpu_a_parity import _tiny_frames
    from vista_amd import models
    from vista_amd.fused_bn import FusedBatchNorm1d, fuse_for_training
    from vista_amd.synth import backbone_state_dict
    clouds, mean, coords, grid, _ = _tiny_frames(dev, 2)
    sd = backbone_state_dict("SpMiddleResNetFHD")
    nets = []
    for fused in (False, True):
        net = models.SpMiddleResNetFHD(num_input_features=5)
        models.load_numpy_state_dict(net, sd).to(dev).train()
        keys = list(net.state_dict().keys())
        if fused:
            fuse_for_training(net)
            assert list(net.state_dict().keys()) == keys
            assert sum(isinstance(m, FusedBatchNorm1d) for m in net.modules()) == 21
            assert not any(type(m) is torch.nn.BatchNorm1d for m in net.modules())
        out = net(mean, coords, 2, grid)
        loss = (out ** 2).mean()
        loss.backward()
        nets.append((net, out.detach(), loss.item()))
    (a, out_a, la), (b, out_b, lb) = nets
    assert abs(la - lb) <= 1e-5 * abs(la)
    assert float((out_a - out_b).abs().max()) <= 1e-4 * float(out_a.abs().max())
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if "running" in k:
            assert torch.allclose(sa[k], sb[k], rtol=1e-5, atol=1e-6), k
        if "num_batches_tracked" in k:
            assert int(sa[k]) == int(sb[k]) == 1
    # conv weight gradients of BOTH paths against the fp64 truth: 1e-4 wherever the ReLU gate pattern agrees with the
    # fp64 run (see test_gpu_a_parity._check_grads_against_fp64)
    from test_gpu_a_parity import (_check_grads_against_fp64, _check_grads_with_gpu_gates_imposed, _gpu_train_step,
                                 _oracle_train_step)
    numel = out_a.numel()
    gates64, gates32 = [], []
    _, g32 = _oracle_train_step(sd, mean, coords, grid, numel, torch.float32, gates32)
    _, g64 = _oracle_train_step(sd, mean, coords, grid, numel, torch.float64, gates64)
    for fused in (False, True):
        net_f, _, _, gates = _gpu_train_step(sd, mean, coords, grid, dev, fused=fused)
        assert _check_grads_against_fp64(net_f, gates, gates64, g32, g64, gates32) >= 1
        _check_grads_with_gpu_gates_imposed(net_f, gates, sd, mean, coords, grid, numel)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for n, p in pa.items():
        g, r = pb[n].grad.double().flatten(), p.grad.double().flatten()
        if float(r.abs().max()) < 1e-12:
            # conv biases in front of a train-mode BatchNorm: the exact gradient is zero, both sides hold rounding noise
            continue
        cos = float(g @ r / (g.norm() * r.norm()))
        rel = float((g - r).abs().max() / r.abs().max())
        stage = int(n.split(".")[1])
        if n.endswith("weight") and p.dim() == 5:
            # direction only: the two runs may open different rounding-level ReLU gates; each was held to 1e-4 against
            # the fp64 step with its own gates imposed just above
            assert cos >= 0.9995, (n, cos, rel, stage)
        elif p.dim() == 1 and "bn" in n or p.dim() == 1 and n.split(".")[-2].isdigit():
            assert cos >= 0.999, (n, cos, rel)


def test_conv_bias_gradient_in_front_of_a_training_norm_is_written_as_zero(dev):
    """fuse_for_training tells every biased conv whose output a BatchNorm consumes; in training mode its backward
    writes exact zeros for the bias gradient (the reduction over rows it skips yields rounding noise around zero:
    checked here against the unfused path, relative to the size of the terms that cancel) and leaves every other
    gradient bitwise as it was. In eval mode the norm is affine and the bias gradient is reduced as before."""
    from test_gpu_a_parity import _tiny_frames
    from vista_amd import models
    from vista_amd.fused_bn import fuse_for_training
    from vista_amd.synth import backbone_state_dict
    import spconv.pytorch as spconv
    clouds, mean, coords, grid, _ = _tiny_frames(dev, 2)
    sd = backbone_state_dict("SpMiddleResNetFHD")

    def run(skip, train=True):
        net = fuse_for_training(models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5), sd).to(dev))
        net.train(train)
        told = [m for m in net.modules() if isinstance(m, spconv.SparseConvolution) and m.norm_after is not None]
        assert len(told) == 16 and all(m.bias is not None for m in told)      # the two convs of each of 8 blocks
        if not skip:
            for m in told:
                object.__setattr__(m, "norm_after", None)
        out = net(mean, coords, 2, grid)
        (out ** 2).mean().backward()
        return net, {n: p.grad.clone() for n, p in net.named_parameters()}, {id(m.bias) for m in told}

    net, g_skip, told = run(True)
    _, g_full, _ = run(False)
    names = {n for n, p in net.named_parameters() if id(p) in told}
    assert len(names) == 16
    for n in g_full:
        if n in names:
            assert float(g_skip[n].abs().max()) == 0.0
            w = g_full[n.replace("bias", "weight")]
            assert float(g_full[n].abs().max()) <= 1e-4 * float(w.abs().max()) + 1e-12, n   # noise, not signal
        else:
            assert torch.equal(g_skip[n], g_full[n]), n
    net_e, g_eval, told_e = run(True, train=False)
    for n, p in net_e.named_parameters():
        if id(p) in told_e:
            assert float(g_eval[n].abs().max()) > 0.0, n


def test_from_dense_is_the_gather_of_to_dense(dev):
    from vista_amd import ops
    rng = np.random.default_rng(3)
    for c, dt in ((128, torch.float32), (64, torch.bfloat16), (5, torch.float32)):
        shape, batch = [2, 37, 41], 3
        cells = batch * shape[0] * shape[1] * shape[2]
        lin = np.sort(rng.choice(cells, size=1777, replace=False))
        idx = np.stack([lin // (shape[0] * shape[1] * shape[2]), (lin // (shape[1] * shape[2])) % shape[0],
                        (lin // shape[2]) % shape[1], lin % shape[2]], 1).astype(np.int32)
        idx = torch.from_numpy(idx).to(dev)
        dense = torch.randn((batch, c, *shape), device=dev).to(dt)
        got = ops.from_dense(dense, idx, shape)
        i = idx.long()
        ref = dense[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]]
        assert torch.equal(got, ref)
        assert torch.equal(ops.from_dense(ops.to_dense(got, idx, batch, shape), idx, shape), got)
    assert ops.from_dense(dense, idx[:0], shape).shape == (0, 5)


def test_fused_sync_bn_two_ranks_equal_one_rank_on_all_rows(dev):
    """sync=True on two "ranks" holding uneven shards of the rows == one unsynchronised module over all rows:
    statistics, outputs and input gradients agree, parameter gradients add up over ranks (SyncBatchNorm semantics,
    det3d/torchie/apis/train.py:301). Each rank runs in turn in this process (a GPU test process must not start
    further GPU processes on this pool); its all-reduce adds the other rank's sums, computed beforehand through the
    same C-ABI calls. The torch.distributed call itself is covered by test_fused_sync_bn_process_group."""
    from vista_amd import ops
    from vista_amd.fused_bn import FusedBatchNorm1d
    n, c = 5000, 32
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(n, c, generator=g) * 1.5 + 0.7).to(dev)
    res = torch.randn(n, c, generator=g).to(dev)
    dy = torch.randn(n, c, generator=g).to(dev)
    gamma = (0.5 + torch.rand(c, generator=g)).to(dev)
    beta = (torch.randn(c, generator=g) * 0.3).to(dev)
    cut = n // 3
    shards = [slice(0, cut), slice(cut, n)]
    # what each rank contributes at the two exchange points
    fwd = [ops.bn_stats(x[s].contiguous()) for s in shards]
    mean, invstd = ops.bn_finalize(fwd[0] + fwd[1], c, 1e-5, 0.1)
    bwd = []
    for s in shards:
        _, mask = ops.bn_apply(x[s].contiguous(), mean, invstd, gamma, beta, res[s].contiguous(), relu=True)
        bwd.append(ops.bn_bwd_reduce(dy[s].contiguous(), x[s].contiguous(), mask, mean, invstd)[0])
    parts = []
    for rank, s in enumerate(shards):
        other = [fwd[1 - rank], bwd[1 - rank]]
        bn = FusedBatchNorm1d(c, relu=True, sync=True, all_reduce=lambda t: t.add_(other.pop(0))).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(gamma)
            bn.bias.copy_(beta)
        xs = x[s].clone().requires_grad_(True)
        rs = res[s].clone().requires_grad_(True)
        y = bn(xs, residual=rs)
        (y * dy[s]).sum().backward()
        assert not other                      # both exchange points were reached, in order
        parts.append(dict(y=y.detach(), dx=xs.grad, dres=rs.grad, dgamma=bn.weight.grad, dbeta=bn.bias.grad,
                          rm=bn.running_mean, rv=bn.running_var))
    bn = FusedBatchNorm1d(c, relu=True).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    xs = x.clone().requires_grad_(True)
    rs = res.clone().requires_grad_(True)
    y = bn(xs, residual=rs)
    (y * dy).sum().backward()
    cat = lambda k: torch.cat([p[k] for p in parts])  # noqa: E731
    _close(cat("y"), y.detach(), 1e-6, "y")
    _close(cat("dx"), xs.grad, 1e-5, "dx")
    _close(cat("dres"), rs.grad, 1e-6, "dres")
    _close(parts[0]["dgamma"] + parts[1]["dgamma"], bn.weight.grad, 1e-5, "dgamma")
    _close(parts[0]["dbeta"] + parts[1]["dbeta"], bn.bias.grad, 1e-5, "dbeta")
    for p in parts:
        _close(p["rm"], bn.running_mean, 1e-6, "running_mean")
        _close(p["rv"], bn.running_var, 1e-6, "running_var")


def test_fused_sync_bn_process_group(dev):
    """sync=True through torch.distributed (a world-size-1 gloo group inside this process: the host-staged
    all-reduce of the rehearsal backends) gives the unsynchronised result bit for bit."""
    import torch.distributed as dist
    from vista_amd.fused_bn import FusedBatchNorm1d
    if dist.is_initialized():
        pytest.skip("a default process group already exists in this process")
    with tempfile.TemporaryDirectory() as td:
        dist.init_process_group("gloo", init_method=f"file://{os.path.join(td, 'rdv')}", rank=0, world_size=1)
        try:
            torch.manual_seed(3)
            x = torch.randn(3000, 64, device=dev)
            dy = torch.randn(3000, 64, device=dev)
            outs = []
            for sync in (False, True):
                bn = FusedBatchNorm1d(64, relu=True, sync=sync).to(dev).train()
                xs = x.clone().requires_grad_(True)
                y = bn(xs)
                (y * dy).sum().backward()
                outs.append((y.detach(), xs.grad, bn.weight.grad, bn.running_var))
            for a, b in zip(*outs):
                assert torch.equal(a, b)
        finally:
            dist.destroy_process_group()
