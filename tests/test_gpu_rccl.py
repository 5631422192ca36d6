"""The RCCL ("nccl" backend) branches of the data-parallel training step, executed on the GPU box: a world-size-1
RCCL process group inside this process (RCCL refuses two ranks on one device, and the box has one). The sum over one
rank is the identity, so the distributed step must reproduce the single-process step bit for bit while every
collective -- the SyncBatchNorm sum exchanges of FusedBatchNorm1d(sync=True), the asynchronous bucket all-reduces of
GradBuckets, the flat all-reduce -- really runs through RCCL on device memory (det3d/torchie/apis/train.py:299-308,
det3d/core/utils/dist_utils.py:7-41). The two-rank logic is covered on CPU with gloo (tests/test_cpu_dist.py)."""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_training_step_through_rccl_world1_equals_single_process_step():
    import torch.distributed as dist
    from test_gpu_a_parity import _tiny_frames
    from vista_amd import models
    from vista_amd.fused_bn import fuse_for_training
    from vista_amd.synth import backbone_state_dict
    from vista_amd.train import GradBuckets, allreduce_grads_flat, train_step
    if dist.is_initialized():
        pytest.skip("a default process group already exists in this process")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    clouds, mean, coords, grid, _ = _tiny_frames(dev, 2)
    sd = backbone_state_dict("SpMiddleResNetFHD")

    def run(use_dist, buckets):
        net = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5), sd).to(dev).train()
        fuse_for_training(net, sync=use_dist)
        opt = torch.optim.SGD(net.parameters(), lr=1e-3)
        gb = GradBuckets(net.named_parameters(), 4, dist if use_dist else None, 1) if buckets else None
        losses = [float(train_step(net, opt, mean, coords, 2, grid, dist if use_dist else None, 1, gb)) for _ in range(2)]
        return losses, [p.detach().clone() for p in net.parameters()], gb

    ref_losses, ref_params, gb0 = run(False, True)
    # every gradient of the step was written straight into its slice of the flat buffer: 21 conv weights, 16 block
    # biases, 21 x (gamma, beta) of the fused BatchNorms -- no accumulate adds (the un-bucketed runs below go through
    # autograd's accumulation and must give the same bits)
    assert gb0.direct == len(gb0.params) == 21 + 16 + 42
    with tempfile.TemporaryDirectory() as td:
        dist.init_process_group("nccl", init_method=f"file://{os.path.join(td, 'rdv')}", rank=0, world_size=1)
        try:
            assert dist.get_backend() == "nccl"
            for buckets in (True, False):
                losses, params, gb = run(True, buckets)
                assert losses == ref_losses
                for a, b in zip(params, ref_params):
                    assert torch.equal(a, b)
                if gb is not None:
                    assert len(gb.works) == len(gb.bounds) >= 2      # every bucket went out as an asynchronous RCCL op
            t = torch.arange(8, dtype=torch.float32, device=dev)
            p = torch.nn.Parameter(torch.zeros(8, device=dev))
            p.grad = t.clone()
            assert allreduce_grads_flat([p], dist, 1) == 32 and torch.equal(p.grad, t)
        finally:
            dist.destroy_process_group()


def test_wgrad_side_stream_is_bitwise_the_single_stream_step(monkeypatch):
    """GradBuckets puts the weight-gradient kernels on a second stream beside dgrad / BatchNorm backward (spconv's
    backward, vista_amd/train.py). Two full-size FHD frames, three optimizer steps: parameters and the flat gradient
    buffer must come out bit for bit as with VX_WGRAD_STREAM=0 -- the kernels are the same, only their streams
    differ, so any difference would be a missing dependency (a gradient read before its wgrad finished, an input
    freed and reused under the side stream)."""
    from _fullsize_cases import _setup
    from vista_amd import models
    from vista_amd.fused_bn import fuse_for_training
    from vista_amd.synth import backbone_state_dict
    from vista_amd.train import GradBuckets, train_step
    from vista_amd.voxel import voxelize_batch
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    gen, clouds, grid, _ = _setup("FHD", [0, 1], 10, dev)
    frames = [torch.from_numpy(c).to(dev) for c in clouds]
    sd = backbone_state_dict("SpMiddleResNetFHD")

    def run(side):
        monkeypatch.setenv("VX_WGRAD_STREAM", "1" if side else "0")
        net = fuse_for_training(models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5), sd).to(dev).train())
        opt = torch.optim.SGD(net.parameters(), lr=1e-3)
        gb = GradBuckets(net.named_parameters(), 4, None, 1)
        assert gb.side_enabled == side
        flats = []
        for _ in range(3):
            mean, coords, _ = voxelize_batch(gen, frames)
            train_step(net, opt, mean, coords, len(frames), grid, None, 1, gb)
            assert gb.side_used == side and gb.direct == len(gb.params)
            flats.append(gb.flat.clone())
        torch.cuda.synchronize()
        return flats, [p.detach().clone() for p in net.parameters()]

    f1, p1 = run(True)
    f0, p0 = run(False)
    for a, b in zip(f1, f0):
        assert torch.equal(a, b) and float(a.abs().sum()) > 0
    for a, b in zip(p1, p0):
        assert torch.equal(a, b)
