"""Data-parallel pieces of the path: frame sharding for inference and the backbone training step.

  shard_frames           det3d/datasets/loader/sampler.py:74-96 (DistributedSampler, shuffle=False: rank::world)
  allreduce_grads_flat   det3d/core/utils/dist_utils.py:7-41 (_allreduce_coalesced: one flat bucket per dtype)
  wrap_sync_bn           det3d/torchie/apis/train.py:301 (SyncBatchNorm.convert_sync_batchnorm)
  fused_bn.fuse_for_training  the same statistics exchange inside the fused BatchNorm+add+ReLU kernels (default)
  train_step             det3d/core/utils/dist_utils.py:50-56 + trainer loop (zero_grad, backward, step)

One process per GPU; the only collective on the data path is the gradient all-reduce (RCCL over xGMI
when the backend is "nccl"; gloo in the CPU tests). The backbone holds ~2.7 M fp32 parameters
(10.8 MB): on point-to-point xGMI a ring all-reduce of that size is latency-bound, so the whole
gradient goes out as ONE flat bucket right after backward instead of DDP's per-bucket hooks.
"""
import time

import numpy as np
import torch


def shard_frames(n_frames, rank, world):
    """Frame indices of this rank (frame i -> rank i mod world), padded like DistributedSampler so
    every rank gets ceil(n/world) frames."""
    per = (n_frames + world - 1) // world
    idx = list(range(n_frames))
    idx += idx[: per * world - n_frames]
    return idx[rank::world][:per]


def allreduce_grads_flat(params, dist=None, world=1):
    """Average gradients over ranks through one flat fp32 bucket. Returns the bucket size in bytes."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    if dist is not None and world > 1:
        if flat.is_cuda and dist.get_backend() != "nccl":
            # CPU rehearsal backends (gloo) stage through host memory; RCCL reduces in place on the GPU
            host = flat.cpu()
            dist.all_reduce(host)
            flat.copy_(host)
        else:
            dist.all_reduce(flat)
        flat.div_(world)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
    return flat.numel() * flat.element_size()


def wrap_sync_bn(module, world, dist=None):
    """SyncBatchNorm like the reference (train.py:301). Its collectives need the nccl (RCCL) backend
    for GPU tensors; under any other backend (CPU rehearsals) the per-rank BatchNorm is kept."""
    if world > 1 and dist is not None and dist.get_backend() == "nccl":
        return torch.nn.SyncBatchNorm.convert_sync_batchnorm(module)
    return module


def train_step(net, opt, mean, coords, batch, grid, dist=None, world=1):
    """fwd + loss = mean(dense_out^2) + bwd (libvx dgrad/wgrad) + gradient all-reduce + optimizer step."""
    opt.zero_grad(set_to_none=True)
    out = net(mean, coords, batch, grid)
    loss = (out * out).mean()
    loss.backward()
    allreduce_grads_flat(list(net.parameters()), dist, world)
    opt.step()
    return loss.detach()


def bench_train(a, dev, rank, world, dist):
    """BASELINE.json configs[4]: backbone training step under DP, batch a.batch frames per GPU."""
    from . import models
    from .synth import GEOMETRIES, backbone_state_dict, grid_size, lidar_cloud
    from .voxel import VoxelGenerator, voxelize_batch
    rg, vs, T, mv = GEOMETRIES[a.geometry]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    net = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5),
                                       backbone_state_dict("SpMiddleResNetFHD")).to(dev).train()
    if getattr(a, "torch_bn", False):
        net = wrap_sync_bn(net, world, dist)          # stock torch BatchNorm1d / SyncBatchNorm + ReLU + add kernels
    else:
        from .fused_bn import fuse_for_training      # libvx BatchNorm(+add+ReLU) kernels; global statistics when world > 1
        net = fuse_for_training(net, sync=world > 1)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    frames = [torch.from_numpy(lidar_cloud(seed=rank * a.batch + b, sweeps=10)).to(dev) for b in range(a.batch)]

    def step():
        mean, coords, _ = voxelize_batch(gen, frames)
        return train_step(net, opt, mean, coords, len(frames), grid, dist, world)

    def barrier():
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[dev.index])
            else:
                dist.barrier()

    for _ in range(a.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    nparam = sum(p.numel() for p in net.parameters())
    return {
        "metric": "backbone training frames/sec (voxelize + mean VFE + SpMiddleResNetFHD fwd+bwd + grad all-reduce + Adam)",
        "value": world * a.batch * a.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.geometry} 10-sweep synthetic frames, batch {a.batch}/GPU, DP training step "
                               "(BASELINE.json configs[4]), loss = mean(out^2), SyncBN when >1 GPU, "
                               + ("torch BatchNorm/ReLU kernels" if getattr(a, "torch_bn", False) else
                                  "fused libvx BatchNorm+add+ReLU kernels"),
                   "frames_per_gpu": a.batch, "parallelism": f"dp{world}", "grad_bucket_bytes": 4 * nparam},
        "final_loss": float(loss.item()),
    }
