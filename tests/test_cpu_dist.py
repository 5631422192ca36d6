"""CPU, world_size 2 over gloo: the N>1 logic of the path (frame sharding + flat-bucket gradient
averaging). The sparse kernels themselves need a GPU; the collectives and bookkeeping do not."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vista_amd.train import allreduce_grads_flat, shard_frames
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    data = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10.0
    mine = shard_frames(8, rank, world)
    loss = (net(data[mine]) ** 2).sum() / 8.0 * world   # so that the rank-average equals the global mean loss
    loss.backward()
    nbytes = allreduce_grads_flat(list(net.parameters()), dist, world)
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    q.put((rank, mine, nbytes, flat))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_equals_single_process_gradient():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0][1] + res[1][1]) == list(range(8))
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    data = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10.0
    ((net(data) ** 2).sum() / 8.0).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert res[0][2] == ref.numel() * 4
    for _, _, _, flat in res:
        assert torch.allclose(flat, ref, rtol=1e-5, atol=1e-6)
    assert torch.equal(res[0][3], res[1][3])      # both ranks hold identical averaged gradients


def _bn_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vista_amd.fused_bn import _all_reduce_sum
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1000, 16, generator=g, dtype=torch.float64) * 2 + 0.5
    cut = 300                                   # uneven shards, as frames hold different voxel counts
    mine = x[:cut] if rank == 0 else x[cut:]
    # the vector vx_bn_stats produces on each rank: (sum x, sum x^2 per channel, row count)
    sums = torch.cat([mine.sum(0), (mine * mine).sum(0), torch.tensor([float(mine.shape[0])], dtype=torch.float64)])
    _all_reduce_sum(sums, None)
    q.put((rank, sums))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_sum_exchange_gives_global_statistics():
    """The one collective of FusedBatchNorm1d(sync=True): after _all_reduce_sum every rank holds the sums and the row
    count of ALL rows, from which vx_bn_finalize takes the batch mean / variance (SyncBatchNorm, train.py:301)."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1000, 16, generator=g, dtype=torch.float64) * 2 + 0.5
    assert torch.equal(res[0][1], res[1][1])
    s = res[0][1]
    n = float(s[32])
    assert n == 1000.0
    mean, var = s[:16] / n, s[16:32] / n - (s[:16] / n) ** 2
    assert torch.allclose(mean, x.mean(0), rtol=1e-12, atol=1e-12)
    assert torch.allclose(var, x.var(0, unbiased=False), rtol=1e-10, atol=1e-12)


class _SyncSum(torch.autograd.Function):
    """Stand-in for FusedBatchNorm1d(sync=True) on the CPU: identity forward, and in BACKWARD an all-reduce of a small
    vector on the default process group -- the collective SyncBatchNorm issues between a layer's gradient kernels."""

    @staticmethod
    def forward(ctx, x, log):
        ctx.log = log
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        t = g.sum(0)
        dist.all_reduce(t)
        ctx.log.append(t.clone())
        return g, None


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vista_amd.train import GradBuckets, allreduce_grads_flat

    def make():
        torch.manual_seed(0)
        # `unused` first: it comes LAST in backward order, so the buckets before its own complete on every rank
        return torch.nn.ModuleDict(dict(unused=torch.nn.Linear(4, 4), a=torch.nn.Linear(6, 8), b=torch.nn.Linear(8, 5),
                                        c=torch.nn.Linear(5, 3)))

    def loss_of(net, x, log):
        y = torch.relu(_SyncSum.apply(net["a"](x), log))
        y = net["c"](torch.relu(_SyncSum.apply(net["b"](y), log)))
        if rank == 1:                       # only rank 1 touches `unused`: on rank 0 its gradient is None
            y = y + net["unused"](x[:, :4]).sum() * 0.1
        return (y ** 2).sum()

    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5, 6, generator=g)
    flat_net, bucket_net = make(), make()
    sync_ref, sync_got = [], []
    loss_of(flat_net, x, sync_ref).backward()
    assert (flat_net["unused"].weight.grad is None) == (rank == 0)
    nbytes = allreduce_grads_flat(list(flat_net.parameters()), dist, world)
    gb = GradBuckets(bucket_net.named_parameters(), 3, dist, world)
    logs = []
    for _ in range(3):                      # the later steps start from zero() with the views still attached
        gb.zero()
        sync_got.clear()
        loss_of(bucket_net, x, sync_got).backward()
        assert gb.finish() == nbytes
        logs.append(list(gb.log))
    names = [n for n, _ in flat_net.named_parameters()]
    flat = torch.cat([p.grad.reshape(-1) for p in flat_net.parameters()])
    buck = torch.cat([dict(bucket_net.named_parameters())[n].grad.reshape(-1) for n in names])
    same_sync = all(torch.equal(u, v) for u, v in zip(sync_ref, sync_got)) and len(sync_got) == 2
    q.put((rank, flat.numpy().tobytes(), buck.numpy().tobytes(), logs, len(gb.bounds), gb.hook_limit, same_sync))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_exchange_equals_flat_exchange_bitwise_with_a_missing_gradient():
    """GradBuckets (gradients as views of one flat buffer, buckets reduced as they complete) against the single flat
    all-reduce: identical bits on both ranks, although rank 0 has no gradient for one parameter AND the backward pass
    issues SyncBatchNorm-style collectives on the same process group between the layers: every rank fires
    every bucket at the same point of its backward pass -- nothing from hooks on the first step, then exactly the
    buckets that every rank completes."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, flat, buck, logs, nb, limit, same_sync in res:
        assert flat == buck and len(flat) > 0 and same_sync
        assert nb == 3 and limit == 2                  # the last bucket holds `unused`: incomplete on rank 0
        for log in logs:
            assert [b for b, _ in log] == list(range(nb))
        assert [why for _, why in logs[0]] == ["finish"] * nb                       # first step: agreement only
        assert [why for _, why in logs[1]] == ["hook", "hook", "finish"] == [why for _, why in logs[2]]
    assert res[0][1] == res[1][1]
    assert res[0][3] == res[1][3]                       # the same firing points on both ranks, step by step
