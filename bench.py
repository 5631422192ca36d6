#!/usr/bin/env python3
"""bench.py -- backbone frames/sec of the VISTA sparse voxel path on MI355X.

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
hard voxelization + mean VFE (fused) + SpMiddleResNetFHD forward (rulebooks, 21 sparse convs with
fused BN/ReLU/residual epilogues, dense()). Workload = BASELINE.json configs[1]: FHD geometry
(0.075 m, sparse_shape [41,1440,1440]), 10-sweep synthetic nuScenes-shaped frames (~247k points).

Multi-GPU: frames are independent, so ranks shard frames with no data-path
collective ("weak" scaling: fixed frames per GPU). `--mode train` runs the DDP backbone training
step instead (fwd + bwd + RCCL gradient all-reduce + Adam).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from vista_amd import models, ops  # noqa: E402
from vista_amd.engine import FramePipeline, FusedBackbone  # noqa: E402
from vista_amd.synth import GEOMETRIES, backbone_state_dict, grid_size, lidar_cloud  # noqa: E402
from vista_amd.voxel import VoxelGenerator, voxelize_batch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix = vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--geometry", default="FHD", choices=["FHD", "VISTA", "STRESS"])
    ap.add_argument("--mode", default="infer", choices=["infer", "train"])
    ap.add_argument("--backbone", default="SpMiddleResNetFHD", choices=["SpMiddleResNetFHD", "MultiViewBackbone"],
                    help="MultiViewBackbone needs --geometry VISTA (multiview.py:88-106)")
    ap.add_argument("--full-net", action="store_true",
                    help="BASELINE configs[2]: MultiViewBackbone + dense tail (necks, cross-view attention, OHS head; "
                         "stock PyTorch-ROCm ops) on the VISTA geometry")
    ap.add_argument("--postprocess", action="store_true",
                    help="with --full-net: add box decode + score filter + rotated NMS (head.predict) to the step")
    ap.add_argument("--streams", type=int, default=1,
                    help="sub-batches of a step on alternating HIP streams (vista_amd.engine.FramePipeline)")
    ap.add_argument("--torch-bn", action="store_true",
                    help="--mode train: stock torch BatchNorm1d/ReLU/add kernels instead of the fused libvx ones (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    return ap.parse_args()


def make_frames(geometry, batch, rank, dev):
    sweeps = 40 if geometry == "STRESS" else 10
    return [torch.from_numpy(lidar_cloud(seed=rank * batch + b, sweeps=sweeps)).to(dev) for b in range(batch)]


def cpu_baseline(geometry):
    """The oracle (a CPU port: numpy voxelizer + C/OpenMP sparse backbone) on ONE frame of the same workload."""
    from oracle import backbone_oracle as bo
    from oracle import spconv_oracle as so
    from oracle import voxelize_np as ov
    rg, vs, T, mv = GEOMETRIES[geometry]
    pts = lidar_cloud(seed=0, sweeps=40 if geometry == "STRESS" else 10)
    sd = backbone_state_dict("SpMiddleResNetFHD")
    threads = so.num_threads()
    t0 = time.perf_counter()
    v, c, n = ov.voxelize(pts, vs, rg, T, mv)
    feat = ov.mean_vfe(v, n)
    t1 = time.perf_counter()
    out, _ = bo.sp_middle_resnet_fhd(feat, ov.collate_coords([c]), 1, grid_size(rg, vs).tolist(), sd)
    t2 = time.perf_counter()
    return {"value": 1.0 / (t2 - t0), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 {geometry} frame (seed 0, {pts.shape[0]} points, {c.shape[0]} voxels): "
                      f"voxelize+VFE {1e3 * (t1 - t0):.0f} ms (numpy, 1 thread), "
                      f"backbone {1e3 * (t2 - t1):.0f} ms (C/OpenMP, {threads} threads)"}


def timed_infer(gen, engine, frames, grid, steps, warmup, barrier, streams=1, tail=None):
    pipe = FramePipeline(gen, engine, groups=streams) if streams > 1 else None

    def step():
        if pipe is not None:
            return pipe(frames, grid, join=False)
        mean, coords, _ = voxelize_batch(gen, frames)
        out = engine(mean, coords, len(frames), grid)
        if tail is not None:
            with torch.no_grad():
                out = tail(*out)
        return out
    for _ in range(warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    ops.TIMING = []
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    timing, ops.TIMING = ops.TIMING, None
    return dt, timing, out


def dominant_kernel_roofline(timing, dtype):
    """HIP-event time of every launch of the dominant conv kernel (Cin=Cout=128) in the timed region
    vs its algorithmic FLOPs 2*pairs*Cin*Cout."""
    sel = [t for t in timing if t["op"] == "conv" and t["cin"] == 128 and t["cout"] == 128 and t["kvol"] == 27]
    if not sel:
        return None
    ms = np.array([t["ev0"].elapsed_time(t["ev1"]) for t in sel])
    pairs = {}
    flops = []
    for t in sel:
        rb = t["rb"]
        if id(rb) not in pairs:
            pairs[id(rb)] = int((rb.nbr >= 0).sum().item())
        flops.append(2.0 * pairs[id(rb)] * 128 * 128)
    flops = np.array(flops)
    tflops = float(flops.sum() / (ms.sum() * 1e-3) / 1e12)
    # rocprofv3 --stats averages over EVERY launch of the kernel symbol: the k=27 SubM launches priced here plus the
    # (3,1,1) last conv, which runs on the same <128,128> instantiation; `symbol_avg_launch_ms` is that average
    sym = np.array([t["ev0"].elapsed_time(t["ev1"]) for t in timing
                    if t["op"] == "conv" and t["cin"] == 128 and t["cout"] == 128])
    common = {"launches": int(len(sel)), "avg_launch_ms": float(ms.mean()), "flops_per_launch": float(flops.mean()),
              "achieved_tflops": tflops, "symbol_launches": int(len(sym)), "symbol_avg_launch_ms": float(sym.mean())}
    if dtype == "fp32":
        # fp32: intensity Cout/2 = 64 flop per gathered byte -> MFMA-bound (157.3 TFLOP/s fp32 matrix peak)
        return {"bound": "mfma", "achieved": tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic("f32"),
                "kernel": "spconv_w8_split_tail<float,128,128> (SubMConv3d 128->128, k=3; spconv_w8 body)", **common}
    # bf16: MFMA is 16x faster, the gather phase bounds the layer (bytes = pairs*Cin*s + N_out*Cout*s
    # + K*Cin*Cout*s + pairs*8, s = 2)
    nbytes = np.array([p_ * 128 * 2 + t["rb"].n_out * 128 * 2 + 27 * 128 * 128 * 2 + p_ * 8
                       for t, p_ in zip(sel, [pairs[id(t["rb"])] for t in sel])], dtype=np.float64)
    gbs = float(nbytes.sum() / (ms.sum() * 1e-3) / 1e9)
    return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
            "traffic": pmc_traffic("bf16"), "kernel": "spconv_w8_split_tail<bf16,128,128> (SubMConv3d 128->128, k=3; spconv_w8 body)",
            "bytes_per_launch": float(nbytes.mean()), **common}


def phase_report(timing, steps, batch):
    """Per-phase time and achieved algorithmic rate (formulas of BASELINE.md section 2), from HIP events
    around every C-ABI call in the timed region. ms are per frame."""
    cache = {}

    def npairs(nbr):
        if id(nbr) not in cache:
            cache[id(nbr)] = int((nbr >= 0).sum().item())
        return cache[id(nbr)]

    acc = {}
    for t in timing:
        ms = t["ev0"].elapsed_time(t["ev1"])
        op = t["op"]
        nbytes = flops = 0.0
        if op == "voxelize":
            nbytes = t["n_points"] * t["n_feat"] * 4 + t["n_points"] * 8 + t.get("n_voxels", 0) * (16 + 4 + t["n_feat"] * 4)
        elif op == "rulebook_subm":
            nbytes = t["n"] * 16 + t["kvol"] * t["n"] * 8 + npairs(t["nbr"]) * 8
        elif op == "rulebook_conv":
            nbytes = t["n"] * 16 + t["kvol"] * t["n"] * 8 + t.get("n_out", 0) * 16 + npairs(t["nbr"]) * 8
        elif op == "conv":
            p_ = npairs(t["rb"].nbr)
            flops = 2.0 * p_ * t["cin"] * t["cout"]
            s_ = t["esize"]
            nbytes = p_ * t["cin"] * s_ + t["rb"].n_out * t["cout"] * s_ + t["kvol"] * t["cin"] * t["cout"] * s_ + p_ * 8
            op = "conv_C>=64" if min(t["cin"], t["cout"]) >= 64 else "conv_C<64"
        elif op == "dense":
            nbytes = t["numel"] * t["esize"] + 2 * t["n"] * t["c"] * t["esize"]
        a_ = acc.setdefault(op, [0.0, 0.0, 0.0, 0])
        a_[0] += ms
        a_[1] += nbytes
        a_[2] += flops
        a_[3] += 1
    out = {}
    for op, (ms, nbytes, flops, calls) in acc.items():
        out[op] = {"ms_per_frame": ms / (steps * batch), "calls_per_step": calls / steps,
                   "algorithmic_GB_per_s": nbytes / (ms * 1e-3) / 1e9 if ms else None,
                   "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS if ms else None}
        if flops:
            out[op]["TFLOP_per_s"] = flops / (ms * 1e-3) / 1e12
    return out


def pmc_traffic(dtype):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_summary.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); None if absent."""
    try:
        d = json.load(open(os.path.join(REPO, "profiles", "r01_pmc_summary.json")))
        return d["dominant_kernel"][dtype]["hbm_bytes_per_launch"]
    except Exception:
        return None


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        # "nccl" is RCCL on ROCm; VX_DIST_BACKEND=gloo only for rehearsing N>1 on a one-GPU box
        dist.init_process_group(os.environ.get("VX_DIST_BACKEND", "nccl"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    def barrier():
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()

    if a.mode == "train":
        from vista_amd.train import bench_train
        res = bench_train(a, dev, rank, world, dist)
        if rank == 0:
            print(json.dumps(res))
        if dist is not None:
            dist.destroy_process_group()
        return

    rg, vs, T, mv = GEOMETRIES[a.geometry]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    dtype = torch.float32 if a.dtype == "fp32" else torch.bfloat16
    net = models.load_numpy_state_dict(getattr(models, a.backbone)(num_input_features=5),
                                       backbone_state_dict(a.backbone)).to(dev).eval()
    engine = FusedBackbone(net, dtype=dtype)
    frames = make_frames(a.geometry, a.batch, rank, dev)
    tail = None
    if a.full_net:
        if a.geometry != "VISTA" or a.backbone != "MultiViewBackbone" or a.streams != 1:
            raise SystemExit("--full-net needs --geometry VISTA --backbone MultiViewBackbone (ohs.py:44-47 fixes the map sizes)")
        from vista_amd.dense_tail import OHS_Multiview, fuse_for_inference
        torch.manual_seed(1234)     # dense-tail weights: torch default init, BN in eval mode with default statistics
        det = OHS_Multiview().eval()
        det.backbone = None         # the sparse part runs through the fused engine above
        det = fuse_for_inference(det).to(dev).to(dtype)
        tail = det.dense_tail
        if a.postprocess:
            from vista_amd.postprocess import VISTA_TEST_CFG

            def tail(bev, rv):
                return det.bbox_head.predict({}, det.dense_tail(bev, rv), VISTA_TEST_CFG)

    dt, timing, out = timed_infer(gen, engine, frames, grid, a.steps, a.warmup, barrier, a.streams, tail)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_frames = world * a.batch * a.steps
    sweeps = 40 if a.geometry == "STRESS" else 10
    cfg_name = "configs[2]" if a.full_net else "configs[3]" if a.geometry == "STRESS" else "configs[1]"
    res = {
        "metric": (f"backbone frames/sec (voxelize + mean VFE + {a.backbone} forward)" if not a.full_net else
                   "full-net frames/sec (voxelize + mean VFE + MultiViewBackbone + necks + cross-view attention + "
                   "OHS head forward" + ("; no loss / box decode / NMS)" if not a.postprocess else
                                         " + box decode + score filter + rotated NMS)")),
        "value": total_frames / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32" if a.dtype == "fp32" else "bf16",
        "data": "synthetic",
        "config": {"workload": f"{a.geometry} {sweeps}-sweep synthetic nuScenes-shaped frames, sparse_shape "
                               f"{[grid[2] + 1, grid[1], grid[0]]}, batch {a.batch}/GPU, inference, "
                               f"inputs resident in HBM (BASELINE.json {cfg_name})",
                   "frames_per_gpu": a.batch, "points_per_frame": int(np.mean([f.shape[0] for f in frames])),
                   "parallelism": f"frame-sharded x{world}"},
    }
    if rank == 0:
        res["roofline"] = dominant_kernel_roofline(timing, a.dtype)
        res["phases"] = phase_report(timing, a.steps, a.batch)
        if world == 1 and not a.no_variants:
            # batch-1 latency variant of the same workload (not part of `value`)
            dt1, _, _ = timed_infer(gen, engine, frames[:1], grid, max(3, a.steps), 2, barrier, 1, tail)
            res["variants"] = {"batch1_frames_per_s": max(3, a.steps) / dt1}
            if a.streams == 1 and a.backbone == "SpMiddleResNetFHD" and a.batch >= 2:
                # same step as two half-batches on alternating HIP streams (engine.FramePipeline): higher
                # throughput, but kernels of the two streams overlap, so per-kernel HIP-event times (and the
                # roofline above) are only meaningful in the single-stream run that `value` comes from
                dt2, _, _ = timed_infer(gen, engine, frames, grid, a.steps, 2, barrier, streams=2)
                res["variants"]["two_stream_frames_per_s"] = a.batch * a.steps / dt2
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.geometry)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
