#!/usr/bin/env python3
"""bench.py -- backbone frames/sec of the VISTA sparse voxel path on MI355X.

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
hard voxelization + mean VFE (fused) + SpMiddleResNetFHD forward (rulebooks, 21 sparse convs with
fused BN/ReLU/residual epilogues, dense()). Workload = BASELINE.json configs[1]: FHD geometry
(0.075 m, sparse_shape [41,1440,1440]), 10-sweep synthetic nuScenes-shaped frames (~247k points).

Multi-GPU: frames are independent, so ranks shard frames with no data-path
collective ("weak" scaling: fixed frames per GPU). `--mode train` runs the DDP backbone training
step instead (fwd + bwd + RCCL gradient all-reduce + Adam).

`python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) is its own launcher: the parent
starts N fresh rank processes BEFORE anything touches HIP (it never initialises the GPU and never
execs), each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT, like
tools/scripts/train.sh:19 and tools/dist_test.py:85-129 of the reference do through
torch.distributed.launch. Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks
already exist and the environment is used as it is. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix = vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense matrix peak (the chip holds a lower clock under bf16 load)
PEAK_HBM_GBS = 8000.0
EV_STEPS = 20                  # steps of the separate instrumented pass (roofline / phases), whatever --steps says


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU). Default: WORLD_SIZE from the environment, else 1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 8; train 4)")
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
    ap.add_argument("--flat-allreduce", action="store_true",
                    help="--mode train: one flat gradient all-reduce after backward instead of overlapped buckets (A/B)")
    ap.add_argument("--eager", action="store_true",
                    help="inference: per-layer launches from the host (no captured plan), the round-1 path (A/B)")
    ap.add_argument("--dispatch", default=None, choices=["queued", "graph"],
                    help="how a sync-free step is issued: launches queued on the stream, or one captured HIP graph")
    ap.add_argument("--pipeline", action="store_true",
                    help="two plans in flight (vista_amd.engine.FramePipeline2): the voxelizer + rulebooks of step k + 1 "
                         "run beside the convolutions of step k; submit() returns the previous step's output")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    a = ap.parse_args(argv)
    if a.batch is None:
        a.batch = 4 if a.mode == "train" else 8
    return a


# ------------------------------------------------------------------------------------------------
# launcher: parent of N rank processes (never touches HIP)
# ------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, script=None, extra_env=None, timeout=3000):
    """Start n fresh rank processes of `script argv` and wait for them. Returns the worst exit code.
    The children inherit stdout/stderr (rank 0 prints the JSON line). If one rank fails the others are
    terminated (by PID: only processes this function started)."""
    script = script or os.path.abspath(__file__)
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "VX_BENCH_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, "-u", script] + list(argv), env=env))
    t_end = time.time() + timeout
    worst = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or rc
        if (worst or time.time() > t_end) and live:
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            worst = worst or 124
            break
        time.sleep(0.05)
    return worst


# ------------------------------------------------------------------------------------------------
def make_frames(geometry, batch, rank, dev):
    import torch
    from vista_amd.synth import lidar_cloud
    sweeps = 40 if geometry == "STRESS" else 10
    return [torch.from_numpy(lidar_cloud(seed=rank * batch + b, sweeps=sweeps)).to(dev) for b in range(batch)]


def cpu_baseline(geometry):
    """The oracle (a CPU port: numpy voxelizer + C/OpenMP sparse backbone) on ONE frame of the same workload."""
    from oracle import backbone_oracle as bo
    from oracle import spconv_oracle as so
    from oracle import voxelize_np as ov
    from vista_amd.synth import GEOMETRIES, backbone_state_dict, grid_size, lidar_cloud
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


def timed_infer(step, steps, warmup, barrier, record=False):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides.
    record=True additionally collects a HIP-event pair around every C-ABI call (a SEPARATE pass is used for
    that: `value` comes from a pass that records nothing)."""
    import torch
    from vista_amd import ops
    finish = getattr(step, "finish", lambda: None)   # a pipelined step: wait for (and check) what is still in flight
    for _ in range(warmup):
        step()
    finish()
    barrier()
    torch.cuda.synchronize()
    ops.TIMING = [] if record else None
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    finish()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    timing, ops.TIMING = ops.TIMING, None
    return dt, timing, out


def dominant_kernel_roofline(timing, dtype):
    """HIP-event time of every launch of the dominant conv kernel (Cin=Cout=128, k=27) vs its algorithmic
    FLOPs 2*pairs*Cin*Cout."""
    import numpy as np
    sel = [t for t in timing if t["op"] == "conv" and t["cin"] == 128 and t["cout"] == 128 and t["kvol"] == 27]
    if not sel:
        return None
    ms = np.array([t["ev0"].elapsed_time(t["ev1"]) for t in sel])
    pairs = {}
    flops = []
    for t in sel:
        rb = t["rb"]
        if id(rb) not in pairs:
            pairs[id(rb)] = rb.num_pairs()
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
        from vista_amd import _lib
        products = int(_lib.lib().vx_spconv_get_f32_products())
        if products:
            # fp32 operands split exactly into three bf16 planes, `products` plane products per fp32 product on the bf16
            # matrix pipe (fp32 accumulate): the bound is the bf16 MFMA peak divided by the MFMAs spent per product
            peak = PEAK_BF16_MFMA_TFLOPS / products
            return {"bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak,
                    "peak_basis": f"bf16 dense MFMA peak {PEAK_BF16_MFMA_TFLOPS:.0f} TFLOP/s / {products} plane products "
                                  "per fp32 product (algorithmic fp32 flops)",
                    "frac_of_fp32_mfma_peak": tflops / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic(f"f32_split{products}"),
                    "limiter": "board power: matrix pipe and L2->LDS fill overlap in cycles but the clock drops from 2.35 to "
                               "1.93 GHz when both run (DESIGN.md section 4, round-3 ablations)",
                    "kernel": f"spconv_w8s body, <128,128,{products}> (SubMConv3d 128->128, k=3; fp32 in/out/accumulate)",
                    **common}
        # fp32 MFMA: intensity Cout/2 = 64 flop per gathered byte -> MFMA-bound (157.3 TFLOP/s fp32 matrix peak)
        return {"bound": "mfma", "achieved": tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic("f32"),
                "kernel": "spconv_w8 body, <float,128,128> (SubMConv3d 128->128, k=3)", **common}
    # bf16: MFMA is 16x faster, the gather phase bounds the layer (bytes = pairs*Cin*s + N_out*Cout*s
    # + K*Cin*Cout*s + pairs*8, s = 2)
    nbytes = np.array([p_ * 128 * 2 + t["rb"].n_out * 128 * 2 + 27 * 128 * 128 * 2 + p_ * 8
                       for t, p_ in zip(sel, [pairs[id(t["rb"])] for t in sel])], dtype=np.float64)
    gbs = float(nbytes.sum() / (ms.sum() * 1e-3) / 1e9)
    traffic = pmc_traffic("bf16")
    out = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
           "traffic": traffic, "kernel": "spconv_w8 body, <bf16,128,128> (SubMConv3d 128->128, k=3)",
           "bytes_per_launch": float(nbytes.mean()), **common,
           # `achieved` prices the ALGORITHMIC gather bytes; most of them are L2 hits (rows are re-used across offsets
           # and tiles). What the counters say about the two real ceilings:
           "frac_of_bf16_mfma_peak": tflops / PEAK_BF16_MFMA_TFLOPS,
           "limiter": "L2->CU fill (on-chip): neither HBM nor the matrix pipe. The gathered rows, their zero rows and the "
                      "W chunk re-read by every 256-row tile arrive at ~25 B/clk/CU (DESIGN.md section 4)"}
    if traffic:
        real = traffic / (float(ms.mean()) * 1e-3) / 1e9
        out["hbm_counter_GB_per_s"] = real
        out["frac_of_hbm_peak_from_counters"] = real / PEAK_HBM_GBS
    return out


def phase_report(timing, steps, batch, geometry="FHD"):
    """Per-phase time and achieved algorithmic rate (formulas of BASELINE.md section 2), from HIP events
    around every C-ABI call of an instrumented pass. ms are per frame. `traffic` = HBM bytes per frame from the
    committed rocprofv3 PMC passes (profiles/r02_pmc_hash.json), where collected."""
    acc = {}
    for t in timing:
        ms = t["ev0"].elapsed_time(t["ev1"])
        op = t["op"]
        nbytes = flops = 0.0
        if op == "voxelize":
            nbytes = t["n_points"] * t["n_feat"] * 4 + t["n_points"] * 8 + t.get("n_voxels", 0) * (16 + 4 + t["n_feat"] * 4)
        elif op == "rulebook_subm":
            nbytes = t["n"] * 16 + t["kvol"] * t["n"] * 8 + t["rb"].num_pairs() * 8
        elif op == "rulebook_conv":
            nbytes = t["n"] * 16 + t["kvol"] * t["n"] * 8 + t["rb"].n_out * 16 + t["rb"].num_pairs() * 8
        elif op == "conv":
            p_ = t["rb"].num_pairs()
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
    pmc = pmc_phase_traffic(geometry, batch)
    out = {}
    for op, (ms, nbytes, flops, calls) in acc.items():
        out[op] = {"ms_per_frame": ms / (steps * batch), "calls_per_step": calls / steps,
                   "algorithmic_GB_per_s": nbytes / (ms * 1e-3) / 1e9 if ms else None,
                   "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS if ms else None,
                   "algorithmic_bytes_per_frame": nbytes / (steps * batch),
                   "traffic": pmc.get(op)}
        if flops:
            out[op]["TFLOP_per_s"] = flops / (ms * 1e-3) / 1e12
    return out


def pmc_traffic(dtype):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); None if absent."""
    for name in ("r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
        try:
            d = json.load(open(os.path.join(REPO, "profiles", name)))
            return d["dominant_kernel"][dtype]["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def pmc_phase_traffic(geometry, batch):
    """{phase: HBM bytes per frame} of the hash / rulebook / dense kernels from profiles/r02_pmc_hash.json."""
    try:
        for name in ("r03_pmc_hash.json", "r02_pmc_hash.json"):
            path = os.path.join(REPO, "profiles", name)
            if os.path.exists(path):
                return json.load(open(path))["per_frame_bytes"][f"{geometry}_b{batch}"]
    except Exception:
        pass
    return {}


def init_dist(a):
    """-> (dev, rank, world, dist | None, backend | None, barrier)."""
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus is not None and a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the ranks that exist do not match the request")
    dist = backend = None
    ndev = max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm. RCCL refuses two ranks on one device, so a rehearsal of N ranks on a box with
        # fewer GPUs (ranks share devices) falls back to gloo -- and says so in the JSON line.
        backend = os.environ.get("VX_DIST_BACKEND") or ("nccl" if ndev >= world else "gloo")
        local = local % ndev
        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != world:
            raise SystemExit("bench.py: process group size differs from WORLD_SIZE")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    def barrier():
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()

    return dev, rank, world, dist, backend, barrier


def max_over_ranks(dt, dist, backend, dev):
    import torch
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def build_infer(a, dev, rank, dtype_name=None, geometry=None, batch=None):
    """-> (step(frames) closure pieces) for one inference configuration."""
    import torch
    from vista_amd import models
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import GEOMETRIES, backbone_state_dict, grid_size
    from vista_amd.voxel import VoxelGenerator
    geometry = geometry or a.geometry
    dtype_name = dtype_name or a.dtype
    batch = batch or a.batch
    rg, vs, T, mv = GEOMETRIES[geometry]
    gen = VoxelGenerator(vs, rg, T, mv)
    grid = grid_size(rg, vs).tolist()
    dtype = torch.float32 if dtype_name == "fp32" else torch.bfloat16
    net = models.load_numpy_state_dict(getattr(models, a.backbone)(num_input_features=5),
                                       backbone_state_dict(a.backbone)).to(dev).eval()
    engine = FusedBackbone(net, dtype=dtype, eager=a.eager, dispatch=a.dispatch)
    frames = make_frames(geometry, batch, rank, dev)
    return gen, grid, engine, frames


def make_step(a, gen, grid, engine, frames, tail=None, pipeline=None):
    import torch
    from vista_amd.engine import FramePipeline, FramePipeline2
    pipe = FramePipeline(gen, engine, groups=a.streams) if a.streams > 1 else None
    pipeline = a.pipeline if pipeline is None else pipeline
    if pipeline and not engine.eager and tail is None and pipe is None:
        pipe2 = FramePipeline2(engine, gen, grid)

        def step2():
            return pipe2.submit(frames)
        step2.finish = pipe2.drain        # timed_infer collects the steps in flight before it stops the clock
        return step2

    def step():
        if pipe is not None:
            return pipe(frames, grid, join=False)
        out = engine.forward_frames(gen, frames, grid)   # captured plan, or per-layer launches when engine.eager
        if tail is not None:
            with torch.no_grad():
                out = tail(*out)
        return out
    return step


def run_infer(a, dev, rank, world, dist, backend, barrier):
    import numpy as np
    import torch
    gen, grid, engine, frames = build_infer(a, dev, rank)
    tail = None
    if a.full_net:
        if a.geometry != "VISTA" or a.backbone != "MultiViewBackbone" or a.streams != 1:
            raise SystemExit("--full-net needs --geometry VISTA --backbone MultiViewBackbone (ohs.py:44-47 fixes the map sizes)")
        from vista_amd.dense_tail import OHS_Multiview, fuse_for_inference
        torch.manual_seed(1234)     # dense-tail weights: torch default init, BN in eval mode with default statistics
        det = OHS_Multiview().eval()
        det.backbone = None         # the sparse part runs through the fused engine above
        dtype = torch.float32 if a.dtype == "fp32" else torch.bfloat16
        det = fuse_for_inference(det).to(dev).to(dtype)
        tail = det.dense_tail
        if a.postprocess:
            from vista_amd.postprocess import VISTA_TEST_CFG

            def tail(bev, rv):
                return det.bbox_head.predict({}, det.dense_tail(bev, rv), VISTA_TEST_CFG)

    step = make_step(a, gen, grid, engine, frames, tail)
    dt, _, _ = timed_infer(step, a.steps, a.warmup, barrier)
    dt = max_over_ranks(dt, dist, backend, dev)
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
        "data": "synthetic", "f32_matrix_arithmetic": f32_arithmetic_name(),
        "config": {"workload": f"{a.geometry} {sweeps}-sweep synthetic nuScenes-shaped frames, sparse_shape "
                               f"{[grid[2] + 1, grid[1], grid[0]]}, batch {a.batch}/GPU, inference, "
                               f"inputs resident in HBM (BASELINE.json {cfg_name})",
                   "frames_per_gpu": a.batch, "points_per_frame": int(np.mean([f.shape[0] for f in frames])),
                   "parallelism": f"frame-sharded x{world}",
                   "dispatch": "eager per-layer launches" if a.eager else engine.dispatch_name()
                   + (", two plans in flight: geometry graph of step k + 1 beside the feature graph of step k"
                      if a.pipeline and not a.full_net and a.streams == 1 else "")},
        "backend": backend, "world_size_seen": dist.get_world_size() if dist is not None else 1,
    }
    if rank == 0:
        # instrumented pass (HIP events around every C-ABI call, eager dispatch so that each launch can be bracketed):
        # NOT the pass `value` comes from
        ev_engine = engine.eager_twin()
        ev_step = make_step(a, gen, grid, ev_engine, frames, tail)
        # at most EV_STEPS steps: every record keeps its rulebook alive (pair counts are read afterwards, off the clock),
        # about 1 GB per batch-8 step
        ev_steps = min(a.steps, EV_STEPS)
        _, timing, _ = timed_infer(ev_step, ev_steps, 1, lambda: None, record=True)
        res["roofline"] = dominant_kernel_roofline(timing, a.dtype)
        res["phases"] = phase_report(timing, ev_steps, a.batch, a.geometry)
    return res, (gen, grid, engine, frames, tail)


def f32_arithmetic_name():
    from vista_amd import _lib
    p = int(_lib.lib().vx_spconv_get_f32_products())
    return ("fp32 MFMA (v_mfma_f32_32x32x2_f32)" if p == 0 else
            f"fp32 operands split exactly into 3 bf16 planes, {p} of 9 plane products on v_mfma_f32_32x32x16_bf16, "
            "fp32 accumulate (Cin >= 64 layers; error vs fp64 <= the fp32 MFMA's, tests/test_gpu_f32_products.py)")


def infer_variant(a, dev, rank, dtype_name, geometry, batch, steps, want_roofline=False, want_phases=False,
                  f32_products=None, pipeline=False):
    """One more configuration of BASELINE.json through the same code, for the `variants` object."""
    if f32_products is not None:
        from vista_amd import ops
        old = ops.set_f32_products(f32_products)
        try:
            out = infer_variant(a, dev, rank, dtype_name, geometry, batch, steps, want_roofline, want_phases)
            out["f32_matrix_arithmetic"] = f32_arithmetic_name()
        finally:
            ops.set_f32_products(old)
        return out
    gen, grid, engine, frames = build_infer(a, dev, rank, dtype_name, geometry, batch)
    step = make_step(a, gen, grid, engine, frames, pipeline=pipeline)
    dt, _, _ = timed_infer(step, steps, 2 if not pipeline else 4, lambda: None)
    out = {"frames_per_s": batch * steps / dt, "ms_per_step": 1e3 * dt / steps}
    if want_roofline or want_phases:
        ev_step = make_step(a, gen, grid, engine.eager_twin(), frames)
        ev_steps = min(steps, EV_STEPS)
        _, timing, _ = timed_infer(ev_step, ev_steps, 1, lambda: None, record=True)
        if want_roofline:
            out["roofline"] = dominant_kernel_roofline(timing, dtype_name)
        if want_phases:
            out["phases"] = phase_report(timing, ev_steps, batch, geometry)
    return out


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus is not None and a.gpus > 1:
        # launcher: nothing above this line has touched HIP (torch is not even imported yet)
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    import torch  # noqa: F401
    dev, rank, world, dist, backend, barrier = init_dist(a)

    if a.mode == "train":
        from vista_amd.train import bench_train
        res = bench_train(a, dev, rank, world, dist)
        res["backend"] = backend
        res["world_size_seen"] = dist.get_world_size() if dist is not None else 1
        if rank == 0:
            print(json.dumps(res), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    res, (gen, grid, engine, frames, tail) = run_infer(a, dev, rank, world, dist, backend, barrier)
    if rank == 0:
        default_cfg = (a.geometry == "FHD" and a.backbone == "SpMiddleResNetFHD" and not a.full_net
                       and a.streams == 1 and a.dtype == "fp32")
        if world == 1 and not a.no_variants:
            # batch-1 latency of the same workload (not part of `value`)
            st = max(5, a.steps)
            step1 = make_step(a, gen, grid, engine, frames[:1], tail)
            dt1, _, _ = timed_infer(step1, st, 3, lambda: None)
            res["variants"] = {"batch1_frames_per_s": st / dt1}
            if default_cfg:
                # the other single-GPU configurations BASELINE.json names, through the same code
                v = res["variants"]
                # the same fp32 workload on the fp32 MFMA and with all nine plane products
                v["f32_mfma_batch8"] = infer_variant(a, dev, rank, "fp32", "FHD", a.batch, a.steps, want_roofline=True,
                                                     f32_products=0)
                v["f32_split9_batch8"] = infer_variant(a, dev, rank, "fp32", "FHD", a.batch, a.steps,
                                                       want_roofline=True, f32_products=9)
                # throughput mode, two plans in flight (geometry of step k + 1 beside the convolutions of step k)
                v["pipelined_batch8"] = infer_variant(a, dev, rank, "fp32", "FHD", a.batch, a.steps, pipeline=True)
                v["pipelined_batch1"] = infer_variant(a, dev, rank, "fp32", "FHD", 1, st, pipeline=True)
                v["bf16_batch8"] = infer_variant(a, dev, rank, "bf16", "FHD", a.batch, a.steps, want_roofline=True)
                v["bf16_batch1"] = infer_variant(a, dev, rank, "bf16", "FHD", 1, st)
                v["stress_batch1"] = infer_variant(a, dev, rank, "fp32", "STRESS", 1, 5, want_phases=True)
                # configs[2] on one GPU: the full VISTA net (two-branch sparse backbone + dense tail), batch 4
                av = parse(["--geometry", "VISTA", "--backbone", "MultiViewBackbone", "--full-net", "--batch", "4",
                            "--steps", str(a.steps), "--warmup", "3"])
                rv, _ = run_infer(av, dev, 0, 1, None, None, lambda: None)
                v["vista_fullnet_batch4"] = {"frames_per_s": rv["value"], "ms_per_step": rv["ms_per_step"],
                                             "metric": rv["metric"]}
                from vista_amd.train import bench_train
                at = parse(["--mode", "train", "--steps", "4", "--warmup", "2"])
                tr = bench_train(at, dev, rank, 1, None)
                v["train_b4_ms_per_step"] = tr["ms_per_step"]
                v["train_b4_frames_per_s"] = tr["value"]
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.geometry)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
