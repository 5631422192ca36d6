"""Do two different sparse-conv kernels finish sooner side by side on two HIP streams than one after the other?
(8 FHD frames; each pair of layers is timed back to back on one stream and concurrently on two.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vista_amd import ops
from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
from vista_amd.voxel import VoxelGenerator, voxelize_batch
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
rg, vs, T, mv = GEOMETRIES["FHD"]
gen = VoxelGenerator(vs, rg, T, mv)
mean, coords, _ = voxelize_batch(gen, [torch.from_numpy(lidar_cloud(seed=b, sweeps=10)).to(dev) for b in range(batch)])
g = grid_size(rg, vs).tolist(); shape = [g[2] + 1, g[1], g[0]]; idx = coords
L = {}
rng = np.random.default_rng(0)
def layer(name, c, rb):
    x = torch.from_numpy(rng.normal(size=(rb.n_in, c)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.normal(size=(c, 3, 3, 3, c)) / np.sqrt(27 * c)).astype(np.float32)).to(dev)
    L[name] = (x, w, rb)
layer("s16", 16, ops.rulebook_subm(idx, batch, shape, 3))
for name, cin, cout, k, s, p in [("c1", 16, 32, 3, 2, 1), ("c2", 32, 64, 3, 2, 1), ("c3", 64, 128, 3, 2, [0, 1, 1])]:
    rbc = ops.rulebook_conv(idx, batch, shape, k, s, p); idx, shape = rbc.out_indices, rbc.out_shape
    layer("s%d" % cout, cout, ops.rulebook_subm(idx, batch, shape, 3, rank=rbc.out_rank))
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def run(name, n):
    x, w, rb = L[name]
    for _ in range(n):
        ops.conv_fwd(x, w, rb)
def timed(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3
def seq(a, na, b, nb):
    def f():
        run(a, na); run(b, nb)
    return f
def con(a, na, b, nb):
    def f():
        main = torch.cuda.current_stream()
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1): run(a, na)
        with torch.cuda.stream(s2): run(b, nb)
        main.wait_stream(s1); main.wait_stream(s2)
    return f
for a, na, b, nb in [("s128", 2, "s32", 5), ("s128", 2, "s64", 3), ("s64", 2, "s32", 4), ("s128", 2, "s16", 16), ("s128", 2, "s128", 2),
                     ("s64", 2, "s64", 2), ("s32", 4, "s32", 4)]:
    ts, tc = timed(seq(a, na, b, nb)), timed(con(a, na, b, nb))
    print(f"{na} x {a} + {nb} x {b}: sequential {ts:.3f} ms, two streams {tc:.3f} ms ({100 * (tc / ts - 1):+.1f} %)", flush=True)
