"""Time individual sparse-conv layers of the FHD backbone on real rulebooks (diagnostic tool).
usage: python tools/conv_probe.py [batch]   (VX_LIB=<alternative libvx.so> to time a diagnostic build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vista_amd import ops
from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
from vista_amd.voxel import VoxelGenerator, voxelize_batch

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
use_ws = len(sys.argv) > 2 and sys.argv[2] == "ws"
sort_rows = len(sys.argv) > 3 and sys.argv[3] == "sort"
compact = not (len(sys.argv) > 2 and sys.argv[2] == "nocb")
dev = torch.device("cuda:0")
rg, vs, T, mv = GEOMETRIES["FHD"]
gen = VoxelGenerator(vs, rg, T, mv)
clouds = [torch.from_numpy(lidar_cloud(seed=b, sweeps=10)).to(dev) for b in range(batch)]
mean, coords, _ = voxelize_batch(gen, clouds)
g = grid_size(rg, vs).tolist()
shape = [g[2] + 1, g[1], g[0]]
idx = coords
stages = []
rb = ops.rulebook_subm(idx, batch, shape, 3); stages.append(("subm0", 16, 16, rb))
for name, cin, cout, k, s, p in [("conv1", 16, 32, 3, 2, 1), ("conv2", 32, 64, 3, 2, 1), ("conv3", 64, 128, 3, 2, [0, 1, 1])]:
    rbc = ops.rulebook_conv(idx, batch, shape, k, s, p)
    stages.append((name, cin, cout, rbc))
    idx, shape = rbc.out_indices, rbc.out_shape
    rbs = ops.rulebook_subm(idx, batch, shape, 3)
    stages.append(("subm" + name[-1], cout, cout, rbs))
rng = np.random.default_rng(0)
print(f"lib={os.environ.get('VX_LIB', 'libvx.so')} batch={batch} dtype={dt}")
for name, cin, cout, rb in stages:
    x = torch.from_numpy(rng.normal(size=(rb.n_in, cin)).astype(np.float32)).to(dev).to(dt)
    w = torch.from_numpy((rng.normal(size=(cout, *rb.ksize, cin)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev).to(dt)
    pairs = int((rb.nbr >= 0).sum().item())
    wp = ops.pack_weight(w) if use_ws else None
    for _ in range(2):
        ops.conv_fwd(x, w, rb, wpack=wp, sort_rows=sort_rows, compact=compact)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ops.conv_fwd(x, w, rb, wpack=wp, sort_rows=sort_rows, compact=compact)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if compact:
        ref = ops.conv_fwd(x, w, rb, compact=False)
        got = ops.conv_fwd(x, w, rb, compact=True)
        err = float((got - ref).abs().max() / ref.abs().max())
        name = f"{name}[err {err:.1e}]"
    fl = 2.0 * pairs * cin * cout
    print(f"{name:6s} {cin:3d}->{cout:3d} n_out={rb.n_out:8d} pairs={pairs:9d} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  "
          f"gather {pairs * cin * x.element_size() / ms / 1e6:7.1f} GB/s")
