"""fp32 convolution through split-bf16 plane products (vx_spconv_set_f32_products): error against the fp64 oracle and
launch time next to the fp32 MFMA, on the FHD 128->128 / 64->64-stage rulebooks.  usage: python tools/split_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vista_amd import ops
from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
from vista_amd.voxel import VoxelGenerator, voxelize_batch
from oracle import spconv_oracle as orc

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
rg, vs, T, mv = GEOMETRIES["FHD"]
gen = VoxelGenerator(vs, rg, T, mv)
clouds = [torch.from_numpy(lidar_cloud(seed=b, sweeps=10)).to(dev) for b in range(batch)]
mean, coords, _ = voxelize_batch(gen, clouds)
g = grid_size(rg, vs).tolist()
shape = [g[2] + 1, g[1], g[0]]
idx = coords
for k, s, p in [(3, 2, 1), (3, 2, 1), (3, 2, [0, 1, 1])]:
    rbc = ops.rulebook_conv(idx, batch, shape, k, s, p)
    idx, shape = rbc.out_indices, rbc.out_shape
rb = ops.rulebook_subm(idx, batch, shape, 3)
rng = np.random.default_rng(0)
pairs = int((rb.nbr >= 0).sum().item())
print(f"batch {batch}: n={rb.n_in} pairs={pairs}", flush=True)
for cin, cout in ((128, 128), (64, 128), (128, 64)):
    x = rng.normal(size=(rb.n_in, cin)).astype(np.float32)
    x *= np.exp(rng.normal(size=x.shape) * 2).astype(np.float32)           # wide dynamic range
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    nsub = 4096                                                              # fp64 truth on the first rows
    time_only = bool(os.environ.get("VX_PROBE_TIME_ONLY"))
    if time_only and (cin, cout) != (128, 128):
        continue
    nbr = rb.nbr[:, :nsub].cpu().numpy()
    ref = orc.conv_fwd(x, w, nbr, dtype=np.float64)
    mag = orc.conv_fwd(np.abs(x), np.abs(w), nbr, dtype=np.float64)          # sum |a||b| per output
    for mode in [int(m) for m in os.environ.get("VX_PROBE_MODES", "0,9,6").split(",")]:
        ops.set_f32_products(mode)
        for _ in range(2):
            out = ops.conv_fwd(xd, wd, rb, compact=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = ops.conv_fwd(xd, wd, rb, compact=False)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        got = out[:nsub].cpu().numpy().astype(np.float64)
        err = np.abs(got - ref)
        print(f"{cin:3d}->{cout:3d} products={mode}: {ms:7.3f} ms {2.0 * pairs * cin * cout / ms / 1e9:7.1f} TFLOP/s   "
              f"max|err|/max|ref| {err.max() / np.abs(ref).max():.2e}   max err/sum|a||b| {np.max(err / np.maximum(mag, 1e-30)):.2e}   "
              f"rms err/rms ref {np.sqrt((err ** 2).mean() / (ref ** 2).mean()):.2e}", flush=True)
    ops.set_f32_products(0)
