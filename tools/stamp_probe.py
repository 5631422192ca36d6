"""Diagnostic: per-segment cycle shares of the spconv_w8 pipeline step (needs a -DVX_STAMP build of libvx,
path in VX_LIB). Segments: wait(cur loads) | issue prefetch | MFMA chunk | store W | barrier."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vista_amd import ops, _lib
from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
from vista_amd.voxel import VoxelGenerator, voxelize_batch
batch = 8
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
dev = torch.device("cuda:0")
rg, vs, T, mv = GEOMETRIES["FHD"]
gen = VoxelGenerator(vs, rg, T, mv)
mean, coords, _ = voxelize_batch(gen, [torch.from_numpy(lidar_cloud(seed=b, sweeps=10)).to(dev) for b in range(batch)])
g = grid_size(rg, vs).tolist(); shape = [g[2] + 1, g[1], g[0]]; idx = coords
stages = []
for name, cin, cout, k, s, p in [("conv1", 16, 32, 3, 2, 1), ("conv2", 32, 64, 3, 2, 1), ("conv3", 64, 128, 3, 2, [0, 1, 1])]:
    rbc = ops.rulebook_conv(idx, batch, shape, k, s, p); idx, shape = rbc.out_indices, rbc.out_shape
    stages.append(("subm" + name[-1], cout, ops.rulebook_subm(idx, batch, shape, 3, rank=rbc.out_rank)))
L = _lib.lib(); buf = (ctypes.c_ulonglong * 8)()
rng = np.random.default_rng(0)
for name, c, rb in stages:
    x = torch.from_numpy(rng.normal(size=(rb.n_in, c)).astype(np.float32)).to(dev).to(dt)
    w = torch.from_numpy((rng.normal(size=(c, 3, 3, 3, c)) / np.sqrt(27 * c)).astype(np.float32)).to(dev).to(dt)
    ops.conv_fwd(x, w, rb); torch.cuda.synchronize(); L.vx_debug_read(buf)
    ops.conv_fwd(x, w, rb); torch.cuda.synchronize(); L.vx_debug_read(buf)
    v = np.array(list(buf)[:6], dtype=np.float64); tot = v[:5].sum()
    print(f"{name} C={c} {dt}: steps/wave-sum={int(v[5])}  cycles/step: wait {v[0]/v[5]:.0f} issue {v[1]/v[5]:.0f} "
          f"mfma {v[2]/v[5]:.0f} storeW {v[3]/v[5]:.0f} barrier {v[4]/v[5]:.0f} | shares " +
          " ".join(f"{100*x/tot:.1f}%" for x in v[:5]))
