"""The hash / voxelize / rulebook / dense kernels of one step, alone (for rocprofv3 --pmc and --kernel-trace):
usage: python tools/hash_probe.py GEOMETRY BATCH [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vista_amd import ops
from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
from vista_amd.voxel import VoxelGenerator, voxelize_batch
geom, batch = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
rg, vs, T, mv = GEOMETRIES[geom]
gen = VoxelGenerator(vs, rg, T, mv)
clouds = [torch.from_numpy(lidar_cloud(seed=b, sweeps=40 if geom == "STRESS" else 10)).to(dev) for b in range(batch)]
g = grid_size(rg, vs).tolist()
for _ in range(reps):
    mean, coords, _ = voxelize_batch(gen, clouds)
    shape = [g[2] + 1, g[1], g[0]]
    idx, rank = coords, None
    ops.rulebook_subm(idx, batch, shape, 3)
    for k, s, p in [(3, 2, 1), (3, 2, 1), (3, 2, [0, 1, 1]), ([3, 1, 1], [2, 1, 1], 0)]:
        rbc = ops.rulebook_conv(idx, batch, shape, k, s, p)
        idx, shape, rank = rbc.out_indices, rbc.out_shape, rbc.out_rank
        if k == 3:
            ops.rulebook_subm(idx, batch, shape, 3, rank=rank)
    feat = torch.zeros((idx.shape[0], 128), device=dev)
    ops.to_dense(feat, idx, batch, shape)
torch.cuda.synchronize()
print("points", sum(c.shape[0] for c in clouds), "voxels", coords.shape[0], "last rows", idx.shape[0])
