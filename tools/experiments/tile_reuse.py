"""How many DISTINCT input rows does a 256-row tile of a submanifold conv touch over its 27 offsets?
(CPU only: oracle rulebooks of one synthetic FHD frame.)  Answers whether gathering the union of a tile's inputs once
into LDS, instead of once per offset, would cut the L2->LDS fill."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import spconv_oracle as so, voxelize_np as vz
from vista_amd import synth

rng, vs, mp_, mv = synth.GEOMETRIES["FHD"]
pts = synth.lidar_cloud(0)
_, coords, _ = vz.voxelize(pts, vs, rng, mp_, mv)[:3]
coords = np.concatenate([np.zeros((coords.shape[0], 1), np.int32), coords.astype(np.int32)], 1)
grid = synth.grid_size(rng, vs)
shape = (np.array(grid[::-1]) + [1, 0, 0]).tolist()
stages = [(3, 2, 1), (3, 2, 1), (3, 2, [0, 1, 1])]
names = ["16ch", "32ch", "64ch", "128ch"]
for si in range(4):
    nbr = so.rulebook_subm(coords, shape, 3)
    n = coords.shape[0]
    pairs = int((nbr >= 0).sum())
    for T in (256, 512):
        tot_u = tot_p = 0
        for t0 in range(0, n, T):
            blk = nbr[:, t0:t0 + T]
            v = blk[blk >= 0]
            tot_u += np.unique(v).size
            tot_p += v.size
        # per (kz,ky) scan-line group: union over the 3 kx offsets
        tot_g = 0
        for t0 in range(0, n, T):
            for g in range(9):
                blk = nbr[3 * g:3 * g + 3, t0:t0 + T]
                tot_g += np.unique(blk[blk >= 0]).size
        print(f"{names[si]} rows {n} pairs {pairs} ({pairs / n:.1f}/row) tile {T}: unique/tile {tot_u / ((n + T - 1) // T):.0f} "
              f"pairs/unique {tot_p / tot_u:.2f}; per scan-line group pairs/unique {tot_p / tot_g:.2f}")
    if si < 3:
        coords, _, shape = so.rulebook_conv(coords, shape, *stages[si])
