"""Offline model (CPU, oracle rulebooks) of wasted MFMA row slots in the output-stationary conv kernel:
useful fraction = pairs / (row slots of groups that have at least one neighbour at that offset),
for different row orders and row-group sizes, plus a chunk-synchronous workgroup time model.
Usage: python tools/waste_model.py  (one FHD frame; stages 1..3)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import spconv_oracle as so, voxelize_np as ov
from vista_amd.synth import GEOMETRIES, lidar_cloud, grid_size

rg, vs, T, mv = GEOMETRIES["FHD"]
pts = lidar_cloud(seed=0, sweeps=10)
v, c, n = ov.voxelize(pts, vs, rg, T, mv)
coords = ov.collate_coords([c])
g = grid_size(rg, vs).tolist()
shape = [g[2] + 1, g[1], g[0]]


def model(nbr, order, name):
    # nbr [K, N]; order: permutation of rows
    K, N = nbr.shape
    act = (nbr[:, order] >= 0)
    pairs = act.sum()
    out = [f"{name:28s} N={N:7d} pairs/N={pairs / N:5.2f}"]
    for gs in (32, 16, 8):
        pad = (-N) % gs
        a = np.pad(act, ((0, 0), (0, pad))).reshape(K, -1, gs)
        grp = a.any(2)                      # [K, groups]
        out.append(f"g{gs}: useful {pairs / (grp.sum() * gs):.3f}")
    # workgroup model: 256 rows, 8 waves x 32 rows, waves w and w+4 share a SIMD; per offset the workgroup
    # takes max over SIMDs of the active groups (barrier per chunk); ideal = sum/4
    pad = (-N) % 256
    a = np.pad(act, ((0, 0), (0, pad))).reshape(K, -1, 8, 32).any(3)      # [K, tiles, 8]
    simd = a[:, :, :4].astype(int) + a[:, :, 4:].astype(int)              # [K, tiles, 4]
    t_sync = simd.max(2).sum()
    t_now = 2 * a.any(2).sum()              # today: any wave active -> all behave as active? upper bound
    t_ideal_groups = a.sum() / 4.0
    t_pairs = pairs / 32.0 / 4.0
    out.append(f"WG time units: wg-skip-only {t_now}, simd-max {t_sync}, perfect-balance {t_ideal_groups:.0f}, pairs-only {t_pairs:.0f}")
    # 16-row groups, 4 per wave pair... per SIMD 4 groups of 16: time = max over SIMDs of active16/2
    a16 = np.pad(act, ((0, 0), (0, pad))).reshape(K, -1, 8, 2, 16).any(4)  # [K,tiles,8,2]
    s16 = (a16[:, :, :4].sum(3) + a16[:, :, 4:].sum(3)) / 2.0
    out.append(f"simd-max(16-row) {s16.max(2).sum():.0f}")
    print("  ".join(out))


def orders(co):
    b, z, y, x = co.T.astype(np.int64)
    lin = np.lexsort((x, y, z, b))
    res = {"linear zyx": lin}
    res["blocked y4 x8"] = np.lexsort((x % 8, y % 4, x // 8, y // 4, z, b))
    res["blocked z2 y4 x4"] = np.lexsort((x % 4, y % 4, z % 2, x // 4, y // 4, z // 2, b))
    res["yx-major z-fastest"] = np.lexsort((z, x, y, b))
    res["blocked y2 x2 zfast"] = np.lexsort((z, x % 4, y % 4, x // 4, y // 4, b))
    return res


cur, shp = coords, shape
convs = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
for stage in range(4):
    for nm, od in orders(cur).items():
        co = np.ascontiguousarray(cur[od])
        nbr = so.rulebook_subm(co, shp, (3, 3, 3))
        model(np.asarray(nbr), np.arange(co.shape[0]), f"stage{stage} {nm}")
    if stage < 3:
        ks, st, pd = convs[stage]
        cur, _, shp = so.rulebook_conv(cur, shp, ks, st, pd)
        shp = list(shp)


def lockstep_model(nbr, name):
    """Design B (per-offset compaction, LDS accumulators): time units (one 32-pair group on one SIMD) for a tile of
    R rows whose waves advance window by window (W of `win` offsets resident), groups dealt evenly to 4 SIMDs."""
    K, N = nbr.shape
    act = nbr >= 0
    out = [name]
    for R in (128, 256, 512):
        pad = (-N) % R
        p = np.pad(act, ((0, 0), (0, pad))).reshape(K, -1, R).sum(2)         # [K, tiles] pairs per offset
        grp = -(-p // 32)
        now = 2 * (R // 256 if R >= 256 else 0.5) * (p > 0).sum()
        for win in (1, 2, 3, 4, 9, 27):
            kpad = (-K) % win
            g = np.pad(grp, ((0, kpad), (0, 0))).reshape(-1, win, grp.shape[1]).sum(1)
            t = (-(-g // 4)).sum()
            out.append(f"R{R} win{win}: {t / (act.sum() / 128.0):.2f}")
        out.append(f"| now {now / (act.sum() / 128.0):.2f}")
    print("  ".join(out))


print("time relative to pairs-only bound (1.00 = no wasted MFMA rows, perfectly balanced)")
cur, shp = coords, shape
for stage in range(4):
    nbr = np.asarray(so.rulebook_subm(np.ascontiguousarray(cur), shp, (3, 3, 3)))
    lockstep_model(nbr, f"stage{stage}")
    if stage < 3:
        ks, st, pd = convs[stage]
        cur, _, shp = so.rulebook_conv(cur, shp, ks, st, pd)
        shp = list(shp)


def half64_model(nbr, name):
    """Design-B variant: each 64-row half of a 128-row tile compacts its own pairs per offset into groups of 32."""
    K, N = nbr.shape
    act = nbr >= 0
    pad = (-N) % 64
    p = np.pad(act, ((0, 0), (0, pad))).reshape(K, -1, 64).sum(2)
    slots = (-(-p // 32)) * 32
    print(f"{name}: half64 compaction: MFMA row slots / pairs = {slots.sum() / act.sum():.2f} "
          f"(32-row groups today: {(np.pad(act, ((0, 0), (0, (-N) % 32))).reshape(K, -1, 32).any(2).sum() * 32) / act.sum():.2f})")


print("per-64-row-half compaction")
cur, shp = coords, shape
for stage in range(4):
    nbr = np.asarray(so.rulebook_subm(np.ascontiguousarray(cur), shp, (3, 3, 3)))
    half64_model(nbr, f"stage{stage} subm")
    if stage < 3:
        ks, st, pd = convs[stage]
        cur2, nbrc, shp2 = so.rulebook_conv(cur, shp, ks, st, pd)
        half64_model(np.asarray(nbrc), f"stage{stage}->{stage + 1} strided conv")
        cur, shp = cur2, list(shp2)
