#!/usr/bin/env python3
"""Generate tests/golden/voxel_*.npz by running the REFERENCE's own voxelizer.

Runs ONLY in the build container (needs /root/reference); the GPU box never
sees the reference. Fixtures are data (inputs + the reference's outputs).

The reference's module imports `numba` (absent here, no network). Its two
numba-decorated helpers on this path are plain Python loops, so a `numba`
module whose `jit` is the identity decorator is installed in sys.modules
before the import; the code that runs is the reference's own
(det3d/ops/point_cloud/point_cloud_ops.py:112-221, :7-55, :224-296) and
det3d/core/input/voxel_generator.py:6-39 (loaded by file path because the
det3d.core package import pulls torchvision).
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def _load_reference():
    nb = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.jit = jit
    nb.njit = jit
    sys.modules["numba"] = nb
    sys.path.insert(0, REF)
    from det3d.ops.point_cloud import point_cloud_ops as pco  # noqa
    spec = importlib.util.spec_from_file_location(
        "ref_voxel_generator", os.path.join(REF, "det3d/core/input/voxel_generator.py"))
    vg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vg)
    return pco, vg


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    from vista_amd.synth import GEOMETRIES, lidar_cloud
    pco, vg = _load_reference()
    os.makedirs(OUT, exist_ok=True)
    manifest = {}

    def run(name, pts, rng_, vs, T, maxv, full=True, also_loop=False, meta=None):
        gen = vg.VoxelGenerator(vs, rng_, T, max_voxels=maxv, faster=True)
        try:
            voxels, coords, nump = gen.generate(pts)
            err = None
        except Exception as e:  # the reference itself raises (e.g. empty input)
            voxels = coords = nump = None
            err = f"{type(e).__name__}: {e}"
        entry = dict(range=list(map(float, rng_)), voxel_size=list(map(float, vs)),
                     max_points=T, max_voxels=maxv, n_points=int(pts.shape[0]),
                     grid_size=[int(g) for g in gen.grid_size], error=err)
        if meta:
            entry.update(meta)
        if err is None:
            entry.update(M=int(coords.shape[0]), coords_dtype=str(coords.dtype),
                         num_points_dtype=str(nump.dtype),
                         coords_sha256=sha(coords.astype(np.int32)),
                         num_points_sha256=sha(nump.astype(np.int32)),
                         voxels_sha256=sha(voxels.astype(np.float32)))
            if also_loop:
                gen2 = vg.VoxelGenerator(vs, rng_, T, max_voxels=maxv, faster=False)
                v2, c2, n2 = gen2.generate(pts)
                entry["loop_path_identical"] = bool(
                    np.array_equal(v2, voxels) and np.array_equal(c2, coords)
                    and np.array_equal(n2, nump))
            if full:
                mean = voxels[:, :, :].sum(axis=1) / nump.astype(np.float32).reshape(-1, 1) \
                    if coords.shape[0] else np.zeros((0, pts.shape[1]), np.float32)
                np.savez_compressed(
                    os.path.join(OUT, f"voxel_{name}.npz"),
                    points=pts.astype(np.float32), voxels=voxels.astype(np.float32),
                    coords=coords.astype(np.int32), num_points=nump.astype(np.int32),
                    mean=mean.astype(np.float32))
            else:
                # large case: seed + hashes + sampled rows only
                r = np.random.default_rng(7)
                rows = np.sort(r.choice(coords.shape[0], size=min(256, coords.shape[0]), replace=False))
                np.savez_compressed(
                    os.path.join(OUT, f"voxel_{name}_sample.npz"),
                    rows=rows.astype(np.int64), voxels=voxels[rows].astype(np.float32),
                    coords=coords[rows].astype(np.int32), num_points=nump[rows].astype(np.int32))
        manifest[name] = entry
        print(name, {k: entry[k] for k in ("n_points", "error") if k in entry}, entry.get("M"))

    # --- small full fixtures --------------------------------------------
    r, vs, T, mv = GEOMETRIES["TINY"]
    cloud1 = lidar_cloud(seed=3, sweeps=1)
    run("tiny_1sweep", cloud1, r, vs, T, mv, also_loop=True)
    r2, vs2, T2, mv2 = GEOMETRIES["FHD"]
    rng = np.random.default_rng(11)
    sub = lidar_cloud(seed=5, sweeps=2)
    sub = sub[np.sort(rng.choice(sub.shape[0], 20000, replace=False))]
    run("fhd_20k", sub, r2, vs2, T2, mv2, also_loop=False)
    r3, vs3, T3, mv3 = GEOMETRIES["VISTA"]
    run("vista_20k", sub, r3, vs3, T3, mv3)

    # --- edge cases -------------------------------------------------------
    lo = np.array(r[:3], np.float32); hi = np.array(r[3:], np.float32); v = np.array(vs, np.float32)
    edge = []
    edge.append(np.r_[lo, 1, 0])                                   # exactly on lo -> voxel 0
    edge.append(np.r_[hi, 2, 0])                                   # exactly on hi -> dropped
    edge.append(np.r_[hi - v * 0.5, 3, 0])                         # last voxel
    edge.append(np.r_[np.nextafter(hi, -np.inf, dtype=np.float32), 4, 0])
    edge.append(np.r_[np.nextafter(lo, -np.inf, dtype=np.float32), 5, 0])  # just below lo
    edge.append(np.array([-0.0, -0.0, -0.0, 6, 0]))                # negative zero
    edge.append(np.array([0.0, 0.0, 0.0, 7, 0]))
    for k in range(1, 40):                                         # voxel boundaries k*vs
        edge.append(np.r_[lo + v * k, 8, 0])
        edge.append(np.r_[np.nextafter(lo + v * k, -np.inf, dtype=np.float32), 9, 0])
    dup = np.tile(np.array([[1.23, -4.56, 0.1, 10, 0]]), (12, 1))  # 12 duplicates > T=5
    dup[:, 3] = np.arange(12)
    edge = np.concatenate([np.stack(edge), dup]).astype(np.float32)
    run("edge_bounds", edge, r, vs, T, mv, also_loop=True)

    dense = np.random.default_rng(2).uniform(-1.0, 1.0, (4000, 5)).astype(np.float32)
    dense[:, 2] = np.random.default_rng(3).uniform(-0.5, 0.5, 4000)
    run("overfull_voxels", dense, r, vs, 3, mv, also_loop=True)      # count > T in most voxels
    run("max_voxels_trunc", cloud1, r, vs, T, 500, also_loop=True)  # M > max_voxels
    run("max_voxels_trunc_small", dense, r, vs, 3, 37, also_loop=True)
    run("single_point", np.array([[0.1, 0.2, 0.3, 1, 0]], np.float32), r, vs, T, mv, also_loop=True)
    run("all_out_of_range", np.full((7, 5), 1000.0, np.float32), r, vs, T, mv)
    run("empty", np.zeros((0, 5), np.float32), r, vs, T, mv)

    # --- large configs: hashes + sampled rows -----------------------------
    big = lidar_cloud(seed=0, sweeps=10)
    run("fhd_10sweep_seed0", big, r2, vs2, T2, mv2, full=False, meta=dict(seed=0, sweeps=10))
    run("vista_10sweep_seed0", big, r3, vs3, T3, mv3, full=False, meta=dict(seed=0, sweeps=10))
    r4, vs4, T4, mv4 = GEOMETRIES["STRESS"]
    stress = lidar_cloud(seed=0, sweeps=40)
    run("stress_40sweep_seed0", stress, r4, vs4, T4, mv4, full=False, meta=dict(seed=0, sweeps=40))
    manifest["_points_sha256"] = dict(seed0_sweeps10=sha(big), seed0_sweeps40=sha(stress))

    with open(os.path.join(OUT, "voxel_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
