"""CPU: the numpy voxelizer oracle against the golden vectors the REFERENCE produced
(tests/golden/voxel_*.npz, made by tools/gen_golden.py from
det3d/ops/point_cloud/point_cloud_ops.py:138-221 and :224-296)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import voxelize_np as ov
from vista_amd.synth import lidar_cloud

G = os.path.join(os.path.dirname(__file__), "golden")
MAN = json.load(open(os.path.join(G, "voxel_manifest.json")))
FULL = [k for k in MAN if not k.startswith("_") and os.path.exists(os.path.join(G, f"voxel_{k}.npz"))]
BIG = [k for k in MAN if not k.startswith("_") and os.path.exists(os.path.join(G, f"voxel_{k}_sample.npz"))]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", FULL)
def test_oracle_matches_reference_full(name):
    m = MAN[name]
    d = np.load(os.path.join(G, f"voxel_{name}.npz"))
    v, c, n = ov.voxelize(d["points"], m["voxel_size"], m["range"], m["max_points"], m["max_voxels"])
    assert c.dtype == np.int32 and n.dtype == np.int32
    np.testing.assert_array_equal(c, d["coords"])
    np.testing.assert_array_equal(n, d["num_points"])
    np.testing.assert_array_equal(v, d["voxels"])           # bit-exact copy of the points
    if c.shape[0]:
        np.testing.assert_allclose(ov.mean_vfe(v, n), d["mean"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["edge_bounds", "overfull_voxels", "max_voxels_trunc_small", "single_point"])
def test_loop_restatement_matches_reference(name):
    m = MAN[name]
    d = np.load(os.path.join(G, f"voxel_{name}.npz"))
    v, c, n = ov.voxelize_loop(d["points"], m["voxel_size"], m["range"], m["max_points"], m["max_voxels"])
    np.testing.assert_array_equal(c, d["coords"])
    np.testing.assert_array_equal(n, d["num_points"])
    np.testing.assert_array_equal(v, d["voxels"])


@pytest.mark.parametrize("name", BIG)
def test_oracle_matches_reference_big(name):
    m = MAN[name]
    pts = lidar_cloud(seed=m["seed"], sweeps=m["sweeps"])
    assert sha(pts) == MAN["_points_sha256"][f"seed{m['seed']}_sweeps{m['sweeps']}"], \
        "synthetic generator drifted from the one the golden vectors were made with"
    v, c, n = ov.voxelize(pts, m["voxel_size"], m["range"], m["max_points"], m["max_voxels"])
    assert c.shape[0] == m["M"]
    assert sha(c) == m["coords_sha256"]
    assert sha(n) == m["num_points_sha256"]
    assert sha(v) == m["voxels_sha256"]
    s = np.load(os.path.join(G, f"voxel_{name}_sample.npz"))
    np.testing.assert_array_equal(c[s["rows"]], s["coords"])
    np.testing.assert_array_equal(v[s["rows"]], s["voxels"])


def test_collate_coords():
    a = np.array([[1, 2, 3]], np.int32)
    b = np.array([[4, 5, 6], [7, 8, 9]], np.int32)
    out = ov.collate_coords([a, b])
    np.testing.assert_array_equal(out, [[0, 1, 2, 3], [1, 4, 5, 6], [1, 7, 8, 9]])
