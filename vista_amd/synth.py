"""Deterministic synthetic nuScenes-shaped LiDAR clouds (no dataset, no network).

Follows the recipe fixed in SURVEY.md section 8(d): a 32-beam spinning sensor,
1084 azimuth steps per beam per sweep, sensor 1.84 m above a ground plane,
multi-sweep accumulation with a per-sweep ego offset and a time-lag feature.
Point layout is ``float32[N, 5] = (x, y, z, intensity, dt)`` in sweep-major
order, which is what the reference's loader hands to the voxelizer
(det3d/datasets/pipelines/loading.py:93-99, det3d/datasets/nuscenes/nusc_common.py:250-252).
"""
import numpy as np

# geometry presets of the path
GEOMETRIES = {
    # name: (point_cloud_range, voxel_size, max_points_per_voxel, max_voxels)
    "FHD": ([-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], [0.075, 0.075, 0.2], 10, 5_000_000),
    "VISTA": ([-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.08, 0.08, 0.08], 6, 5_000_000),
    "STRESS": ([-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], [0.05, 0.05, 0.2], 10, 5_000_000),
    # small geometries for fast parity tests (SMALL keeps 40 z cells so all 4 stride-2 stages exist)
    "SMALL": ([-12.8, -12.8, -5.0, 12.8, 12.8, 3.0], [0.2, 0.2, 0.2], 5, 100_000),
    "TINY": ([-12.8, -12.8, -3.0, 12.8, 12.8, 1.0], [0.2, 0.2, 0.2], 5, 100_000),
}

N_BEAMS = 32
N_AZIMUTH = 1084
SENSOR_HEIGHT = 1.84


def lidar_cloud(seed: int = 0, sweeps: int = 10) -> np.ndarray:
    """Return one synthetic multi-sweep frame as float32[N, 5]."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-30.67, 10.67, N_BEAMS))
    azim = np.linspace(-np.pi, np.pi, N_AZIMUTH, endpoint=False)
    out = []
    for s in range(sweeps):
        el = np.repeat(elev, N_AZIMUTH)
        az = np.tile(azim, N_BEAMS) + rng.uniform(0, 2 * np.pi / N_AZIMUTH)
        n = el.shape[0]
        # range: ground-plane hit for downward beams, else a far random return
        with np.errstate(divide="ignore", invalid="ignore"):
            r_ground = np.where(el < 0, SENSOR_HEIGHT / np.maximum(-np.sin(el), 1e-6), np.inf)
        r_far = rng.uniform(0.2, 1.0, n) * 100.0
        r = np.minimum(r_ground, r_far)
        # 35 % of returns are shortened by objects / walls
        short = rng.uniform(0, 1, n) < 0.35
        r = np.where(short, r * rng.uniform(0.15, 1.0, n), r)
        r = r + rng.normal(0.0, 0.02, n)
        keep = (rng.uniform(0, 1, n) >= 0.28) & (r > 1.0) & (r < 100.0)
        r, el, az = r[keep], el[keep], az[keep]
        x = r * np.cos(el) * np.cos(az) + 0.5 * s
        y = r * np.cos(el) * np.sin(az)
        z = r * np.sin(el)  # sensor frame: ground near z = -1.84
        inten = rng.uniform(0, 255, r.shape[0])
        dt = np.full(r.shape[0], 0.05 * s)
        out.append(np.stack([x, y, z, inten, dt], axis=1))
    return np.concatenate(out, axis=0).astype(np.float32)


def grid_size(pc_range, voxel_size) -> np.ndarray:
    """[X, Y, Z] int64 grid, computed in fp32 like det3d/core/input/voxel_generator.py:8-12."""
    r = np.asarray(pc_range, dtype=np.float32)
    v = np.asarray(voxel_size, dtype=np.float32)
    return np.round((r[3:] - r[:3]) / v).astype(np.int64)


def backbone_weights(layer_shapes, seed: int = 1234):
    """Seeded conv weights normal(0, 1/sqrt(Kvol*Cin)), layout [Cout, kz, ky, kx, Cin]."""
    rng = np.random.default_rng(seed)
    ws = []
    for (cout, kz, ky, kx, cin) in layer_shapes:
        std = 1.0 / np.sqrt(kz * ky * kx * cin)
        ws.append(rng.normal(0.0, std, (cout, kz, ky, kx, cin)).astype(np.float32))
    return ws


# ---------------------------------------------------------------------------
# Seeded parameters, keyed by the reference's state_dict names
# (backbone.middle_conv.<i>[.conv1|.bn1|...], scn.py:332-364, multiview.py:33-86).
# ---------------------------------------------------------------------------
def _seq_entries(prefix, layers):
    """layers: list of ('subm'|'conv', cin, cout, ksize) | ('bn', c) | ('relu',) | ('block', c)."""
    out = []
    for i, l in enumerate(layers):
        p = f"{prefix}.{i}"
        if l[0] in ("subm", "conv"):
            out.append((f"{p}.weight", ("w", l[1], l[2], l[3])))
        elif l[0] == "bn":
            out.append((p, ("bn", l[1])))
        elif l[0] == "block":
            c = l[1]
            out.append((f"{p}.conv1.weight", ("w", c, c, (3, 3, 3))))
            out.append((f"{p}.conv1.bias", ("b", c)))
            out.append((f"{p}.bn1", ("bn", c)))
            out.append((f"{p}.conv2.weight", ("w", c, c, (3, 3, 3))))
            out.append((f"{p}.conv2.bias", ("b", c)))
            out.append((f"{p}.bn2", ("bn", c)))
    return out


def middle_layers(cin=5, with_last=True):
    L = [("subm", cin, 16, (3, 3, 3)), ("bn", 16), ("relu",), ("block", 16), ("block", 16),
         ("conv", 16, 32, (3, 3, 3)), ("bn", 32), ("relu",), ("block", 32), ("block", 32),
         ("conv", 32, 64, (3, 3, 3)), ("bn", 64), ("relu",), ("block", 64), ("block", 64),
         ("conv", 64, 128, (3, 3, 3)), ("bn", 128), ("relu",), ("block", 128), ("block", 128)]
    if with_last:
        L += [("conv", 128, 128, (3, 1, 1)), ("bn", 128), ("relu",)]
    return L


BEV_LAYERS = [("conv", 128, 128, (3, 1, 1)), ("bn", 128), ("relu",)]
RV_LAYERS = [("conv", 128, 32, (1, 1, 3)), ("bn", 32), ("relu",), ("block", 32),
             ("conv", 32, 32, (1, 1, 3)), ("bn", 32), ("relu",), ("block", 32),
             ("conv", 32, 32, (1, 1, 3)), ("bn", 32), ("relu",)]


def backbone_state_dict(kind="SpMiddleResNetFHD", seed=1234, num_input_features=5):
    """Ordered {name: float32 ndarray}. Conv weights KRSC [Cout,kz,ky,kx,Cin] ~ N(0, 1/sqrt(Kvol*Cin));
    conv bias ~ N(0, 0.05); BN gamma~U(.5,1.5), beta~N(0,.1), mean~N(0,.1), var~U(.5,1.5)."""
    if kind == "SpMiddleResNetFHD":
        ents = _seq_entries("middle_conv", middle_layers(num_input_features, True))
    elif kind == "MultiViewBackbone":
        ents = (_seq_entries("middle_conv", middle_layers(num_input_features, False))
                + _seq_entries("bev_conv", BEV_LAYERS) + _seq_entries("rv_conv", RV_LAYERS))
    else:
        raise ValueError(kind)
    rng = np.random.default_rng(seed)
    sd = {}
    for name, spec in ents:
        if spec[0] == "w":
            _, cin, cout, ks = spec
            std = 1.0 / np.sqrt(ks[0] * ks[1] * ks[2] * cin)
            sd[name] = rng.normal(0.0, std, (cout, *ks, cin)).astype(np.float32)
        elif spec[0] == "b":
            sd[name] = rng.normal(0.0, 0.05, (spec[1],)).astype(np.float32)
        else:
            c = spec[1]
            sd[name + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            sd[name + ".bias"] = rng.normal(0.0, 0.1, c).astype(np.float32)
            sd[name + ".running_mean"] = rng.normal(0.0, 0.1, c).astype(np.float32)
            sd[name + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    return sd
