"""GPU mirror of the reference's voxelization front-end (same names, argument meaning and outputs).

  VoxelGenerator   det3d/core/input/voxel_generator.py:6-55
  Voxelization     det3d/datasets/pipelines/preprocess.py:155-208 (dict contract only)
  collate_voxels   det3d/torchie/parallel/collate.py:105-106,135-142 (batch assembly of the voxel keys)

Inputs are CUDA float32 tensors [N, F]; outputs are CUDA tensors. In the reference this stage runs
in DataLoader workers on the CPU; here it is a handful of HBM-bound kernels on the GPU.
"""
import numpy as np
import torch

from . import ops
from .registry import PIPELINES


class VoxelGenerator:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, faster=True):
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        self._grid_size = np.round(grid_size).astype(np.int64)
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = max_num_points
        self._max_voxels = max_voxels
        self.faster = faster  # both reference paths give identical results; one GPU path serves both

    def generate(self, points, max_voxels=None):
        """-> (voxels f32[M,T,F], coordinates int32[M,3] (z,y,x), num_points_per_voxel int32[M])."""
        r = ops.voxelize(points, self._voxel_size, self._point_cloud_range, self._max_num_points,
                         self._max_voxels, with_voxels=True, with_mean=False)
        return r["voxels"], r["coords"][:, 1:], r["num_points"]

    def generate_many(self, clouds):
        """generate() for several clouds in one launch sequence -> list of (voxels, coordinates, num_points)."""
        r = ops.voxelize_batch(list(clouds), self._voxel_size, self._point_cloud_range, self._max_num_points,
                               self._max_voxels, with_voxels=True, with_mean=False)
        out, lo = [], 0
        for m in r["frame_voxels"]:
            out.append((r["voxels"][lo:lo + m], r["coords"][lo:lo + m, 1:], r["num_points"][lo:lo + m]))
            lo += m
        return out

    def generate_mean(self, points, batch_idx=0):
        """Fused voxelization + mean VFE: -> (mean f32[M,F], coordinates int32[M,4] (b,z,y,x), num_points)."""
        r = ops.voxelize(points, self._voxel_size, self._point_cloud_range, self._max_num_points,
                         self._max_voxels, batch_idx=batch_idx, with_voxels=False, with_mean=True)
        return r["mean"], r["coords"], r["num_points"]

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size


@PIPELINES.register_module
class Voxelization:
    """Pipeline stage: res["lidar"]["points"] -> res["lidar"]["voxels"] dict (preprocess.py:175-208) and,
    with cfg.double_flip outside training, the three flipped clouds' dicts (preprocess.py:210-254).

    The four clouds of the double-flip case go through ONE batched launch sequence; each dict is
    bit-identical to a separate generate() call. Box filtering of the training branch (:182-189) is data
    augmentation and stays with the caller."""

    FLIPS = ("yflip", "xflip", "double_flip")

    def __init__(self, **kwargs):
        cfg = kwargs.get("cfg", None)
        get = cfg.get if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        self.range = get("range")
        self.voxel_size = get("voxel_size")
        self.max_points_in_voxel = get("max_points_in_voxel")
        self.max_voxel_num = get("max_voxel_num")
        self.drop_rate = get("drop_rate", 0)
        self.faster = get("faster", False)
        self.double_flip = get("double_flip", False)
        self.voxel_generator = VoxelGenerator(self.voxel_size, self.range, self.max_points_in_voxel,
                                              self.max_voxel_num, self.faster)

    def _dict(self, voxels, coordinates, num_points, flipped):
        g = self.voxel_generator
        d = dict(voxels=voxels, coordinates=coordinates, num_points=num_points,
                 num_voxels=np.array([voxels.shape[0]], dtype=np.int64), shape=g.grid_size)
        if flipped:
            d.update(range=g.point_cloud_range, size=g.voxel_size)
        return d

    def __call__(self, res, info):
        mode = res.get("mode", "val")
        lidar = res["lidar"]
        if self.double_flip and mode != "train":
            triples = self.voxel_generator.generate_many([lidar["points"]] + [lidar[f"{k}_points"] for k in self.FLIPS])
            lidar["voxels"] = self._dict(*triples[0], flipped=False)
            for k, t in zip(self.FLIPS, triples[1:]):
                lidar[f"{k}_voxels"] = self._dict(*t, flipped=True)
            return res, info
        voxels, coordinates, num_points = self.voxel_generator.generate(lidar["points"])
        if mode == "train" and self.drop_rate > 0:
            keep = torch.rand(voxels.shape[0], device=voxels.device) >= self.drop_rate
            voxels, coordinates, num_points = voxels[keep], coordinates[keep], num_points[keep]
        lidar["voxels"] = self._dict(voxels, coordinates, num_points, flipped=False)
        return res, info


@PIPELINES.register_module
class DoubleFlip:
    """Test-time augmentation stage (test_aug.py:52-77) on the GPU: adds the y-, x- and xy-flipped copies of
    res["lidar"]["points"]; with cfg.double_flip the Voxelization stage then voxelizes all four in one batch
    (one frame's TTA = a batch of 4 through the backbone, in the order formating.py:98 returns them)."""

    def __call__(self, res, info):
        pts = res["lidar"]["points"]
        for key, cols in (("yflip_points", [1]), ("xflip_points", [0]), ("double_flip_points", [0, 1])):
            q = pts.clone()
            q[:, cols] = -q[:, cols]
            res["lidar"][key] = q
        return res, info


def collate_voxels(frames):
    """frames: list of per-frame dicts from Voxelization -> batch dict with coordinates [sum M, 4] (b,z,y,x)."""
    coords = [torch.nn.functional.pad(f["coordinates"], (1, 0), value=i) for i, f in enumerate(frames)]
    return dict(voxels=torch.cat([f["voxels"] for f in frames]),
                num_points=torch.cat([f["num_points"] for f in frames]),
                num_voxels=np.concatenate([f["num_voxels"] for f in frames]),
                coordinates=torch.cat(coords), shape=np.stack([f["shape"] for f in frames]))


def voxelize_batch(generator, clouds):
    """Fused path for a list of CUDA clouds: -> (mean f32[sum M,F], coords int32[sum M,4], num_points)."""
    r = ops.voxelize_batch(list(clouds), generator.voxel_size, generator.point_cloud_range,
                           generator.max_num_points_per_voxel, generator._max_voxels, with_voxels=False,
                           with_mean=True)
    return r["mean"], r["coords"], r["num_points"]
