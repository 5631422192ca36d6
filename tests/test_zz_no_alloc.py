"""The allocation-free voxelizer entry point (vx_voxelize_hard_static, include/vx_abi.h) against the workspace form
(vx_voxelize_hard) it shares kernels with: bit-exact outputs on two calls in a row, the second on a smaller cloud
(the static scratch keeps the first call's contents). Called straight through the C-ABI."""
import ctypes

import numpy as np
import pytest
import torch

from vista_amd import _lib as L
from vista_amd.synth import GEOMETRIES, lidar_cloud

pytestmark = pytest.mark.gpu


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _run(points, static, max_points, max_voxels, vs, rg, batch_idx=0, host_count=True):
    lib = L.lib()
    n, f = points.shape
    dev = points.device
    cap = max(1, min(n, max_voxels))
    coords = torch.full((cap, 4), -7, dtype=torch.int32, device=dev)
    nump = torch.full((cap,), -7, dtype=torch.int32, device=dev)
    voxels = torch.full((cap, max_points, f), -7.0, dtype=torch.float32, device=dev)
    mean = torch.full((cap, f), -7.0, dtype=torch.float32, device=dev)
    d_m = torch.zeros((1,), dtype=torch.int64, device=dev)
    h_m = ctypes.c_int64(-1)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    common = (_ptr(points), n, f, vs.ctypes.data_as(L.c_f32p), rg.ctypes.data_as(L.c_f32p), max_points, max_voxels,
              batch_idx, _ptr(coords), _ptr(nump), _ptr(voxels), _ptr(mean), cap, _ptr(d_m),
              ctypes.byref(h_m) if host_count else None)
    if static:
        rc = lib.vx_voxelize_hard_static(*common, stream)
        L.check(rc, "vx_voxelize_hard_static")
    else:
        wsb = lib.vx_voxelize_workspace_bytes(n)
        ws = torch.empty((max(wsb, 256),), dtype=torch.uint8, device=dev)
        rc = lib.vx_voxelize_hard(*common, _ptr(ws), wsb, stream)
        L.check(rc, "vx_voxelize_hard")
    torch.cuda.synchronize()
    m = int(d_m.item())
    if host_count:
        assert h_m.value == m
    return m, coords[:m].cpu(), nump[:m].cpu(), voxels[:m].cpu(), mean[:m].cpu()


def test_static_scratch_entry_point_matches_workspace_entry_point_twice():
    dev = torch.device("cuda:0")
    lib = L.lib()
    assert lib.vx_voxelize_static_init() == 0
    assert lib.vx_voxelize_static_max_points() == 262144
    rg, vs, T, mv = GEOMETRIES["FHD"]
    vs, rg = np.asarray(vs, dtype=np.float32), np.asarray(rg, dtype=np.float32)
    big = lidar_cloud(seed=3, sweeps=10)                       # ~247 k points: close to the limit
    small = lidar_cloud(seed=5, sweeps=2)[:30011]              # second call: a smaller, different cloud
    assert big.shape[0] <= 262144 and small.shape[0] < big.shape[0]
    for cloud, host_count, bidx in ((big, True, 0), (small, False, 3)):
        pts = torch.from_numpy(cloud).to(dev)
        ref = _run(pts, False, T, mv, vs, rg, bidx)
        got = _run(pts, True, T, mv, vs, rg, bidx, host_count=host_count)
        assert got[0] == ref[0] and got[0] > 0
        for a, b, name in zip(got[1:], ref[1:], ("coords", "num_points", "voxels", "mean")):
            assert a.dtype == b.dtype and a.shape == b.shape, name
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{name} differs bitwise"


def test_static_scratch_entry_point_rejects_oversized_clouds_and_truncates_like_the_workspace_form():
    dev = torch.device("cuda:0")
    lib = L.lib()
    rg, vs, T, mv = GEOMETRIES["FHD"]
    vs, rg = np.asarray(vs, dtype=np.float32), np.asarray(rg, dtype=np.float32)
    too_big = torch.zeros((262145, 5), dtype=torch.float32, device=dev)
    with pytest.raises(L.VxError, match="static scratch limit"):
        _run(too_big, True, T, mv, vs, rg)
    # max_voxels truncation (point_cloud_ops.py:196-209) through the static path
    pts = torch.from_numpy(lidar_cloud(seed=9, sweeps=1)).to(dev)
    ref = _run(pts, False, 3, 1000, vs, rg)
    got = _run(pts, True, 3, 1000, vs, rg)
    assert got[0] == ref[0] == 1000
    for a, b in zip(got[1:], ref[1:]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # empty cloud
    empty = torch.zeros((0, 5), dtype=torch.float32, device=dev)
    assert _run(empty, True, T, mv, vs, rg)[0] == 0
