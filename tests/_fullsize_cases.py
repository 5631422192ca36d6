"""Full-size parity cases shared by tests/test_gpu_fullsize.py (per-layer dispatch) and tests/test_x_gpu_plan.py (the
same cases through the queued / captured FramePlan). Not collected by pytest (no test_ prefix)."""
import numpy as np
import torch


def _setup(geometry, seeds, sweeps, dev):
    from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES[geometry]
    gen = VoxelGenerator(vs, rg, T, mv)
    clouds = [lidar_cloud(seed=s, sweeps=sweeps) for s in seeds]
    grid = grid_size(rg, vs).tolist()
    return gen, clouds, grid, (rg, vs, T, mv)


def _engine(net, dispatch, dtype=torch.float32):
    from vista_amd.engine import FusedBackbone
    if dispatch == "eager":
        return FusedBackbone(net, dtype=dtype, eager=True)
    return FusedBackbone(net, dtype=dtype, dispatch=dispatch)


def vista_multiview_case(dev, dispatch):
    """BASELINE configs[2], sparse part, at FULL size: one VISTA frame (grid [101,1280,1280]) through
    MultiViewBackbone -- fused engine and module path -- against the oracle, both outputs (BEV [1,640,160,160] and
    RV [1,640,160,12]) within 1e-4 relative, zeros exactly outside the oracle's active sites."""
    from oracle import backbone_oracle as bo
    from oracle import voxelize_np as ov
    from vista_amd import models
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import backbone_state_dict
    from vista_amd.voxel import voxelize_batch
    gen, clouds, grid, (rg, vs, T, mv) = _setup("VISTA", [0], 10, dev)
    mean, coords, nump = voxelize_batch(gen, [torch.from_numpy(clouds[0]).to(dev)])
    v, c, n = ov.voxelize(clouds[0], vs, rg, T, mv)
    o_coords = ov.collate_coords([c])
    np.testing.assert_array_equal(coords.cpu().numpy(), o_coords)
    np.testing.assert_array_equal(nump.cpu().numpy(), n)
    o_feat = ov.mean_vfe(v, n)
    sd = backbone_state_dict("MultiViewBackbone")
    net = models.load_numpy_state_dict(models.MultiViewBackbone(num_input_features=5), sd).to(dev).eval()
    bev_o, rv_o, tb, tr = bo.multi_view_backbone(o_feat, o_coords, 1, grid, sd, return_tensors=True)
    masks = [bo.active_mask(tb), bo.active_mask(tr, "rv")]
    with torch.no_grad():
        out_mod = net(mean, coords, 1, grid)
    out_fused = _engine(net, dispatch).forward_frames(gen, [torch.from_numpy(clouds[0]).to(dev)], grid)
    assert bev_o.shape == (1, 640, 160, 160) and rv_o.shape == (1, 640, 160, 12)
    for outs in (out_mod, out_fused):
        for got, ref, act in zip(outs, (bev_o, rv_o), masks):
            got = got.cpu().numpy()
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
            assert not (got[~act] != 0).any()


def stress_case(dev, dispatch):
    """BASELINE configs[3] at FULL size: one 40-sweep cloud (~986 k points) on the 0.05 m grid [41,2160,2160]:
    voxels bit-exact, backbone output within 1e-4 relative of the oracle (fp32) / 2e-2 (bf16)."""
    from oracle import backbone_oracle as bo
    from oracle import voxelize_np as ov
    from vista_amd import models
    from vista_amd.engine import FusedBackbone
    from vista_amd.synth import backbone_state_dict
    from vista_amd.voxel import voxelize_batch
    gen, clouds, grid, (rg, vs, T, mv) = _setup("STRESS", [7], 40, dev)
    frame = torch.from_numpy(clouds[0]).to(dev)
    mean, coords, nump = voxelize_batch(gen, [frame])
    v, c, n = ov.voxelize(clouds[0], vs, rg, T, mv)
    o_coords = ov.collate_coords([c])
    assert coords.shape[0] > 500_000
    np.testing.assert_array_equal(coords.cpu().numpy(), o_coords)
    np.testing.assert_array_equal(nump.cpu().numpy(), n)
    o_feat = ov.mean_vfe(v, n)
    np.testing.assert_allclose(mean.cpu().numpy(), o_feat, rtol=1e-6, atol=1e-6)
    sd = backbone_state_dict("SpMiddleResNetFHD")
    net = models.load_numpy_state_dict(models.SpMiddleResNetFHD(num_input_features=5), sd).to(dev).eval()
    ref, t = bo.sp_middle_resnet_fhd(o_feat, o_coords, 1, grid, sd)
    act = bo.active_mask(t)
    scale = np.abs(ref).max()
    out = _engine(net, dispatch).forward_frames(gen, [frame], grid).cpu().numpy()
    assert out.shape == ref.shape == (1, 256, 270, 270)
    assert np.abs(out - ref).max() <= 1e-4 * scale
    assert not (out[~act] != 0).any()
    out16 = _engine(net, dispatch, torch.bfloat16).forward_frames(gen, [frame], grid).float().cpu().numpy()
    assert np.abs(out16 - ref).max() <= 2e-2 * scale
