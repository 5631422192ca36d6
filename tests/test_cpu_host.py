"""CPU: host logic + the C-ABI library loads and exports every symbol include/vx_abi.h declares
(no compute calls: there is no GPU here and no CPU fallback in the product path)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(REPO, "include", "vx_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vista_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libvx.so not built (run __graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"libvx.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert _lib.lib().vx_version() >= 1


def test_ops_fail_loudly_without_gpu_tensors():
    from vista_amd import ops
    from vista_amd._lib import VxError
    with pytest.raises(VxError):
        ops.voxelize(torch.zeros(4, 5), [1, 1, 1], [0, 0, 0, 1, 1, 1], 3, 10)
    with pytest.raises(VxError):
        ops.rulebook_subm(torch.zeros((3, 4), dtype=torch.int32), 1, [4, 4, 4], 3)


def test_missing_library_is_an_error(monkeypatch):
    from vista_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libvx.so")
    with pytest.raises(_lib.VxError):
        _lib.lib()


def test_spconv_surface_matches_reference_imports():
    import spconv
    import spconv.pytorch as sp
    from spconv.pytorch import SparseConv3d, SubMConv3d, ops  # noqa: F401  (scn.py:4-6)
    assert spconv.SparseConvTensor is sp.SparseConvTensor     # voxelnet.py:244
    feats = torch.zeros(3, 5)
    idx = torch.zeros(3, 4, dtype=torch.int32)
    t = sp.SparseConvTensor(feats, idx, np.array([41, 1440, 1440]), 2)
    assert t.spatial_shape == [41, 1440, 1440] and t.batch_size == 2 and t.indice_dict == {}
    assert "replace_feature" in t.__dir__()                    # scn.py:21
    with pytest.raises(ValueError):
        t.features = feats
    t2 = t.replace_feature(feats + 1)
    assert t2.indices is t.indices and t2.indice_dict is t.indice_dict and float(t2.features.sum()) == 15
    with pytest.raises(ValueError):
        sp.SparseConvTensor(feats, idx.long(), [4, 4, 4], 1)
    with pytest.raises(ValueError):
        sp.SparseConvTensor(feats, idx[:2], [4, 4, 4], 1)


def test_conv_module_parameters_and_geometry():
    import spconv.pytorch as sp
    c = sp.SparseConv3d(64, 128, 3, 2, padding=[0, 1, 1], bias=False)
    assert tuple(c.weight.shape) == (128, 3, 3, 3, 64) and c.bias is None
    assert c.kernel_size == [3, 3, 3] and c.stride == [2, 2, 2] and c.padding == [0, 1, 1]
    c = sp.SparseConv3d(128, 32, (1, 1, 3), (1, 1, 2), (0, 0, 1), bias=False)  # positional padding, multiview.py:70
    assert c.padding == [0, 0, 1] and tuple(c.weight.shape) == (32, 1, 1, 3, 128)
    s = sp.SubMConv3d(16, 16, 3, stride=1, padding=1, bias=True, indice_key="res0")
    assert s.subm and s.padding == [1, 1, 1] and tuple(s.bias.shape) == (16,)
    s1 = sp.SubMConv3d(16, 16, kernel_size=1, padding=1)       # scn.py:41-51 passes padding=1 with k=1
    assert s1.padding == [0, 0, 0]
    with pytest.raises(TypeError):
        s(torch.zeros(3, 16))
    # a wrong checkpoint weight layout must be loud, not silently skipped
    with pytest.raises(RuntimeError):
        s.load_state_dict({"weight": torch.zeros(3, 3, 3, 16, 16), "bias": torch.zeros(16)})
    from spconv.pytorch.conv import convert_weight
    w = torch.arange(3 * 3 * 3 * 4 * 5, dtype=torch.float32).view(3, 3, 3, 4, 5)  # RSCK (spconv 1.x)
    k = convert_weight(w, "RSCK")
    assert tuple(k.shape) == (5, 3, 3, 3, 4) and float(k[2, 1, 0, 2, 3]) == float(w[1, 0, 2, 3, 2])


def test_sparse_sequential_semantics_and_state_dict_keys():
    import spconv.pytorch as sp
    from vista_amd import models
    from vista_amd.synth import backbone_state_dict

    class Tag(sp.SparseModule):
        def forward(self, x):
            return x.replace_feature(x.features + 100)

    seq = sp.SparseSequential(Tag(), torch.nn.ReLU(), torch.nn.BatchNorm1d(5))
    assert list(seq.state_dict())[:2] == ["2.weight", "2.bias"] and len(seq) == 3 and isinstance(seq[1], torch.nn.ReLU)
    seq.eval()
    x = sp.SparseConvTensor(-torch.ones(4, 5), torch.zeros(4, 4, dtype=torch.int32), [4, 4, 4], 1)
    y = seq(x)
    assert isinstance(y, sp.SparseConvTensor) and torch.allclose(y.features, torch.full((4, 5), 99.0), atol=1e-3)
    e = sp.SparseConvTensor(torch.zeros(0, 5), torch.zeros(0, 4, dtype=torch.int32), [4, 4, 4], 1)
    assert seq(e).features.shape == (0, 5)                      # dense modules are skipped on empty tensors
    for kind in ("SpMiddleResNetFHD", "MultiViewBackbone"):
        net = getattr(models, kind)(num_input_features=5)
        sd = backbone_state_dict(kind)
        keys = {k for k in net.state_dict() if not k.endswith("num_batches_tracked")}
        assert keys == set(sd)
        models.load_numpy_state_dict(net, sd)
    net = models.SpMiddleResNetFHD(num_input_features=5)
    assert sum(p.numel() for p in net.parameters()) == 2695312
    blk = net.middle_conv[3]
    assert blk.conv1.bias is not None and blk.conv2.bias is not None   # scn.py:68-72 quirk
    assert net.middle_conv[0].bias is None and net.middle_conv[0].indice_key == "res0"
    assert net.middle_conv[15].padding == [0, 1, 1] and net.middle_conv[20].kernel_size == [3, 1, 1]


def test_voxel_generator_host_side():
    from vista_amd.synth import GEOMETRIES, grid_size, lidar_cloud
    from vista_amd.voxel import VoxelGenerator
    rg, vs, T, mv = GEOMETRIES["FHD"]
    g = VoxelGenerator(vs, rg, T, mv)
    assert g.grid_size.dtype == np.int64 and list(g.grid_size) == [1440, 1440, 40]
    assert g.voxel_size.dtype == np.float32 and g.point_cloud_range.dtype == np.float32
    assert list(grid_size(*GEOMETRIES["VISTA"][:2])) == [1280, 1280, 100]
    a, b = lidar_cloud(3, 1), lidar_cloud(3, 1)
    assert a.dtype == np.float32 and a.shape[1] == 5 and np.array_equal(a, b)


def test_mean_vfe_reader_matches_reference_formula():
    from vista_amd.models import VoxelFeatureExtractorV3
    v = torch.arange(2 * 3 * 5, dtype=torch.float32).view(2, 3, 5)
    n = torch.tensor([3, 2])
    out = VoxelFeatureExtractorV3(num_input_features=4)(v, n)
    assert out.shape == (2, 4) and torch.allclose(out[1], v[1, :, :4].sum(0) / 2)


def test_shard_frames():
    from vista_amd.train import shard_frames
    assert shard_frames(10, 1, 4) == [1, 5, 9]
    assert shard_frames(10, 3, 4) == [3, 7, 1]      # padded like DistributedSampler
    got = sorted(i for r in range(8) for i in shard_frames(32, r, 8))
    assert got == list(range(32))


def test_registries_and_build_from_cfg():
    """det3d/utils/registry.py:6-78, models/registry.py:3-4: name -> class registries + dict construction."""
    import vista_amd.models  # noqa: F401  (registers the classes)
    import vista_amd.voxel  # noqa: F401
    from vista_amd.registry import BACKBONES, PIPELINES, READERS, Registry, build_from_cfg
    assert set(BACKBONES.module_dict) >= {"SpMiddleResNetFHD", "MultiViewBackbone"}
    assert "VoxelFeatureExtractorV3" in READERS and "Voxelization" in PIPELINES
    net = build_from_cfg(dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8), BACKBONES)
    assert type(net).__name__ == "SpMiddleResNetFHD"
    rd = build_from_cfg(dict(type="VoxelFeatureExtractorV3"), READERS, default_args=dict(num_input_features=5))
    assert rd.num_input_features == 5
    assert build_from_cfg(dict(type=dict), BACKBONES, dict(a=1)) == {"a": 1}      # a class instead of a name
    with pytest.raises(KeyError):
        build_from_cfg(dict(type="NoSuchBackbone"), BACKBONES)
    with pytest.raises(TypeError):
        build_from_cfg(dict(type=3), BACKBONES)
    with pytest.raises(AssertionError):
        build_from_cfg(dict(kind="x"), BACKBONES)
    r = Registry("t")

    @r.register_module
    class A:
        pass
    assert r.get("A") is A and r.get("B") is None and r.name == "t"
    with pytest.raises(KeyError):
        r.register_module(A)
    with pytest.raises(TypeError):
        r.register_module(lambda: 0)


def test_fuse_for_training_keeps_parameters_and_keys():
    """vista_amd.fused_bn.fuse_for_training (host logic only): the rewritten backbone has the same state_dict keys,
    shares the original parameter and buffer tensors, holds 21 fused norms (9 of them taking the block's residual)
    and no stand-alone ReLU after a fused norm; in eval mode a fused norm is nn.BatchNorm1d + ReLU."""
    import torch
    from vista_amd import models
    from vista_amd.fused_bn import FusedBatchNorm1d, FusedSparseBasicBlock, fuse_for_training
    net = models.SpMiddleResNetFHD(num_input_features=5)
    keys = list(net.state_dict().keys())
    params = {k: v.data_ptr() for k, v in net.state_dict().items()}
    fuse_for_training(net)
    assert list(net.state_dict().keys()) == keys
    assert {k: v.data_ptr() for k, v in net.state_dict().items()} == params
    fused = [m for m in net.modules() if isinstance(m, FusedBatchNorm1d)]
    assert len(fused) == 21 and all(m.relu for m in fused)
    assert sum(isinstance(m, FusedSparseBasicBlock) for m in net.modules()) == 8
    assert not any(type(m) is torch.nn.BatchNorm1d for m in net.modules())
    seq = list(net.middle_conv._modules.values())
    for a, b in zip(seq, seq[1:]):
        if isinstance(a, FusedBatchNorm1d):
            assert isinstance(b, torch.nn.Identity)
    # fusing twice changes nothing
    fuse_for_training(net)
    assert list(net.state_dict().keys()) == keys
    bn = torch.nn.BatchNorm1d(16).eval()
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    f = FusedBatchNorm1d.from_bn(bn, relu=True)
    x, r = torch.randn(10, 16), torch.randn(10, 16)
    assert torch.equal(f(x, residual=r), torch.relu(bn(x) + r))
    with pytest.raises(ValueError):
        FusedBatchNorm1d.from_bn(torch.nn.BatchNorm1d(16, affine=False))


def test_vx_trace_names_every_abi_call_before_it_runs(tmp_path):
    """VX_TRACE=1: one stderr line per C-ABI call, written BEFORE the call (so a fatal failure below Python names it);
    size / version queries are not traced. Runs in a child process (the setting is read when vista_amd._lib loads)."""
    import os
    import subprocess
    import sys
    code = ("from vista_amd import _lib as L\n"
            "lib = L.lib()\n"
            "assert lib.vx_version() >= 1\n"
            "assert lib.vx_voxelize_workspace_bytes(1000) > 0\n"
            "assert lib.vx_spconv_get_f32_products() in (0, 6, 9)\n"
            "assert lib.vx_fill(None, 0, 0, None) == 0\n"        # a traced call that needs no GPU: zero bytes
            "L.trace('marker')\n")
    env = dict(os.environ, VX_TRACE="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stderr.splitlines() if ln.startswith("[vx] ")]
    assert lines == ["[vx] vx_fill", "[vx] marker"]          # queries and settings are not launches
