"""Dense tail of the full VISTA net (vista_amd/dense_tail.py) on CPU: the cross-view attention
neck against outputs of the reference's own attention.py (tests/golden/attention_*.npz, made by
tools/gen_golden_attention.py), necks and head by structure (state_dict keys, shapes, functional re-evaluation)."""
import ast
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vista_amd import dense_tail as dt

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "attention_*.npz"))))
def test_cross_attention_matches_reference_outputs(path):
    d = np.load(path)
    cfg = ast.literal_eval(str(d["cfg"]))
    m = dt.Cross_Attention_Decouple(cfg["embed_dim"], cfg["embed_dim"], cfg["embed_dim"], cfg["num_heads"],
                                    cfg["bev_size"], cfg["bev_block_res"], cfg["rv_size"], cfg["rv_block_res"],
                                    cfg["hidden_channels"]).eval()
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd/")}
    m.load_state_dict(sd, strict=True)          # same module / parameter names as the reference
    bev, rv = torch.from_numpy(d["bev"]), torch.from_numpy(d["rv"])
    bev0, rv0 = bev.clone(), rv.clone()
    with torch.no_grad():
        outs, maps = m((bev, rv))
    for got, key in zip(outs + maps, ("out_sem", "out_geo", "map_sem", "map_geo")):
        np.testing.assert_allclose(got.numpy(), d[key], rtol=1e-5, atol=1e-5)
    assert torch.equal(bev, bev0) and torch.equal(rv, rv0)      # inputs untouched
    with torch.no_grad():
        outs2, maps2 = m((bev, rv), need_maps=False)
    assert maps2 == [None, None] and torch.equal(outs2[0], outs[0])
    with pytest.raises(AssertionError):
        m((bev[..., :-1], rv))


def test_rpnt_structure_and_shapes():
    torch.manual_seed(0)
    cfg = dict(layer_nums=[1, 2], ds_layer_strides=[1, 2], ds_num_filters=[8, 16], us_layer_strides=[1, 2],
               us_num_filters=[12, 6], num_input_features=10)
    m = dt.RPNT(**cfg).eval()
    keys = set(m.state_dict())
    # rpn.py:121-139: ZeroPad, Conv, BN, ReLU, then (Conv, BN, ReLU) x n appended to the same Sequential
    assert {"blocks.0.1.weight", "blocks.0.2.running_mean", "blocks.0.4.weight", "blocks.1.7.weight",
            "deblocks.0.0.weight", "deblocks.1.1.weight"} <= keys
    assert isinstance(m.deblocks[0][0], torch.nn.ConvTranspose2d) and m.deblocks[0][0].kernel_size == (1, 1)
    assert isinstance(dt.RPN(**cfg).deblocks[0][0], torch.nn.Conv2d)     # rpn.py:94 vs rpnt.py:61
    assert m.blocks[0][2].eps == 1e-3 and m.blocks[0][2].momentum == 0.01  # rpn.py:41 default norm
    x = torch.randn(2, 10, 16, 12)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (2, 18, 16, 12) and m.downsample_factor == 1
    # functional re-evaluation of scale 0 from the state_dict
    sd = m.state_dict()
    def bn(t, p):
        return F.batch_norm(t, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            False, 0.0, 1e-3)
    t = F.relu(bn(F.conv2d(F.pad(x, (1, 1, 1, 1)), sd["blocks.0.1.weight"]), "blocks.0.2"))
    t = F.relu(bn(F.conv2d(t, sd["blocks.0.4.weight"], padding=1), "blocks.0.5"))
    up = F.relu(bn(F.conv_transpose2d(t, sd["deblocks.0.0.weight"]), "deblocks.0.1"))
    np.testing.assert_allclose(y[:, :12].numpy(), up.numpy(), rtol=1e-5, atol=1e-5)
    with pytest.raises(AssertionError):
        dt.RPNT(**{**cfg, "us_layer_strides": [1, 4]})


def test_head_forward_layout_and_soft_argmin():
    torch.manual_seed(1)
    head = dt.DeepMultiGroupOHSHeadClear_Decouple(mode="bev", in_channels=24, tasks=dt.VISTA_TASKS).eval()
    assert head.box_coder.n_dim == 10 and len(head.tasks) == 6
    assert {"sem_conv.0.weight", "geo_conv.1.running_var", "tasks.0.conv_cls.3.bias", "tasks.5.conv_velo.0.weight",
            "tasks.2.conv_xy.3.weight", "tasks.1.conv_r.3.weight"} <= set(head.state_dict())
    assert not any("bins" in k for k in head.state_dict())       # bin centres are not parameters in the reference
    xs = (torch.randn(2, 24, 10, 8), torch.randn(2, 24, 10, 8))
    with torch.no_grad():
        preds = head(xs)
    for p, t in zip(preds, dt.VISTA_TASKS):
        assert p["cls_preds"].shape == (2, 10, 8, t["num_class"]) and p["box_preds"].shape == (2, 10, 8, 10)
        b = p["box_preds"]
        assert float(b[..., 0:2].abs().max()) <= 4.5 and -5 <= float(b[..., 2].min()) and float(b[..., 2].max()) <= 3
        assert float(b[..., 3:6].abs().max()) <= 3                       # soft-argmin stays inside the bin range
        np.testing.assert_allclose((b[..., 8:10] ** 2).sum(-1).numpy(), 1.0, rtol=1e-5)   # rotation = unit vector
    # soft-argmin of task 0's z branch, re-evaluated by hand (deep_decouple...:144-147)
    with torch.no_grad():
        g = head.geo_conv(xs[1])
        z = head.tasks[0].conv_z(g)
        ref = (torch.softmax(z, 1) * torch.linspace(-5, 3, 12).view(1, 12, 1, 1)).sum(1)
    np.testing.assert_allclose(preds[0]["box_preds"][..., 2].numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


def test_detector_registry_names():
    assert {"RPN", "RPNT", "Cross_Attention_Decouple"} <= set(dt.NECKS.module_dict)
    assert "DeepMultiGroupOHSHeadClear_Decouple" in dt.HEADS and "OHS_Multiview" in dt.DETECTORS
    det = dt.OHS_Multiview()
    names = {n for n, _ in det.named_children()}
    assert names == {"reader", "backbone", "neck", "rv_neck", "ca_neck", "bbox_head"}       # ohs.py:50-51 names
    assert det.ca_neck.bev_block_res == (40, 40) and det.ca_neck.rv_block_res == (40, 12)
    with pytest.raises(NotImplementedError):
        det({}, return_loss=True)


def test_fuse_for_inference_keeps_outputs():
    torch.manual_seed(3)
    neck = dt.RPNT(layer_nums=[1, 1], ds_layer_strides=[1, 2], ds_num_filters=[8, 16], us_layer_strides=[1, 2],
                   us_num_filters=[12, 6], num_input_features=10).eval()
    head = dt.DeepMultiGroupOHSHeadClear_Decouple(mode="bev", in_channels=18, tasks=dt.VISTA_TASKS[:2]).eval()
    with torch.no_grad():
        for m in list(neck.modules()) + list(head.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
        x = torch.randn(2, 10, 16, 12)
        y0 = neck(x)
        p0 = head((y0, y0))
        dt.fuse_for_inference(neck)
        dt.fuse_for_inference(head)
        assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in list(neck.modules()) + list(head.modules()))
        y1 = neck(x)
        p1 = head((y1, y1))
    np.testing.assert_allclose(y1.numpy(), y0.numpy(), rtol=1e-4, atol=1e-5)
    for a, b in zip(p0, p1):
        np.testing.assert_allclose(b["box_preds"].numpy(), a["box_preds"].numpy(), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):
        dt.fuse_for_inference(dt.RPNT(layer_nums=[1], ds_layer_strides=[1], ds_num_filters=[4], us_layer_strides=[1],
                                      us_num_filters=[4], num_input_features=4))       # still in training mode
