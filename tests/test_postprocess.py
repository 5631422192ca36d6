"""Detection post-processing (vista_amd/postprocess.py; reference clear_mg_ohs_head.py:569-852, box_coders.py:53-101).
CPU: the numpy oracle against hand-computed cases and the mirror-consistency property of the double-flip merge.
GPU: the torch + HIP-NMS path against the oracle on seeded head outputs. PARITY UNPINNED (no reference fixture)."""
import numpy as np
import pytest
import torch

from oracle import predict_np as op

PC = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
CFG = dict(nms=dict(nms_pre_max_size=1000, nms_post_max_size=80, nms_iou_threshold=0.2), score_threshold=0.1,
           post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def head_outputs(rng, b, h, w, ncls, hot=0.04):
    """Random head maps: a few confident cells, boxes of 1.5-5 m so neighbours overlap."""
    box = np.zeros((b, h, w, 10), dtype=np.float32)
    box[..., 0:2] = rng.uniform(-3, 3, size=(b, h, w, 2))
    box[..., 2] = rng.uniform(-2, 0, size=(b, h, w))
    box[..., 3:6] = np.log(rng.uniform(1.5, 5.0, size=(b, h, w, 3)))
    box[..., 6:8] = rng.normal(size=(b, h, w, 2))
    th = rng.uniform(-np.pi, np.pi, size=(b, h, w))
    box[..., 8], box[..., 9] = np.cos(th), np.sin(th)
    cls = rng.normal(-4.0, 1.0, size=(b, h, w, ncls)).astype(np.float32)
    hotm = rng.uniform(size=(b, h, w, ncls)) < hot
    cls[hotm] = rng.uniform(-1.0, 3.0, size=int(hotm.sum()))
    return box, cls


def mirrored_passes(box, cls):
    """What a mirror-equivariant network would output for (plain, y-flip, x-flip, xy-flip) of one scene, in the head's
    own conventions (clear_mg_ohs_head.py:628-663): frame-major [4b, H, W, *]."""
    def y(bx):
        o = np.flip(bx, 1).copy(); o[..., 1] *= -1; o[..., 7] *= -1; o[..., 8] *= -1; return o
    def x(bx):
        o = np.flip(bx, 2).copy(); o[..., 0] *= -1; o[..., 6] *= -1; o[..., 9] *= -1; return o
    b = box.shape[0]
    bs = np.stack([box, y(box), x(box), x(y(box))], 1).reshape(4 * b, *box.shape[1:])
    cs = np.stack([cls, np.flip(cls, 1), np.flip(cls, 2), np.flip(cls, (1, 2))], 1).reshape(4 * b, *cls.shape[1:])
    return np.ascontiguousarray(bs), np.ascontiguousarray(cs)


def test_oracle_decode_hand_case():
    """2 x 2 map over [-51.2, 51.2]^2: cell centres at +-25.6; one cell checked by hand."""
    s = np.zeros((1, 4, 10))
    s[0, 3] = [1.0, -2.0, -1.0, np.log(2.0), np.log(4.0), np.log(1.5), 0.3, -0.4, 0.0, 1.0]   # cell (h=1, w=1)
    out = op.decode(s, 2, 2, PC)
    assert out.shape == (1, 4, 9)
    np.testing.assert_allclose(out[0, 3], [25.6 + 1.0, 25.6 - 2.0, -1.0 + 0.75, 2.0, 4.0, 1.5, 0.3, -0.4, np.pi / 2],
                               rtol=0, atol=1e-12)
    np.testing.assert_allclose(out[0, 0, :2], [-25.6, -25.6])          # cell (0, 0); (h=0, w=1) is x = +25.6
    np.testing.assert_allclose(out[0, 1, :2], [25.6, -25.6])


def test_oracle_detections_hand_case():
    """Three confident cells: two heavily overlapping boxes (the weaker one is suppressed), one far away and one
    outside the centre range (dropped by the range filter)."""
    h = w = 8
    box = np.zeros((1, h, w, 10)); box[..., 3:6] = np.log(2.0); box[..., 8] = 1.0
    cls = np.full((1, h, w, 1), -10.0)
    cls[0, 2, 2, 0], cls[0, 2, 3, 0], cls[0, 6, 6, 0], cls[0, 0, 0, 0] = 3.0, 2.0, 1.0, 2.5
    box[0, 2, 3, 0] = -12.0          # cell (2,3) lands 0.8 m from cell (2,2): IoU of the 2 x 2 boxes 0.43 > 0.2
    box[0, 0, 0, 2] = 30.0           # z outside [-10, 10] -> range filter
    det = op.predict([{"box_preds": box, "cls_preds": cls}], CFG, [1], PC)[0]
    assert det["scores"].shape == (2,)
    np.testing.assert_allclose(det["scores"], 1 / (1 + np.exp(-np.array([3.0, 1.0]))))
    np.testing.assert_allclose(det["box3d_lidar"][0, :2], [-51.2 + 2.5 * 12.8, -51.2 + 2.5 * 12.8])
    assert (det["label_preds"] == 0).all()


def test_oracle_double_flip_of_mirrored_passes_is_the_plain_result():
    rng = np.random.default_rng(3)
    box, cls = head_outputs(rng, 2, 6, 5, 2)
    b4, c4 = mirrored_passes(box, cls)
    plain = op.predict([{"box_preds": box, "cls_preds": cls}], CFG, [2], PC)
    merged = op.predict([{"box_preds": b4, "cls_preds": c4}], dict(CFG, double_flip=True), [2], PC)
    assert len(merged) == 2
    for p, m in zip(plain, merged):
        assert p["scores"].shape == m["scores"].shape and p["scores"].shape[0] > 0
        np.testing.assert_allclose(m["box3d_lidar"], p["box3d_lidar"], atol=1e-9)
        np.testing.assert_allclose(m["scores"], p["scores"], atol=1e-12)
        assert (m["label_preds"] == p["label_preds"]).all()


def test_label_offsets_over_tasks_cpu_oracle():
    rng = np.random.default_rng(5)
    preds = [dict(zip(("box_preds", "cls_preds"), head_outputs(rng, 1, 8, 8, n, hot=0.1))) for n in (1, 2, 2)]
    det = op.predict(preds, CFG, [1, 2, 2], PC)[0]
    assert det["label_preds"].min() >= 0 and det["label_preds"].max() <= 4
    n0 = op.predict(preds[:1], CFG, [1], PC)[0]["scores"].shape[0]
    assert (det["label_preds"][:n0] == 0).all() and (det["label_preds"][n0:] >= 1).all()


def _to(dev, preds):
    return [{k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in p.items()} for p in preds]


def _compare(got, ref):
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g["scores"].shape[0] == r["scores"].shape[0] and r["scores"].shape[0] > 0
        np.testing.assert_allclose(g["scores"].cpu().numpy(), r["scores"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(g["box3d_lidar"].cpu().numpy(), r["box3d_lidar"], rtol=1e-4, atol=1e-4)
        assert (g["label_preds"].cpu().numpy() == r["label_preds"]).all()
        assert g["label_preds"].dtype == torch.long and g["box3d_lidar"].shape[1] == 9


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_predict_matches_oracle(seed, dev):
    """Decode + score filter + HIP rotated NMS + range filter + label offsets against the oracle (float64, reference
    algorithm for the overlap); 2 frames, 3 task groups."""
    from vista_amd.dense_tail import BoxCoderSpec
    from vista_amd.postprocess import predict
    rng = np.random.default_rng(seed)
    preds = [dict(zip(("box_preds", "cls_preds"), head_outputs(rng, 2, 24, 24, n))) for n in (1, 2, 2)]
    ref = op.predict(preds, CFG, [1, 2, 2], PC)
    got = predict(_to(dev, preds), CFG, [1, 2, 2], BoxCoderSpec(pc_range=PC))
    _compare(got, ref)


@pytest.mark.gpu
def test_predict_double_flip_matches_oracle(dev):
    """Independent random maps for the 4 passes (not mirror images): exercises every sign / flip of the merge."""
    from vista_amd.dense_tail import BoxCoderSpec
    from vista_amd.postprocess import predict
    rng = np.random.default_rng(7)
    preds = [dict(zip(("box_preds", "cls_preds"), head_outputs(rng, 8, 20, 16, n, hot=0.3))) for n in (1, 2)]
    cfg = dict(CFG, double_flip=True, score_threshold=0.12)
    ref = op.predict(preds, cfg, [1, 2], PC)
    got = predict(_to(dev, preds), cfg, [1, 2], BoxCoderSpec(pc_range=PC), metadata=[{"token": i} for i in range(8)])
    assert len(got) == 2 and got[0]["metadata"] == {"token": 0} and got[1]["metadata"] == {"token": 4}
    _compare(got, ref)


@pytest.mark.gpu
def test_predict_empty_and_head_method(dev):
    """No cell above the threshold -> empty detections with the right shapes; the head's predict() is the same call."""
    from vista_amd.dense_tail import DeepMultiGroupOHSHeadClear_Decouple, VISTA_TASKS
    from vista_amd.postprocess import VISTA_TEST_CFG
    head = DeepMultiGroupOHSHeadClear_Decouple(mode="bev", in_channels=384, tasks=VISTA_TASKS)
    preds = [{"box_preds": torch.zeros(1, 8, 8, 10, device=dev), "cls_preds": torch.full((1, 8, 8, n), -9.0, device=dev)}
             for n in head.num_classes]
    det = head.predict({"metadata": [None]}, preds, VISTA_TEST_CFG)[0]
    assert det["box3d_lidar"].shape == (0, 9) and det["scores"].shape == (0,) and det["label_preds"].shape == (0,)
