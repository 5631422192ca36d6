"""Rotated BEV overlap / IoU / NMS kernels through the C-ABI against oracle/iou3d_np.py.
Tolerance, not bit-exact: the kernels clip polygons, the oracle restates the reference's intersection-point
algorithm, and the oracle computes in float64 (parity unpinned by the reference)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def random_boxes(rng, n, spread=12.0):
    c = rng.uniform(-spread, spread, (n, 2))
    wh = rng.uniform(0.6, 5.0, (n, 2))
    ang = rng.uniform(-np.pi, np.pi, (n, 1))
    return np.concatenate([c - wh / 2, c + wh / 2, ang], 1).astype(np.float32)


def test_overlap_and_iou_vs_oracle(dev):
    from oracle import iou3d_np as o
    from vista_amd import iou3d_nms as m
    rng = np.random.default_rng(0)
    a, b = random_boxes(rng, 70), random_boxes(rng, 55)
    a[:5, 4] = 0.0                                   # some axis-aligned boxes
    b[:3] = a[:3]                                    # identical pairs
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    ov = torch.empty((70, 55), device=dev)
    iou = torch.empty((70, 55), device=dev)
    assert m.boxes_overlap_bev_gpu(ta, tb, ov) == 1 and m.boxes_iou_bev_gpu(ta, tb, iou) == 1
    ref_ov, ref_iou = o.pairwise(a, b, o.box_overlap), o.pairwise(a, b, o.iou_bev)
    np.testing.assert_allclose(ov.cpu().numpy(), ref_ov, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(iou.cpu().numpy(), ref_iou, rtol=1e-4, atol=2e-5)
    assert (ref_ov > 0).sum() > 100
    with pytest.raises(Exception):
        m.boxes_iou_bev_gpu(ta.cpu(), tb, iou)


@pytest.mark.parametrize("n,thresh,rotated", [(1, 0.5, True), (63, 0.3, True), (200, 0.1, True), (1000, 0.2, True),
                                              (1000, 0.5, False), (4100, 0.3, False)])
def test_nms_vs_oracle(n, thresh, rotated, dev):
    from oracle import iou3d_np as o
    from vista_amd import iou3d_nms as m
    rng = np.random.default_rng(n)
    boxes = random_boxes(rng, n, spread=6.0 + n ** 0.5)
    t = torch.from_numpy(boxes).to(dev)
    # IoU matrix from the GPU kernel for the rotated case at large n (the oracle's python loops are slow); the greedy
    # scan is still the oracle's, and the matrix itself is checked against the oracle in the test above
    iou = None
    if rotated and n > 300:
        buf = torch.empty((n, n), device=dev)
        m.boxes_iou_bev_gpu(t, t, buf)
        iou = buf.cpu().numpy().astype(np.float64)
    elif not rotated:
        x1, y1, x2, y2 = (boxes[:, i].astype(np.float64) for i in range(4))
        w = np.clip(np.minimum(x2[:, None], x2[None]) - np.maximum(x1[:, None], x1[None]), 0, None)
        h = np.clip(np.minimum(y2[:, None], y2[None]) - np.maximum(y1[:, None], y1[None]), 0, None)
        s = (x2 - x1) * (y2 - y1)
        iou = w * h / np.maximum(s[:, None] + s[None] - w * h, 1e-8)
    ref = o.nms(boxes, thresh, rotated, iou)
    keep = torch.zeros(n, dtype=torch.int64)          # CPU LongTensor like box_torch_ops.py:569
    num = (m.nms_gpu if rotated else m.nms_normal_gpu)(t, keep, thresh)
    np.testing.assert_array_equal(keep[:num].numpy(), ref)
    assert 0 < num <= n


def test_rotate_nms_pcdet_contract(dev):
    from vista_amd import iou3d_nms as m
    rng = np.random.default_rng(5)
    boxes = torch.from_numpy(random_boxes(rng, 300)).to(dev)
    scores = torch.from_numpy(rng.uniform(size=300).astype(np.float32)).to(dev)
    sel = m.rotate_nms_pcdet(boxes, scores, 0.2, pre_maxsize=200, post_max_size=50)
    assert sel.dtype == torch.int64 and sel.is_cuda and 0 < sel.numel() <= 50
    s = scores[sel]
    assert bool((s[:-1] >= s[1:]).all())                     # best first
    iou = torch.empty((sel.numel(), sel.numel()), device=dev)
    m.boxes_iou_bev_gpu(boxes[sel].contiguous(), boxes[sel].contiguous(), iou)
    iou.fill_diagonal_(0)
    assert float(iou.max()) <= 0.2 + 1e-6                    # survivors do not overlap beyond the threshold
    assert m.rotate_nms_pcdet(boxes[:0], scores[:0], 0.2).numel() == 0
