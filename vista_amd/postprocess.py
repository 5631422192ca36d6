"""Inference post-processing of the OHS head: box decoding, double-flip merging, score filter, rotated NMS.

Forward-only restatement (stock torch ops + the HIP NMS of vista_amd.iou3d_nms) of
  GroundBox3dCoderAF.layout/_decode          det3d/core/bbox/box_coders.py:53-68,90-101
  boxes3d_to_bevboxes_lidar_torch            det3d/core/bbox/box_torch_ops.py:579-593
  MultiGroupOHSHeadClear.predict             det3d/models/bbox_heads/clear_mg_ohs_head.py:569-730
  MultiGroupOHSHeadClear.get_task_detections det3d/models/bbox_heads/clear_mg_ohs_head.py:732-852
Unlike the reference nothing here writes into the head's output tensors (its in-place edits never reach a result).
"""
import torch

from .iou3d_nms import rotate_nms_pcdet


class AttrDict(dict):
    """dict with attribute access, nested (what the reference's ConfigDict offers to predict())."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


# configs/vista/vista.py:107-120
VISTA_TEST_CFG = AttrDict(nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000,
                                   nms_post_max_size=80, nms_iou_threshold=0.2),
                          score_threshold=0.1, post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                          max_per_img=500, double_flip=False)


def grid_centers(box_coder, w, h, device, dtype=torch.float32):
    """Cell centres in sensor coordinates, row-major over (h, w) (box_coders.py:53-68)."""
    pc = box_coder.pc_range
    ww = (torch.arange(w, device=device, dtype=dtype) + 0.5) / w * (pc[3] - pc[0]) + pc[0]
    hh = (torch.arange(h, device=device, dtype=dtype) + 0.5) / h * (pc[4] - pc[1]) + pc[1]
    return torch.stack([ww[None, :].expand(h, w).reshape(-1), hh[:, None].expand(h, w).reshape(-1)], 1)


def decode_boxes(box_coder, shifts, w, h):
    """[B, h*w, code] head regression -> [B, h*w, 7 (+2 velocity)] boxes (x, y, z, dx, dy, dz, [vx, vy], r);
    box_coders.py:90-101. Returns a new tensor."""
    out = shifts.clone()
    g = grid_centers(box_coder, w, h, shifts.device, shifts.dtype)
    out[:, :, 0] += g[:, 0]
    out[:, :, 1] += g[:, 1]
    if box_coder.rotation == "vector":
        out[:, :, -2] = torch.atan2(shifts[:, :, -1], shifts[:, :, -2])
        out = out[:, :, :-1]
    if "log" in box_coder.dim:
        out[:, :, 3:6] = torch.exp(out[:, :, 3:6])
    if "bottom" in box_coder.height:
        out[:, :, 2] += 0.5 * out[:, :, 5]
    return out


def boxes3d_to_bevboxes_lidar_torch(boxes3d):
    cu, cv = boxes3d[:, 0], boxes3d[:, 1]
    hw, hl = boxes3d[:, 3] / 2, boxes3d[:, 4] / 2
    return torch.stack([cu - hw, cv - hl, cu + hw, cv + hl, boxes3d[:, -1]], 1)


def _unflip(t):
    """[b, 4, H, W, C]: undo the spatial flips of the y-, x- and xy-flipped passes (clear_mg_ohs_head.py:593-595)."""
    return torch.stack([t[:, 0], torch.flip(t[:, 1], dims=[1]), torch.flip(t[:, 2], dims=[2]),
                        torch.flip(t[:, 3], dims=[1, 2])], 1)


def merge_double_flip(box_coder, box_preds, cls_preds):
    """Head outputs of the 4 test-time-augmentation passes, frame-major [4b, H, W, *] in the order (plain, y-flip,
    x-flip, xy-flip) -> (scores [b, H*W, ncls], boxes [b, H*W, 9]); clear_mg_ohs_head.py:579-678."""
    n4, H, W, C = box_preds.shape
    if n4 % 4:
        raise ValueError(f"double flip needs 4 passes per frame, got batch {n4}")
    b = n4 // 4
    box = box_preds.reshape(b, 4, H, W, C)
    scores = _unflip(cls_preds.reshape(b, 4, H, W, -1)).sigmoid().mean(1).reshape(b, H * W, -1)
    # heading vector (cos, sin): pass 1 mirrors y (theta -> -theta ... expressed on the vector), pass 2 mirrors x
    sign_cos = box.new_tensor([1.0, -1.0, 1.0, -1.0]).view(1, 4, 1, 1)
    sign_sin = box.new_tensor([1.0, 1.0, -1.0, -1.0]).view(1, 4, 1, 1)
    cos, sin = box[..., -2] * sign_cos, box[..., -1] * sign_sin
    fixed = torch.cat([box[..., :-2], cos[..., None], sin[..., None]], -1)
    reg = decode_boxes(box_coder, fixed.reshape(b * 4, H * W, C), W, H).reshape(b, 4, H, W, -1).clone()
    sy = box.new_tensor([1.0, -1.0, 1.0, -1.0]).view(1, 4, 1, 1)     # y, vy negated in passes 1 and 3
    sx = box.new_tensor([1.0, 1.0, -1.0, -1.0]).view(1, 4, 1, 1)     # x, vx negated in passes 2 and 3
    reg[..., 1] *= sy
    reg[..., 7] *= sy
    reg[..., 0] *= sx
    reg[..., 6] *= sx
    reg = _unflip(reg).mean(1).reshape(b, H * W, -1)
    vec = _unflip(torch.stack([cos, sin], -1)).mean(1).reshape(b, H * W, 2)
    reg[:, :, -1] = torch.atan2(vec[:, :, 1], vec[:, :, 0])
    return scores, reg


def _range_index(box, rng):
    """Rows of box [n, >=3] whose centre lies inside rng = (x0, y0, z0, x1, y1, z1); one host sync."""
    return ((box[:, :3] >= rng[:3]).all(1) & (box[:, :3] <= rng[3:]).all(1)).nonzero().squeeze(1)


def get_task_detections(test_cfg, scores, boxes, meta_list=None, range_filter=True):
    """Per frame: best class per cell, score threshold, rotated NMS (pre/post max sizes), centre range filter.
    scores [b, N, ncls] (already sigmoid), boxes [b, N, 9]. Every data-dependent selection is ONE nonzero() (one host
    sync) followed by index_selects; range_filter=False leaves the (per-box) centre range filter to the caller."""
    test_cfg = AttrDict(test_cfg)
    rng = test_cfg.post_center_limit_range
    rng = boxes.new_tensor(rng).float() if range_filter and rng is not None and len(rng) > 0 else None
    meta_list = meta_list or [None] * boxes.shape[0]
    out = []
    for box, sc, meta in zip(boxes, scores, meta_list):
        box, sc = box.float(), sc.float()
        if sc.shape[-1] == 1:
            top, labels = sc.squeeze(-1), torch.zeros(sc.shape[0], device=sc.device, dtype=torch.long)
        else:
            top, labels = torch.max(sc, dim=-1)
        if test_cfg.score_threshold > 0.0:
            keep = (top >= test_cfg.score_threshold).nonzero().squeeze(1)
            top, box, labels = top[keep], box[keep], labels[keep]
        if top.shape[0] != 0:
            sel = rotate_nms_pcdet(boxes3d_to_bevboxes_lidar_torch(box), top, thresh=test_cfg.nms.nms_iou_threshold,
                                   pre_maxsize=test_cfg.nms.nms_pre_max_size,
                                   post_max_size=test_cfg.nms.nms_post_max_size)
            box, top, labels = box[sel], top[sel], labels[sel]
        if box.shape[0] != 0 and rng is not None:
            m = _range_index(box, rng)
            box, top, labels = box[m], top[m], labels[m]
        out.append({"box3d_lidar": box, "scores": top, "label_preds": labels, "metadata": meta})
    return out


def predict(preds_dicts, test_cfg, num_classes, box_coder, metadata=None):
    """List of per-task {"box_preds" [B,H,W,code], "cls_preds" [B,H,W,ncls]} -> one dict per frame with
    box3d_lidar [n, 9], scores [n], label_preds [n] (task label + number of classes of the earlier tasks)."""
    test_cfg = AttrDict(test_cfg)
    double_flip = bool(test_cfg.get("double_flip", False))
    rets = []
    for preds in preds_dicts:
        box, cls = preds["box_preds"], preds["cls_preds"]
        B, H, W, C = box.shape
        if double_flip:
            scores, reg = merge_double_flip(box_coder, box, cls)
            meta = metadata[:4 * (B // 4):4] if metadata else None
        else:
            scores = cls.sigmoid().reshape(B, H * W, -1)
            reg = decode_boxes(box_coder, box.reshape(B, H * W, C), W, H)
            meta = metadata
        # the centre range filter acts on each box alone: it is applied once per frame below, on the concatenated
        # tasks, instead of once per (task, frame) -- same detections, a sixth of the host syncs
        rets.append(get_task_detections(test_cfg, scores, reg, meta, range_filter=False))
    rng = test_cfg.post_center_limit_range
    out = []
    for i in range(len(rets[0])):
        flag, labels = 0, []
        for j, n in enumerate(num_classes):
            labels.append(rets[j][i]["label_preds"] + flag)
            flag += n
        box = torch.cat([r[i]["box3d_lidar"] for r in rets])
        top = torch.cat([r[i]["scores"] for r in rets])
        labels = torch.cat(labels)
        if box.shape[0] != 0 and rng is not None and len(rng) > 0:
            m = _range_index(box, box.new_tensor(rng))
            box, top, labels = box[m], top[m], labels[m]
        out.append({"box3d_lidar": box, "scores": top, "label_preds": labels, "metadata": rets[0][i]["metadata"]})
    return out
