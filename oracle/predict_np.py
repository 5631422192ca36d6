"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of the reference's detection post-processing, following its
control flow statement by statement:
  layout / _decode         det3d/core/bbox/box_coders.py:53-68, 90-101
  bev boxes                det3d/core/bbox/box_torch_ops.py:579-593
  rotate_nms_pcdet         det3d/core/bbox/box_torch_ops.py:555-576 (NMS itself: oracle/iou3d_np.py)
  predict                  det3d/models/bbox_heads/clear_mg_ohs_head.py:569-730 (both the plain and the double-flip branch)
  get_task_detections      det3d/models/bbox_heads/clear_mg_ohs_head.py:732-852
PARITY UNPINNED: the reference holds no fixture for these functions and its head module cannot be imported here
(det3d.models pulls spconv / numba / torchvision); pinned only by hand-checked cases in tests/test_postprocess.py.
float64 arithmetic.
"""
import numpy as np

from . import iou3d_np


def layout(pc_range, w, h):
    ww, hh = np.meshgrid(range(w), range(h))
    ww = ww.reshape(-1).astype(np.float64) + 0.5
    hh = hh.reshape(-1).astype(np.float64) + 0.5
    ww = ww / w * (pc_range[3] - pc_range[0]) + pc_range[0]
    hh = hh / h * (pc_range[4] - pc_range[1]) + pc_range[1]
    return np.stack([ww, hh], 1)


def decode(shifts, w, h, pc_range, rotation="vector", dim="log_soft_argmin", height="bottom_soft_argmin"):
    shifts = np.array(shifts, dtype=np.float64)
    g = layout(pc_range, w, h)
    shifts[:, :, 0] += g[:, 0]
    shifts[:, :, 1] += g[:, 1]
    if rotation == "vector":
        shifts[:, :, -2] = np.arctan2(shifts[:, :, -1], shifts[:, :, -2])
        shifts = shifts[:, :, :-1]
    if "log" in dim:
        shifts[:, :, 3:6] = np.exp(shifts[:, :, 3:6])
    if "bottom" in height:
        shifts[:, :, 2] += 0.5 * shifts[:, :, 5]
    return shifts


def bev_boxes(b):
    return np.stack([b[:, 0] - b[:, 3] / 2, b[:, 1] - b[:, 4] / 2, b[:, 0] + b[:, 3] / 2, b[:, 1] + b[:, 4] / 2, b[:, -1]], 1)


def rotate_nms(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    order = np.argsort(-scores, kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = iou3d_np.nms(boxes[order], thresh, rotated=True)
    sel = order[keep]
    if post_max_size is not None:
        sel = sel[:post_max_size]
    return sel


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def task_detections(cfg, cls, reg):
    out = []
    rng = np.asarray(cfg["post_center_limit_range"], dtype=np.float64)
    for box, sc in zip(reg, cls):
        if sc.shape[-1] == 1:
            top, labels = sc[:, 0], np.zeros(sc.shape[0], dtype=np.int64)
        else:
            top, labels = sc.max(-1), sc.argmax(-1)
        if cfg["score_threshold"] > 0.0:
            keep = top >= np.float32(cfg["score_threshold"])
            top, box, labels = top[keep], box[keep], labels[keep]
        if top.shape[0] != 0:
            sel = rotate_nms(bev_boxes(box), top, cfg["nms"]["nms_iou_threshold"], cfg["nms"]["nms_pre_max_size"],
                             cfg["nms"]["nms_post_max_size"])
        else:
            sel = np.zeros(0, dtype=np.int64)
        box, top, labels = box[sel], top[sel], labels[sel]
        if box.shape[0] != 0:
            m = (box[:, :3] >= rng[:3]).all(1) & (box[:, :3] <= rng[3:]).all(1)
            box, top, labels = box[m], top[m], labels[m]
        out.append({"box3d_lidar": box, "scores": top, "label_preds": labels})
    return out


def predict(preds_dicts, cfg, num_classes, pc_range):
    rets = []
    for preds in preds_dicts:
        box_all = np.array(preds["box_preds"], dtype=np.float64)
        cls_all = np.array(preds["cls_preds"], dtype=np.float64)
        if cfg.get("double_flip", False):
            n4, H, W, C = box_all.shape
            b = n4 // 4
            box_all = box_all.reshape(b, 4, H, W, C)
            sincos = box_all[:, :, :, :, 8:10].copy()
            cls_all = cls_all.reshape(b, 4, H, W, -1)
            cls_all[:, 1] = np.flip(cls_all[:, 1], axis=1)
            cls_all[:, 2] = np.flip(cls_all[:, 2], axis=2)
            cls_all[:, 3] = np.flip(cls_all[:, 3], axis=(1, 2))
            cls = sigmoid(cls_all).mean(1).reshape(b, H * W, -1)
            sincos[:, 1] = np.flip(sincos[:, 1], axis=1)
            sincos[:, 2] = np.flip(sincos[:, 2], axis=2)
            sincos[:, 3] = np.flip(sincos[:, 3], axis=(1, 2))
            reg = np.zeros((b, 4, H * W, 9))
            for i in range(4):
                bp = box_all[:, i].reshape(b, -1, C).copy()
                if i == 1:
                    bp[:, :, -2] = -bp[:, :, -2]
                    sincos[:, i, :, :, 0] = -sincos[:, i, :, :, 0]
                elif i == 2:
                    bp[:, :, -1] = -bp[:, :, -1]
                    sincos[:, i, :, :, 1] = -sincos[:, i, :, :, 1]
                elif i == 3:
                    bp[:, :, -1] = -bp[:, :, -1]
                    bp[:, :, -2] = -bp[:, :, -2]
                    sincos[:, i, :, :, 0] = -sincos[:, i, :, :, 0]
                    sincos[:, i, :, :, 1] = -sincos[:, i, :, :, 1]
                t = decode(bp, W, H, pc_range)
                if i == 1:
                    t[:, :, 1] = -t[:, :, 1]
                    t[:, :, 7] = -t[:, :, 7]
                elif i == 2:
                    t[:, :, 0] = -t[:, :, 0]
                    t[:, :, 6] = -t[:, :, 6]
                elif i == 3:
                    t[:, :, 1] = -t[:, :, 1]
                    t[:, :, 0] = -t[:, :, 0]
                    t[:, :, 7] = -t[:, :, 7]
                    t[:, :, 6] = -t[:, :, 6]
                reg[:, i] = t
            sincos = sincos.mean(1).reshape(b, -1, 2)
            rads = np.arctan2(sincos[:, :, 1], sincos[:, :, 0])
            reg = reg.reshape(b, 4, H, W, 9)
            reg[:, 1] = np.flip(reg[:, 1], axis=1)
            reg[:, 2] = np.flip(reg[:, 2], axis=2)
            reg[:, 3] = np.flip(reg[:, 3], axis=(1, 2))
            reg = reg.mean(1).reshape(b, -1, 9)
            reg[:, :, -1] = rads
        else:
            B, H, W, C = box_all.shape
            cls = sigmoid(cls_all).reshape(B, H * W, -1)
            reg = decode(box_all.reshape(B, -1, C), W, H, pc_range)
        rets.append(task_detections(cfg, cls, reg))
    out = []
    for i in range(len(rets[0])):
        flag, labels = 0, []
        for j, n in enumerate(num_classes):
            labels.append(rets[j][i]["label_preds"] + flag)
            flag += n
        out.append({"box3d_lidar": np.concatenate([r[i]["box3d_lidar"] for r in rets]),
                    "scores": np.concatenate([r[i]["scores"] for r in rets]),
                    "label_preds": np.concatenate(labels)})
    return out
