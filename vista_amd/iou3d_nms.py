"""Rotated BEV box overlap / IoU / NMS on the GPU behind the reference's four extension names
(det3d/ops/iou3d_nms/src/iou3d_nms.cpp:180-185) and its caller rotate_nms_pcdet
(det3d/core/bbox/box_torch_ops.py:555-576). Boxes are float32 CUDA tensors [N, 5] = [x1, y1, x2, y2, ry]."""
import ctypes

import torch

from . import _lib as L
from .ops import _need_gpu, _ptr, _stream, _ws


def _check(boxes, name):
    _need_gpu(boxes, name)
    if boxes.dtype != torch.float32 or boxes.dim() != 2 or boxes.shape[1] != 5 or not boxes.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float32 [N, 5] tensor")


def _pairwise(boxes_a, boxes_b, ans, iou):
    _check(boxes_a, "boxes_a")
    _check(boxes_b, "boxes_b")
    _need_gpu(ans, "ans")
    if ans.dtype != torch.float32 or tuple(ans.shape) != (boxes_a.shape[0], boxes_b.shape[0]) or not ans.is_contiguous():
        raise ValueError("ans must be a contiguous float32 [N, M] tensor")
    with torch.cuda.device(boxes_a.device):
        rc = L.lib().vx_boxes_overlap_bev(_ptr(boxes_a), boxes_a.shape[0], _ptr(boxes_b), boxes_b.shape[0], _ptr(ans),
                                          int(iou), _stream())
    L.check(rc, "vx_boxes_overlap_bev")
    return 1


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    return _pairwise(boxes_a, boxes_b, ans_overlap, 0)


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    return _pairwise(boxes_a, boxes_b, ans_iou, 1)


def _nms(boxes, keep, thresh, rotated):
    """boxes sorted by descending score; fills keep[:num] (int64, CPU like the reference's, or CUDA) -> num."""
    _check(boxes, "boxes")
    if keep.dtype != torch.int64 or keep.numel() < boxes.shape[0] or not keep.is_contiguous():
        raise ValueError("keep must be a contiguous int64 tensor with room for every box")
    n = boxes.shape[0]
    d_keep = keep if keep.is_cuda else torch.empty((max(n, 1),), dtype=torch.int64, device=boxes.device)
    lib = L.lib()
    wsb = lib.vx_nms_workspace_bytes(n)
    ws = _ws(wsb, boxes.device)
    h = ctypes.c_int64(0)
    with torch.cuda.device(boxes.device):
        rc = lib.vx_nms_bev(_ptr(boxes), n, float(thresh), int(rotated), _ptr(d_keep), ctypes.byref(h), _ptr(ws), wsb,
                            _stream())
    L.check(rc, "vx_nms_bev")
    num = int(h.value)
    if not keep.is_cuda:
        keep[:num] = d_keep[:num].cpu()
    return num


def nms_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, True)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, False)


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """Same contract as box_torch_ops.py:555-576: -> indices into `boxes`, best first."""
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    sel = boxes[order].contiguous()
    keep = torch.empty((sel.shape[0],), dtype=torch.int64, device=sel.device)
    num = nms_gpu(sel, keep, thresh)
    selected = order[keep[:num]].contiguous()
    return selected[:post_max_size] if post_max_size is not None else selected
