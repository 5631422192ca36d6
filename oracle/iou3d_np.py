"""TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the reference's rotated BEV overlap / IoU / NMS,
following ITS algorithm step by step (not the clipping formulation of vista_amd/csrc/vx_nms.hip):
  box_overlap   det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:109-213  (edge intersections :68-97, inside test :51-66,
                angular sort :105-107,187-197, fan area :206-212)
  iou_bev       iou3d_nms_kernel.cu:215-222          iou_normal  :296-304
  nms           mask :251-293 + greedy scan det3d/ops/iou3d_nms/src/iou3d_nms.cpp:106-125
PARITY UNPINNED: the reference holds no fixture for these kernels and building its .cu into oracle/_ref was refused
(DESIGN.md section 2); this file is pinned only by closed-form cases in tests/test_oracle_iou3d.py.
Arithmetic is float64 (the reference computes in float32); comparisons against the GPU use a tolerance.
"""
import math

import numpy as np

EPS = 1e-8
MARGIN = 1e-5


def _cross3(p1, p2, p0):
    return (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1])


def _rect_cross(p1, p2, q1, q2):
    return (min(p1[0], p2[0]) <= max(q1[0], q2[0]) and min(q1[0], q2[0]) <= max(p1[0], p2[0]) and
            min(p1[1], p2[1]) <= max(q1[1], q2[1]) and min(q1[1], q2[1]) <= max(p1[1], p2[1]))


def _intersection(p1, p0, q1, q0):
    if not _rect_cross(p0, p1, q0, q1):
        return None
    s1, s2 = _cross3(q0, p1, p0), _cross3(p1, q1, p0)
    s3, s4 = _cross3(p0, q1, q0), _cross3(q1, p1, q0)
    if not (s1 * s2 > 0 and s3 * s4 > 0):
        return None
    s5 = _cross3(q1, p1, p0)
    if abs(s5 - s1) > EPS:
        return ((s5 * q0[0] - s1 * q1[0]) / (s5 - s1), (s5 * q0[1] - s1 * q1[1]) / (s5 - s1))
    a0, b0, c0 = p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]
    a1, b1, c1 = q0[1] - q1[1], q1[0] - q0[0], q0[0] * q1[1] - q1[0] * q0[1]
    d = a0 * b1 - a1 * b0
    return ((b0 * c1 - b1 * c0) / d, (a1 * c0 - a0 * c1) / d)


def _in_box(box, p):
    cx, cy = (box[0] + box[2]) / 2, (box[1] + box[3]) / 2
    cs, sn = math.cos(-box[4]), math.sin(-box[4])
    rx = (p[0] - cx) * cs + (p[1] - cy) * sn + cx
    ry = -(p[0] - cx) * sn + (p[1] - cy) * cs + cy
    return box[0] - MARGIN < rx < box[2] + MARGIN and box[1] - MARGIN < ry < box[3] + MARGIN


def _corners(box):
    cx, cy = (box[0] + box[2]) / 2, (box[1] + box[3]) / 2
    cs, sn = math.cos(box[4]), math.sin(box[4])
    out = []
    for x, y in ((box[0], box[1]), (box[2], box[1]), (box[2], box[3]), (box[0], box[3])):
        out.append(((x - cx) * cs + (y - cy) * sn + cx, -(x - cx) * sn + (y - cy) * cs + cy))
    return out + [out[0]]


def box_overlap(a, b):
    a, b = [float(v) for v in a], [float(v) for v in b]
    ca, cb = _corners(a), _corners(b)
    pts = []
    for i in range(4):
        for j in range(4):
            p = _intersection(ca[i + 1], ca[i], cb[j + 1], cb[j])
            if p is not None:
                pts.append(p)
    for k in range(4):
        if _in_box(a, cb[k]):
            pts.append(cb[k])
        if _in_box(b, ca[k]):
            pts.append(ca[k])
    if not pts:
        return 0.0
    cx, cy = sum(p[0] for p in pts) / len(pts), sum(p[1] for p in pts) / len(pts)
    pts.sort(key=lambda p: math.atan2(p[1] - cy, p[0] - cx))
    area = 0.0
    for k in range(len(pts) - 1):
        area += ((pts[k][0] - pts[0][0]) * (pts[k + 1][1] - pts[0][1]) -
                 (pts[k][1] - pts[0][1]) * (pts[k + 1][0] - pts[0][0]))
    return abs(area) / 2.0


def iou_bev(a, b):
    sa, sb = (a[2] - a[0]) * (a[3] - a[1]), (b[2] - b[0]) * (b[3] - b[1])
    so = box_overlap(a, b)
    return so / max(sa + sb - so, EPS)


def iou_normal(a, b):
    w = max(min(a[2], b[2]) - max(a[0], b[0]), 0.0)
    h = max(min(a[3], b[3]) - max(a[1], b[1]), 0.0)
    so = w * h
    sa, sb = (a[2] - a[0]) * (a[3] - a[1]), (b[2] - b[0]) * (b[3] - b[1])
    return so / max(sa + sb - so, EPS)


def pairwise(boxes_a, boxes_b, fn):
    out = np.zeros((len(boxes_a), len(boxes_b)), dtype=np.float64)
    for i, a in enumerate(boxes_a):
        for j, b in enumerate(boxes_b):
            out[i, j] = fn(a, b)
    return out


def nms(boxes, thresh, rotated=True, iou=None):
    """boxes sorted by descending score -> kept indices (greedy, iou3d_nms.cpp:111-122). `iou` = precomputed matrix."""
    n = len(boxes)
    if iou is None:
        iou = pairwise(boxes, boxes, iou_bev if rotated else iou_normal)
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        removed[i + 1:] |= iou[i, i + 1:] > thresh
    return np.asarray(keep, dtype=np.int64)
