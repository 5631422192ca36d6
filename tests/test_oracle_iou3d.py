"""oracle/iou3d_np.py (restatement of the reference's rotated overlap / IoU / NMS) against closed-form cases.
The reference holds no fixture for these kernels: parity unpinned."""
import math

import numpy as np

from oracle import iou3d_np as o


def test_axis_aligned_and_rotated_closed_forms():
    a = [0, 0, 2, 2, 0.0]
    assert abs(o.box_overlap(a, [1, 1, 3, 3, 0.0]) - 1.0) < 1e-9
    assert abs(o.iou_bev(a, [1, 1, 3, 3, 0.0]) - 1.0 / 7.0) < 1e-9
    assert o.box_overlap(a, [5, 5, 6, 6, 0.3]) == 0.0
    assert abs(o.box_overlap(a, a) - 4.0) < 1e-6 and abs(o.iou_bev(a, a) - 1.0) < 1e-6
    # a unit-diagonal square turned by 45 degrees inside a big box: its own area
    s = math.sqrt(2.0)
    assert abs(o.box_overlap([-5, -5, 5, 5, 0.0], [-s / 2, -s / 2, s / 2, s / 2, math.pi / 4]) - 2.0) < 1e-9
    # the same square against the half plane x >= 0 (a box covering it): half of it
    assert abs(o.box_overlap([0, -5, 5, 5, 0.0], [-s / 2, -s / 2, s / 2, s / 2, math.pi / 4]) - 1.0) < 1e-9
    # two equal squares, one turned by 45 degrees about the common centre: regular octagon 8(sqrt2-1)r^2, r = 1
    assert abs(o.box_overlap([-1, -1, 1, 1, 0.0], [-1, -1, 1, 1, math.pi / 4]) - 8 * (s - 1)) < 1e-9
    # rotation by 90 degrees of a 4 x 2 box about its centre against itself: the central 2 x 2 square
    assert abs(o.box_overlap([-2, -1, 2, 1, 0.0], [-2, -1, 2, 1, math.pi / 2]) - 4.0) < 1e-6
    assert abs(o.iou_normal([0, 0, 2, 2, 9.9], [1, 0, 3, 2, -1.0]) - 2.0 / 6.0) < 1e-12


def test_nms_greedy_order():
    boxes = np.array([[0, 0, 2, 2, 0.0], [0.1, 0, 2.1, 2, 0.0], [5, 5, 7, 7, 0.2], [0, 0.2, 2, 2.2, 0.0],
                      [5.1, 5, 7.1, 7, 0.2]], dtype=np.float64)
    np.testing.assert_array_equal(o.nms(boxes, 0.5), [0, 2])
    np.testing.assert_array_equal(o.nms(boxes, 0.95), [0, 1, 2, 3, 4])
    np.testing.assert_array_equal(o.nms(boxes[:0], 0.5), [])
    # suppression is not transitive: 1 is removed by 0, so it cannot remove 2
    chain = np.array([[0, 0, 2, 2, 0.0], [0.8, 0, 2.8, 2, 0.0], [1.6, 0, 3.6, 2, 0.0]])
    np.testing.assert_array_equal(o.nms(chain, 0.4, rotated=False), [0, 2])
