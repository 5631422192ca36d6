"""`spconv.utils` as imported by det3d/core/bbox/box_np_ops.py:9 (data augmentation / evaluation
only, outside the backbone path)."""


def rbbox_iou(*args, **kwargs):
    raise NotImplementedError("spconv.utils.rbbox_iou is outside the sparse-backbone path")


def rbbox_intersection(*args, **kwargs):
    raise NotImplementedError("spconv.utils.rbbox_intersection is outside the sparse-backbone path")
