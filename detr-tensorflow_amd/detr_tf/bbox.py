"""Box helpers of the hot path (torch tensors on any device) -- the TF half of the reference's
detr_tf/bbox.py (:171-196); the pairwise IoU/GIoU arithmetic itself runs inside the HIP set-loss
kernels (csrc/setloss.hip)."""
import torch


def xcycwh_to_xy_min_xy_max(bbox):
    """bbox.py:171-183 -- including the clip to [0, 1]."""
    xyxy = torch.cat([bbox[:, :2] - (bbox[:, 2:] / 2), bbox[:, :2] + (bbox[:, 2:] / 2)], dim=-1)
    return torch.clamp(xyxy, 0.0, 1.0)


def xy_min_xy_max_to_yx_min_yx_max(bbox):
    """bbox.py:126-139."""
    return torch.cat([bbox[:, 1:2], bbox[:, 0:1], bbox[:, 3:4], bbox[:, 2:3]], dim=-1)


def xcycwh_to_yx_min_yx_max(bbox):
    """bbox.py:186-196."""
    return xy_min_xy_max_to_yx_min_yx_max(xcycwh_to_xy_min_xy_max(bbox))
