"""Post-processing of the model outputs -- detr_tf/inference.py:68-95 (batch element 0 only)."""
import torch

from . import bbox


def get_model_inference(m_outputs: dict, background_class, bbox_format="xy_center"):
    predicted_bbox = m_outputs["pred_boxes"][0]
    predicted_labels = m_outputs["pred_logits"][0]
    softmax = torch.softmax(predicted_labels, dim=-1)
    predicted_scores, predicted_labels = softmax.max(dim=-1)
    indices = torch.nonzero(predicted_labels != background_class)[:, 0]
    predicted_scores = predicted_scores[indices]
    predicted_labels = predicted_labels[indices]
    predicted_bbox = predicted_bbox[indices]
    if bbox_format == "xy_center":
        predicted_bbox = predicted_bbox
    elif bbox_format == "xyxy":
        predicted_bbox = bbox.xcycwh_to_xy_min_xy_max(predicted_bbox)
    elif bbox_format == "yxyx":
        predicted_bbox = bbox.xcycwh_to_yx_min_yx_max(predicted_bbox)
    else:
        raise NotImplementedError()
    return predicted_bbox, predicted_labels, predicted_scores
