"""Post-processing of the model outputs (reference detr_tf/inference.py:68-95).

`get_model_inference(m_outputs, background_class, bbox_format)` keeps the reference's signature and semantics (batch
element 0 only).  `get_model_inference_batched` is the same for EVERY image of the batch in one launch of the HIP kernel
`detr_hip_postprocess` (csrc/postprocess.hip): softmax score / arg-max label per query, background queries dropped with the
order preserved, boxes converted (xyxy / yxyx clipped to [0, 1] like bbox.py:171-183).  CUDA inputs only -- there is no
eager fallback; tensors on the CPU are moved to the model's device first by the caller.
"""
from ctypes import byref

import torch

from . import _hip as hip

BBOX_FORMATS = {"xy_center": 0, "xyxy": 1, "yxyx": 2}


def _post(logits, boxes, background_class, bbox_format):
    if bbox_format not in BBOX_FORMATS:
        raise NotImplementedError()                                     # inference.py:92-93
    if not (logits.is_cuda and boxes.is_cuda):
        raise RuntimeError("get_model_inference: the outputs must live on the GPU (the post-processing is a HIP kernel; no CPU fallback)")
    logits, boxes = logits.float(), boxes.float()
    if logits.stride(-1) != 1:
        logits = logits.contiguous()
    if boxes.stride(-1) != 1:
        boxes = boxes.contiguous()
    B, Q, C = logits.shape
    dev = logits.device
    out_b = torch.zeros(B, Q, 4, dtype=torch.float32, device=dev)
    out_l = torch.zeros(B, Q, dtype=torch.int64, device=dev)
    out_s = torch.zeros(B, Q, dtype=torch.float32, device=dev)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    d = hip.PostprocessDesc()
    d.B, d.Q, d.C = B, Q, C
    d.logits, d.sL_b, d.sL_q = logits.data_ptr(), logits.stride(0), logits.stride(1)
    d.boxes, d.sB_b, d.sB_q = boxes.data_ptr(), boxes.stride(0), boxes.stride(1)
    d.background_class, d.bbox_format = int(background_class), BBOX_FORMATS[bbox_format]
    d.out_boxes, d.out_labels, d.out_scores, d.counts = out_b.data_ptr(), out_l.data_ptr(), out_s.data_ptr(), counts.data_ptr()
    hip._check(hip.load().detr_hip_postprocess(byref(d), hip._stream()), "detr_hip_postprocess")
    return out_b, out_l, out_s, counts


def get_model_inference_batched(m_outputs: dict, background_class, bbox_format="xy_center"):
    """[(boxes [k_b, 4], labels [k_b] int64, scores [k_b])] for every image b of the batch: one kernel launch and ONE
    host synchronisation (the per-image detection counts)."""
    out_b, out_l, out_s, counts = _post(m_outputs["pred_logits"], m_outputs["pred_boxes"], background_class, bbox_format)
    n = counts.tolist()
    return [(out_b[b, :k], out_l[b, :k], out_s[b, :k]) for b, k in enumerate(n)]


def get_model_inference(m_outputs: dict, background_class, bbox_format="xy_center"):
    """inference.py:68-95: (predicted_bbox, predicted_labels, predicted_scores) of batch element 0."""
    out_b, out_l, out_s, counts = _post(m_outputs["pred_logits"][0:1], m_outputs["pred_boxes"][0:1], background_class, bbox_format)
    k = int(counts[0])
    return out_b[0, :k], out_l[0, :k], out_s[0, :k]
