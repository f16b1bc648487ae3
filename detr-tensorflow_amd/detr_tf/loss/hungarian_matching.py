"""Hungarian matching on the device (drop-in for detr_tf/loss/hungarian_matching.py:163-203).

The reference builds the [100, n] cost matrix with TF ops and then leaves the TF runtime through
tf.numpy_function into scipy.optimize.linear_sum_assignment, once per image per decoder level
(48 host round trips per step at B=8).  Here K12 (cost) and K13 (exact assignment) are HIP
kernels in libdetr_hip.so; nothing is synchronised with the host.
"""
from ctypes import byref

import torch

from .. import _hip as hip


def make_desc(logits, boxes, t_bbox, t_class, background_class):
    """logits [Lv,B,Q,C], boxes [Lv,B,Q,4] (any strides with a unit last stride), targets in the
    reference's header layout (detr_tf/data/processing.py:35-55)."""
    Lv, B, Q, C = logits.shape
    assert logits.stride(3) == 1 and boxes.stride(3) == 1
    assert t_bbox.is_contiguous() and t_class.is_contiguous() and t_class.dtype == torch.int64
    d = hip.SetLossDesc()
    d.levels, d.B, d.Q, d.C, d.R = Lv, B, Q, C, t_bbox.shape[1]
    d.logits, d.sL_l, d.sL_b, d.sL_q = logits.data_ptr(), logits.stride(0), logits.stride(1), logits.stride(2)
    d.boxes, d.sB_l, d.sB_b, d.sB_q = boxes.data_ptr(), boxes.stride(0), boxes.stride(1), boxes.stride(2)
    d.t_bbox, d.t_class = t_bbox.data_ptr(), t_class.data_ptr()
    d.background_class = int(background_class)
    return d


before_assign = None     # hook of the training step (training.run_train_step): called between the cost and the assignment launches


class Matcher:
    """Persistent device buffers of the matcher for one (levels, B, Q, R) shape."""

    def __init__(self, levels, B, Q, R, device):
        self.P, self.Q, self.R, self.B = levels * B, Q, R, B
        self.cost = torch.empty(self.P, Q, R - 1, dtype=torch.float32, device=device)
        self.tgt_for_pred = torch.empty(self.P, Q, dtype=torch.int32, device=device)
        self.pred_for_tgt = torch.empty(self.P, R - 1, dtype=torch.int32, device=device)
        self.status = torch.empty(self.P, dtype=torch.int32, device=device)

    def run(self, desc, t_bbox):
        hip.call("detr_hip_match_cost_f32", byref(desc), self.cost.data_ptr())
        if before_assign is not None:       # work of another stream queued behind the cost matrix: the assignment keeps ~50 CUs busy
            before_assign()
        hip.call("detr_hip_assign_f32", self.cost.data_ptr(), self.P, self.Q, self.R - 1, t_bbox.data_ptr(), self.B,
                 self.R, self.tgt_for_pred.data_ptr(), self.pred_for_tgt.data_ptr(), self.status.data_ptr())
        return self.tgt_for_pred


def hungarian_matching(t_bbox, t_class, p_bbox, p_class, fcost_class=1, fcost_bbox=5, fcost_giou=2, slice_preds=True):
    """Single-image API with the reference's signature and return convention
    (pred_indices, target_indices, pred_selector, target_selector, t_bbox, t_class) --
    after the reference's double name swap position 0 holds the TARGET indices,
    position 1 the PREDICTION indices, position 2 the bool[n] selector over targets and position 3
    the bool[Q] selector over predictions.  Runs the same two HIP kernels on a batch of one."""
    assert (fcost_class, fcost_bbox, fcost_giou) == (1, 5, 2), "cost weights are compiled into K12"
    assert slice_preds, "targets must carry the header row"
    dev = p_bbox.device
    Q, C = p_class.shape
    tb = t_bbox.reshape(1, -1, 4).contiguous().float()
    tc = t_class.reshape(1, -1).contiguous().long()
    lg = p_class.reshape(1, 1, Q, C).contiguous().float()
    bx = p_bbox.reshape(1, 1, Q, 4).contiguous().float()
    m = Matcher(1, 1, Q, tb.shape[1], dev)
    m.run(make_desc(lg, bx, tb, tc, 0), tb)
    n = int(tb[0, 0, 0])
    pred_for_tgt = m.pred_for_tgt[0, :n].long()
    if int(m.status[0]) != 0:
        raise ValueError("cost matrix is infeasible or contains invalid numeric entries")
    order = torch.argsort(pred_for_tgt)                       # SciPy returns rows (predictions) ascending
    p_idx = pred_for_tgt[order]
    t_idx = order
    p_sel = m.tgt_for_pred[0] >= 0
    t_sel = torch.ones(n, dtype=torch.bool, device=dev)
    return t_idx, p_idx, t_sel, p_sel, tb[0, 1:1 + n], tc[0, 1:1 + n]
