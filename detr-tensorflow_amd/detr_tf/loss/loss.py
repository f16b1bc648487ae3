"""DETR set loss on the device (drop-in for detr_tf/loss/loss.py:6-179).

`get_losses(m_outputs, t_bbox, t_class, config)` keeps the reference signature and returns
`(total_loss, log_dict)` with the same 36 keys (label_cost, true_neg, true_pos, pos_accuracy,
giou_loss, l1_loss, + suffixes _0.._4) and the 1/2/5 weighting of `get_total_losss`.
All arithmetic runs in csrc/setloss.hip: cost matrix (K12), exact assignment (K13), loss sums /
finalisation / gradients (K14).  Under data parallelism the per-level normalisers (sum of CE
weights, number of matched boxes) are all-reduced between the "sums" and "finalize" kernels, so
a B x world step has the reference's single-device whole-batch semantics (loss.py:66-67,82,94).
"""
from ctypes import byref, c_float

import torch

from .. import _hip as hip
from .hungarian_matching import Matcher, make_desc

LOSS_NAMES = ("label_cost", "true_neg", "true_pos", "pos_accuracy", "giou_loss", "l1_loss")


def get_total_losss(losses):
    """loss.py:6-19: sum of 1*label_cost + 2*giou_loss + 5*l1_loss over every (suffixed) key."""
    weights = {"label_cost": 1, "giou_loss": 2, "l1_loss": 5}
    total = 0
    for key, value in losses.items():
        hit = [w for n, w in weights.items() if n in key]
        if len(hit) == 1:
            total = total + value * hit[0]
    return total


class SetLoss:
    """Device state of the set loss for one (levels, B, Q, C, R) shape."""

    _cache = {}

    @classmethod
    def get(cls, levels, B, Q, C, R, device):
        key = (levels, B, Q, C, R, str(device))
        if key not in cls._cache:
            cls._cache[key] = cls(levels, B, Q, C, R, device)
        return cls._cache[key]

    def __init__(self, levels, B, Q, C, R, device):
        self.levels, self.B, self.Q, self.C, self.R = levels, B, Q, C, R
        self.matcher = Matcher(levels, B, Q, R, device)
        self.sums = torch.zeros(levels * 10, dtype=torch.float32, device=device)
        self.losses = torch.zeros(levels, 6, dtype=torch.float32, device=device)
        self.total = torch.zeros(1, dtype=torch.float32, device=device)
        self.d_logits = torch.zeros(levels, B, Q, C, dtype=torch.float32, device=device)
        self.d_boxes = torch.zeros(levels, B, Q, 4, dtype=torch.float32, device=device)
        self.desc = None
        self.reduce_sums = None          # hook: callable(tensor) doing the data-parallel all-reduce

    def forward(self, logits, boxes, t_bbox, t_class, background_class):
        self.desc = make_desc(logits, boxes, t_bbox, t_class, background_class)
        self._keep = (logits, boxes, t_bbox, t_class)
        tfp = self.matcher.run(self.desc, t_bbox)
        hip.zero_(self.sums)
        hip.call("detr_hip_set_loss_sums_f32", byref(self.desc), tfp.data_ptr(), self.sums.data_ptr())
        if self.reduce_sums is not None:
            self.reduce_sums(self.sums)
        hip.call("detr_hip_set_loss_finalize_f32", self.sums.data_ptr(), self.levels, self.losses.data_ptr(),
                 self.total.data_ptr())
        return self.total, self.losses

    def grad(self, loss_scale=1.0):
        """Gradients of loss_scale * total w.r.t. logits / boxes, laid out [Lv,B,Q,*] contiguous."""
        lg = self._keep[0]
        d = make_desc(lg, self._keep[1], self._keep[2], self._keep[3], self.desc.background_class)
        # the gradient tensors use the same (level, image, query) strides as contiguous [Lv,B,Q,*]
        assert lg.is_contiguous() and self._keep[1].is_contiguous(), "set-loss gradients need contiguous head outputs"
        hip.call("detr_hip_set_loss_grad_f32", byref(d), self.matcher.tgt_for_pred.data_ptr(), self.sums.data_ptr(),
                 c_float(loss_scale), self.d_logits.data_ptr(), self.d_boxes.data_ptr())
        return self.d_logits, self.d_boxes


def _stack_levels(m_outputs):
    """[aux_0 .. aux_{n-1}, main] -> contiguous [Lv,B,Q,*] (main is the LAST level, detr.py:190-202)."""
    if hasattr(m_outputs, "levels_logits"):
        return m_outputs.levels_logits, m_outputs.levels_boxes
    aux = m_outputs.get("aux", [])
    lg = torch.stack([a["pred_logits"] for a in aux] + [m_outputs["pred_logits"]]).contiguous().float()
    bx = torch.stack([a["pred_boxes"] for a in aux] + [m_outputs["pred_boxes"]]).contiguous().float()
    return lg, bx


def _prep_targets(t_bbox, t_class, device):
    tb = torch.as_tensor(t_bbox).to(device=device, dtype=torch.float32).contiguous()
    tc = torch.as_tensor(t_class).to(device=device, dtype=torch.int64)
    tc = tc.reshape(tb.shape[0], tb.shape[1]).contiguous()
    return tb, tc


def log_from_losses(vals):
    """The reference's log dict (loss.py:23-30: main level first, then aux 0..n-1 with suffixes) as 0-d views of ONE
    [levels, 6] tensor -- the caller passes a fresh clone, so a step's log never aliases the next step's."""
    Lv = vals.shape[0]
    log = {}
    for lv in [Lv - 1] + list(range(Lv - 1)):
        suffix = "" if lv == Lv - 1 else f"_{lv}"
        for k, name in enumerate(LOSS_NAMES):
            log[name + suffix] = vals[lv, k]
    return log


def get_losses(m_outputs, t_bbox, t_class, config):
    """loss.py:22-34.  Returns (total_loss 0-d tensor, dict name -> 0-d tensor)."""
    lg, bx = _stack_levels(m_outputs)
    tb, tc = _prep_targets(t_bbox, t_class, lg.device)
    Lv, B, Q, C = lg.shape
    sl = SetLoss.get(Lv, B, Q, C, tb.shape[1], lg.device)
    hook = getattr(m_outputs, "reduce_sums", None)
    sl.reduce_sums = hook
    total, losses = sl.forward(lg, bx, tb, tc, config.background_class)
    if getattr(config, "check_matching", False):
        # opt-in (it synchronises): the reference's SciPy call raises ValueError on NaN / inf / infeasible cost matrices
        # (hungarian_matching.py:29); the device matcher records a status per problem instead of stopping the step
        if bool((sl.matcher.status != 0).any()):
            raise ValueError("cost matrix is infeasible or contains invalid numeric entries")
    log = log_from_losses(losses.clone())
    total_loss = total.clone()[0]
    if hasattr(m_outputs, "set_loss"):
        m_outputs.set_loss = sl
    return total_loss, log
