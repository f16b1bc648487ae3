"""ORACLE (test infrastructure only) -- CPU restatement of the reference's Hungarian set loss.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Pinning: the assignment step IS the reference's own third-party dependency --
`scipy.optimize.linear_sum_assignment` (detr_tf/loss/hungarian_matching.py:7,29; SciPy is
unpinned by the reference, 1.15.3 is what this image ships) -- so matching parity is pinned
against the real thing.  Everything computed by TensorFlow ops is PARITY UNPINNED (no
reference tests / golden vectors exist, TF not installable): a line-faithful torch-CPU
restatement, float op order kept as in the reference.

All functions work on torch tensors (fp32 or fp64) so that autograd gives the oracle
gradients w.r.t. logits and boxes.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment


# ---- detr_tf/bbox.py ------------------------------------------------------------------
def xcycwh_to_xy_min_xy_max(b):
    """bbox.py:171-183 -- note the clip to [0, 1]."""
    xyxy = torch.cat([b[:, :2] - (b[:, 2:] / 2), b[:, :2] + (b[:, 2:] / 2)], dim=-1)
    return torch.clamp(xyxy, 0.0, 1.0)


def xcycwh_to_yx_min_yx_max(b):
    """bbox.py:186-196."""
    b = xcycwh_to_xy_min_xy_max(b)
    return torch.cat([b[:, 1:2], b[:, 0:1], b[:, 3:4], b[:, 2:3]], dim=-1)


def intersect(box_a, box_b):
    """bbox.py:29-72 (tile-based; broadcasting is the same arithmetic)."""
    above_right = torch.minimum(box_a[:, None, 2:], box_b[None, :, 2:])
    upper_left = torch.maximum(box_a[:, None, :2], box_b[None, :, :2])
    inter = torch.relu(above_right - upper_left)
    return inter[:, :, 0] * inter[:, :, 1]


def jaccard(box_a, box_b):
    """bbox.py:75-105 with return_union=True."""
    inter = intersect(box_a, box_b)
    area_a = ((box_a[:, 2] - box_a[:, 0]) * (box_a[:, 3] - box_a[:, 1]))[:, None]
    area_b = ((box_b[:, 2] - box_b[:, 0]) * (box_b[:, 3] - box_b[:, 1]))[None, :]
    union = area_a + area_b - inter
    return inter / union, union


def giou_matrix(p_xy, t_xy):
    """hungarian_matching.py:186-192 / loss.py:84-91: iou - (hull - union) / hull."""
    iou, union = jaccard(p_xy, t_xy)
    top_left = torch.minimum(p_xy[:, None, :2], t_xy[None, :, :2])
    bottom_right = torch.maximum(p_xy[:, None, 2:], t_xy[None, :, 2:])
    size = torch.relu(bottom_right - top_left)
    area = size[:, :, 0] * size[:, :, 1]
    return iou - (area - union) / area


# ---- detr_tf/loss/hungarian_matching.py -------------------------------------------------
def strip_header(t_bbox, t_class):
    """hungarian_matching.py:165-169; target layout detr_tf/data/processing.py:35-55."""
    n = int(t_bbox[0, 0])
    return t_bbox[1:1 + n], t_class[1:1 + n].reshape(-1)


def cost_matrix(t_bbox, t_class, p_bbox, p_class, fcost_class=1, fcost_bbox=5, fcost_giou=2):
    """hungarian_matching.py:171-195; t_* already stripped; returns [Q, n]."""
    p_xy = xcycwh_to_xy_min_xy_max(p_bbox)
    t_xy = xcycwh_to_xy_min_xy_max(t_bbox)
    softmax = torch.softmax(p_class, dim=-1)
    cost_class = -softmax[:, t_class.long()]
    cost_bbox = (p_bbox[:, None, :] - t_bbox[None, :, :]).abs().sum(-1)
    cost_giou = -giou_matrix(p_xy, t_xy)
    return fcost_bbox * cost_bbox + fcost_class * cost_class + fcost_giou * cost_giou


def lsap(cost_np):
    """np_tf_linear_sum_assignment hungarian_matching.py:27-46 -> (pred_idx, tgt_idx, pred_selector)."""
    rows, cols = linear_sum_assignment(cost_np)
    sel = np.zeros(cost_np.shape[0], dtype=bool)
    sel[rows] = True
    return rows.astype(np.int64), cols.astype(np.int64), sel


def hungarian_matching(t_bbox, t_class, p_bbox, p_class):
    """hungarian_matching.py:163-203 after the double name swap:
    returns (t_indices [n] true target idx, p_indices [n] true pred idx, p_selector bool[Q],
    stripped t_bbox, stripped t_class)."""
    tb, tc = strip_header(t_bbox, t_class)
    C = cost_matrix(tb, tc, p_bbox, p_class)
    rows, cols, sel = lsap(C.detach().cpu().numpy())
    return torch.from_numpy(cols), torch.from_numpy(rows), torch.from_numpy(sel), tb, tc


# ---- detr_tf/loss/loss.py ---------------------------------------------------------------
def loss_labels(p_class, t_class, t_indices, p_indices, p_selector, background_class):
    """loss.py:37-69."""
    neg_p = p_class[~p_selector]
    neg_t = torch.full((neg_p.shape[0],), background_class, dtype=torch.int64)
    weights = torch.cat([torch.full((neg_p.shape[0],), 0.1, dtype=p_class.dtype),
                         torch.full((t_indices.shape[0],), 1.0, dtype=p_class.dtype)])
    pos_p = p_class[p_indices]
    pos_t = t_class[t_indices].long()
    true_neg = (neg_p.argmax(-1) == background_class).to(p_class.dtype).mean()
    cls_pos = pos_p.argmax(-1)
    true_pos = (cls_pos != background_class).to(p_class.dtype).mean()
    pos_accuracy = (cls_pos == pos_t).to(p_class.dtype).mean()
    targets = torch.cat([neg_t, pos_t])
    preds = torch.cat([neg_p, pos_p])
    ce = torch.logsumexp(preds, -1) - preds.gather(1, targets[:, None])[:, 0]
    loss = (ce * weights).sum() / weights.sum()
    return loss, true_neg, true_pos, pos_accuracy


def loss_boxes(p_bbox, t_bbox, t_indices, p_indices):
    """loss.py:72-96 (the [N,N] GIoU matrix + diag_part is computed pairwise here: same values)."""
    pb = p_bbox[p_indices]
    tb = t_bbox[t_indices]
    n = pb.shape[0]
    p_xy = xcycwh_to_xy_min_xy_max(pb)
    t_xy = xcycwh_to_xy_min_xy_max(tb)
    l1 = (pb - tb).abs().sum() / n
    giou = torch.diagonal(giou_matrix(p_xy, t_xy))
    return (1 - giou).sum() / n, l1


def get_detr_losses(m_outputs, target_bbox, target_label, background_class, suffix=""):
    """loss.py:98-179."""
    pb_all, pc_all = m_outputs["pred_boxes"], m_outputs["pred_logits"]
    tbs, tcs, tis, pis, psel = [], [], [], [], []
    t_off = p_off = 0
    for b in range(pb_all.shape[0]):
        ti, pi, sel, tb, tc = hungarian_matching(target_bbox[b], target_label[b], pb_all[b], pc_all[b])
        tis.append(ti + t_off)
        pis.append(pi + p_off)
        psel.append(sel)
        tbs.append(tb)
        tcs.append(tc)
        t_off += tb.shape[0]
        p_off += pb_all.shape[1]
    tb = torch.cat(tbs)
    tc = torch.cat(tcs)
    ti = torch.cat(tis)
    pi = torch.cat(pis)
    sel = torch.cat(psel)
    pb = pb_all.reshape(-1, 4)
    pc = pc_all.reshape(-1, pc_all.shape[-1])
    label_cost, true_neg, true_pos, pos_acc = loss_labels(pc, tc, ti, pi, sel, background_class)
    giou_loss, l1_loss = loss_boxes(pb, tb, ti, pi)
    return {f"label_cost{suffix}": label_cost, f"true_neg{suffix}": true_neg,
            f"true_pos{suffix}": true_pos, f"pos_accuracy{suffix}": pos_acc,
            f"giou_loss{suffix}": giou_loss, f"l1_loss{suffix}": l1_loss}


def get_total_losss(losses):
    """loss.py:6-19 (substring match on the key names; weights 1/2/5)."""
    names, w = ["label_cost", "giou_loss", "l1_loss"], [1, 2, 5]
    total = 0
    for key in losses:
        sel = [i for i, n in enumerate(names) if n in key]
        if len(sel) == 1:
            total = total + losses[key] * w[sel[0]]
    return total


def get_losses(m_outputs, t_bbox, t_class, background_class):
    """loss.py:22-34."""
    losses = get_detr_losses(m_outputs, t_bbox, t_class, background_class)
    if "aux" in m_outputs:
        for a, aux in enumerate(m_outputs["aux"]):
            losses.update(get_detr_losses(aux, t_bbox, t_class, background_class, suffix=f"_{a}"))
    return get_total_losss(losses), losses


# ---- detr_tf/inference.py:68-95 -----------------------------------------------------------
def get_model_inference(m_outputs, background_class, bbox_format="xy_center"):
    pb = m_outputs["pred_boxes"][0]
    pl = m_outputs["pred_logits"][0]
    sm = torch.softmax(pl, -1)
    scores, labels = sm.max(-1)
    keep = torch.nonzero(labels != background_class)[:, 0]
    scores, labels, pb = scores[keep], labels[keep], pb[keep]
    if bbox_format == "xy_center":
        pass
    elif bbox_format == "xyxy":
        pb = xcycwh_to_xy_min_xy_max(pb)
    elif bbox_format == "yxyx":
        pb = xcycwh_to_yx_min_yx_max(pb)
    else:
        raise NotImplementedError()
    return pb, labels, scores


# ---- synthetic targets (layout detr_tf/data/processing.py:35-55) -----------------
def make_targets(B, seed=1235, max_rows=100, force_full=True, n_classes=90):
    rng = np.random.default_rng(seed)
    t_bbox = np.zeros((B, max_rows, 4), np.float32)
    t_class = np.zeros((B, max_rows, 1), np.int64)
    for b in range(B):
        n = int(np.clip(rng.poisson(7), 1, max_rows - 1))
        if force_full and b == B - 1:
            n = max_rows - 1
        t_bbox[b, 0, 0] = n
        t_bbox[b, 1:1 + n, 0:2] = rng.uniform(0.2, 0.8, (n, 2))
        t_bbox[b, 1:1 + n, 2:4] = rng.uniform(0.05, 0.4, (n, 2))
        t_class[b, 1:1 + n, 0] = rng.integers(1, n_classes + 1, n)
    return t_bbox, t_class
