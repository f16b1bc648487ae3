"""COCO-style box mAP accumulator -- the arithmetic of the reference's detr_tf/loss/compute_map.py (:17-83 APDataObject,
:124-139 compute_overlaps, :141-168 calc_map, :183-272 cal_map), vectorised with NumPy.

The reference walks predictions x ground truths x classes x 10 IoU thresholds in pure Python for every image.  Here the
IoU matrix is one broadcast, the greedy matching of an image runs for all ten thresholds at once (the only sequential
dimension left is the detections of ONE class in score order, because a ground truth is consumed by the first detection
that claims it), and the AP integration is cummax + searchsorted.  Semantics kept exactly:
  * boxes are [y1, x1, y2, x2]; a detection matches the unused same-class ground truth of LARGEST IoU strictly above the
    threshold, the lowest index on ties (cal_map :233-247);
  * detections are visited by descending score, stable (:207); the data points of a class are sorted by descending
    score, stable, before the precision / recall sweep (:40);
  * AP = mean of the 101-point interpolated precision envelope (:61-83); classes with neither detections nor ground
    truths are left out of the mean (:149-150); `all` = mean over the ten thresholds (:160);
  * the reference also accumulates a 'mask' metric on all-zero dummy masks (eval.py:54): every detection is a false
    positive there, so every non-empty class has mask AP 0; `result()` reports exactly that.
"""
from collections import OrderedDict

import numpy as np

IOU_THRESHOLDS = [x / 100.0 for x in range(50, 100, 5)]          # eval.py:33


def compute_overlaps(boxes1, boxes2):
    """IoU matrix [len(boxes1), len(boxes2)] of [y1, x1, y2, x2] boxes (compute_map.py:103-139)."""
    b1, b2 = np.asarray(boxes1, np.float64).reshape(-1, 4), np.asarray(boxes2, np.float64).reshape(-1, 4)
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    y1 = np.maximum(b1[:, None, 0], b2[None, :, 0])
    y2 = np.minimum(b1[:, None, 2], b2[None, :, 2])
    x1 = np.maximum(b1[:, None, 1], b2[None, :, 1])
    x2 = np.minimum(b1[:, None, 3], b2[None, :, 3])
    inter = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area1[:, None] + area2[None, :] - inter)


def average_precision(scores, is_true, num_gt_positives):
    """APDataObject.get_ap (:37-83) for one (class, threshold): arrays in push order."""
    if num_gt_positives == 0:
        return 0
    order = np.argsort(-np.asarray(scores, np.float64), kind="stable")
    tp = np.asarray(is_true, bool)[order]
    if tp.size == 0:
        return 0.0
    num_true = np.cumsum(tp)
    precisions = num_true / np.arange(1, tp.size + 1)
    recalls = num_true / num_gt_positives
    precisions = np.maximum.accumulate(precisions[::-1])[::-1]            # right-to-left running maximum (:61-63)
    idx = np.searchsorted(recalls, np.arange(101) / 100.0, side="left")
    y = np.where(idx < tp.size, precisions[np.minimum(idx, tp.size - 1)], 0.0)
    return float(y.sum() / 101)


class APAccumulator:
    """ap_data of eval.py:34-37 / WandbSender.init_ap_data, box metric, all thresholds."""

    def __init__(self, nb_class, iou_thresholds=None):
        self.thresholds = list(IOU_THRESHOLDS if iou_thresholds is None else iou_thresholds)
        self.nb_class = int(nb_class)
        nt = len(self.thresholds)
        self.scores = [[] for _ in range(self.nb_class)]                 # per class: detection scores in push order
        self.flags = [[] for _ in range(self.nb_class)]                  # per class: [n_det, nt] true-positive flags
        self.num_gt = np.zeros(self.nb_class, np.int64)
        self._nt = nt

    def add_image(self, p_bbox, p_labels, p_scores, t_bbox, t_labels):
        """cal_map (:183-272) for one image; boxes [y1, x1, y2, x2]."""
        p_bbox = np.asarray(p_bbox, np.float64).reshape(-1, 4)
        t_bbox = np.asarray(t_bbox, np.float64).reshape(-1, 4)
        classes = np.asarray(p_labels).astype(int).reshape(-1)
        scores = np.asarray(p_scores).astype(float).reshape(-1)
        gt_classes = np.asarray(t_labels).astype(int).reshape(-1)
        iou = compute_overlaps(p_bbox, t_bbox) if len(classes) and len(gt_classes) else np.zeros((len(classes), len(gt_classes)))
        order = np.argsort(-scores, kind="stable")
        thr = np.asarray(self.thresholds)[:, None]
        for c in sorted(set(classes.tolist()) | set(gt_classes.tolist())):
            gt_idx = np.nonzero(gt_classes == c)[0]
            self.num_gt[c] += len(gt_idx)
            det = order[classes[order] == c]
            if len(det) == 0:
                continue
            flags = np.zeros((len(det), self._nt), bool)
            if len(gt_idx):
                used = np.zeros((self._nt, len(gt_idx)), bool)
                sub = iou[np.ix_(det, gt_idx)]
                for k in range(len(det)):                                  # sequential: a ground truth is consumed once
                    cand = np.where(used | ~(sub[k][None, :] > thr), -np.inf, sub[k][None, :])
                    j = np.argmax(cand, axis=1)                            # first index of the largest IoU (ties: lowest j)
                    hit = cand[np.arange(self._nt), j] > -np.inf
                    used[np.nonzero(hit)[0], j[hit]] = True
                    flags[k] = hit
            self.scores[c].extend(scores[det].tolist())
            self.flags[c].append(flags)

    def result(self, class_names=None, print_result=False):
        """calc_map (:141-168): {'box': {'all', 50, 55, ..., 95}, 'mask': {...}} rounded to 2 decimals."""
        n_cls = self.nb_class if class_names is None else len(class_names)
        aps = [[] for _ in self.thresholds]
        n_nonempty = 0
        for c in range(n_cls):
            n_det = len(self.scores[c])
            if n_det == 0 and self.num_gt[c] == 0:
                continue                                                   # is_empty()
            n_nonempty += 1
            flags = np.concatenate(self.flags[c], 0) if n_det else np.zeros((0, self._nt), bool)
            for t in range(self._nt):
                aps[t].append(average_precision(self.scores[c], flags[:, t], int(self.num_gt[c])))
        all_maps = {"box": OrderedDict(), "mask": OrderedDict()}
        for kind in ("box", "mask"):
            all_maps[kind]["all"] = 0
            for t, th in enumerate(self.thresholds):
                vals = aps[t] if kind == "box" else [0.0] * n_nonempty    # dummy zero masks: every detection is a false positive
                all_maps[kind][int(th * 100)] = sum(vals) / len(vals) * 100 if len(vals) > 0 else 0
            all_maps[kind]["all"] = sum(all_maps[kind].values()) / (len(all_maps[kind].values()) - 1)
        if print_result:
            print_maps(all_maps)
        return {k: {j: round(u, 2) for j, u in v.items()} for k, v in all_maps.items()}


def print_maps(all_maps):
    """The table of compute_map.py:170-181."""
    def make_row(vals):
        return (" %5s |" * len(vals)) % tuple(vals)

    def make_sep(n):
        return "-------+" * n

    print()
    print(make_row([""] + [(".%d " % x if isinstance(x, int) else x + " ") for x in all_maps["box"].keys()]))
    print(make_sep(len(all_maps["box"]) + 1))
    for kind in ("box", "mask"):
        print(make_row([kind] + ["%.2f" % x if x < 100 else "%.1f" % x for x in all_maps[kind].values()]))
    print(make_sep(len(all_maps["box"]) + 1))
    print()
