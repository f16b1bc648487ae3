"""the vectorised mAP accumulator against the outputs of the reference's own cal_map / calc_map
(fixture tests/golden/refpy_map.npz, produced by scripts/crosscheck_reference.py under the TF stand-in), and the batched
device post-processing against the per-image oracle."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with np.load(os.path.join(GOLD, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def test_vectorised_map_equals_reference_code_outputs():
    from detr_tf.loss.compute_map import APAccumulator, average_precision
    fx = _load("refpy_map.npz")
    for ci in range(int(fx["n_cases"])):
        nb, n_img = int(fx[f"m{ci}_nb_class"]), int(fx[f"m{ci}_n_images"])
        acc = APAccumulator(nb)
        for i in range(n_img):
            acc.add_image(fx[f"m{ci}_{i}_p_bbox"], fx[f"m{ci}_{i}_p_cls"], fx[f"m{ci}_{i}_p_score"], fx[f"m{ci}_{i}_t_bbox"], fx[f"m{ci}_{i}_t_cls"])
        ref_pc = fx[f"m{ci}_per_class_ap"]                      # [10 thresholds, classes], -1 = empty class
        for c in range(nb):
            empty = len(acc.scores[c]) == 0 and acc.num_gt[c] == 0
            assert empty == bool(ref_pc[0, c] < 0)
            if empty:
                continue
            flags = np.concatenate(acc.flags[c], 0) if acc.flags[c] else np.zeros((0, 10), bool)
            for t in range(10):
                ap = average_precision(acc.scores[c], flags[:, t], int(acc.num_gt[c]))
                assert abs(ap - ref_pc[t, c]) < 1e-12, (ci, c, t, ap, ref_pc[t, c])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            maps = acc.result([f"class_{i}" for i in range(nb)], print_result=True)
        assert [str(k) for k in maps["box"].keys()] == [str(k) for k in fx[f"m{ci}_box_keys"]]
        assert list(maps["box"].values()) == list(fx[f"m{ci}_box_vals"])
        assert list(maps["mask"].values()) == list(fx[f"m{ci}_mask_vals"])
        assert buf.getvalue() == str(fx[f"m{ci}_table"])          # the printed table, character for character


def test_map_edge_cases():
    from detr_tf.loss.compute_map import APAccumulator
    acc = APAccumulator(3)
    assert acc.result()["box"]["all"] == 0 and acc.result()["mask"][50] == 0
    # a perfect detector on one class, nothing on the others
    gt = np.array([[0.1, 0.1, 0.4, 0.4], [0.5, 0.5, 0.9, 0.9]], np.float32)
    acc.add_image(gt, [1, 1], [0.9, 0.8], gt, [1, 1])
    r = acc.result()
    assert r["box"][50] == 100.0 and r["box"][95] == 100.0 and r["box"]["all"] == 100.0 and r["mask"]["all"] == 0.0
    # a duplicate detection is a false positive; images without predictions only add ground truths
    acc.add_image(np.concatenate([gt[:1], gt[:1]]), [1, 1], [0.7, 0.6], gt[:1], [1])
    acc.add_image(np.zeros((0, 4)), [], [], gt, [2, 2])
    r2 = acc.result()
    assert 0 < r2["box"][50] < 100.0


@pytest.mark.gpu
@pytest.mark.parametrize("B,Q,C,bg", [(5, 100, 92, 91), (2, 300, 92, 91), (3, 100, 5, 0), (1, 7, 3, 2)])
def test_batched_postprocess_vs_oracle(hip, B, Q, C, bg):
    """get_model_inference_batched: every image of the batch in one launch == the oracle's get_model_inference applied to
    each image (inference.py:68-95), three box formats, order of the kept queries preserved; images with no / only
    foreground queries included."""
    from detr_tf.inference import get_model_inference, get_model_inference_batched
    from oracle import set_loss_ref as L
    rng = np.random.default_rng(B * 1000 + Q + C)
    logits = rng.normal(size=(B, Q, C)).astype(np.float32) * 2
    logits[..., bg] += 1.5                                         # roughly half background
    if B > 1:
        logits[0, :, bg] += 50.0                                   # image 0: everything is background
        logits[1, :, bg] -= 50.0                                   # image 1: nothing is
    boxes = np.concatenate([rng.uniform(-0.1, 1.1, (B, Q, 2)), rng.uniform(0.01, 0.8, (B, Q, 2))], -1).astype(np.float32)
    out = {"pred_logits": torch.from_numpy(logits).cuda(), "pred_boxes": torch.from_numpy(boxes).cuda()}
    for fmt in ("xy_center", "xyxy", "yxyx"):
        dets = get_model_inference_batched(out, bg, fmt)
        assert len(dets) == B
        for b in range(B):
            rb, rl, rs = L.get_model_inference({"pred_logits": torch.from_numpy(logits[b:b + 1]), "pred_boxes": torch.from_numpy(boxes[b:b + 1])}, bg, fmt)
            gb, gl, gs = dets[b]
            assert gl.dtype == torch.int64 and torch.equal(gl.cpu(), rl), (fmt, b)
            assert torch.allclose(gb.cpu(), rb, rtol=0, atol=1e-6) and torch.allclose(gs.cpu(), rs, rtol=2e-6, atol=1e-7)
        if B > 1:
            assert dets[0][1].numel() == 0 and dets[1][1].numel() == Q
        e0 = get_model_inference(out, bg, fmt)                     # the reference-signature function: element 0 only
        assert torch.equal(e0[1], dets[0][1]) and torch.equal(e0[0], dets[0][0])
    with pytest.raises(NotImplementedError):
        get_model_inference(out, bg, "corners")
    with pytest.raises(RuntimeError):
        get_model_inference({"pred_logits": out["pred_logits"].cpu(), "pred_boxes": out["pred_boxes"].cpu()}, bg)


@pytest.mark.gpu
def test_eval_model_map_loop(hip, capsys):
    """evaluation.eval_model (eval.py:30-61 for any batch size): with a 'model' that returns the targets themselves as
    confident predictions the box mAP is 100 at every threshold and the reference's table is printed."""
    from detr_tf.evaluation import eval_model
    from detr_tf.training_config import TrainingConfig
    from oracle import set_loss_ref as L
    cfg = TrainingConfig()
    cfg.background_class = 91
    batches = []
    for i in range(3):
        tb, tc = L.make_targets(4, seed=60 + i, force_full=False)
        batches.append((np.zeros((4, 32, 32, 3), np.float32), tb, tc))

    class Echo:
        def __call__(self, images, training=False):
            tb, tc = self.cur
            B = tb.shape[0]
            logits = torch.full((B, 100, 92), -10.0)
            logits[:, :, 91] = 10.0
            boxes = torch.full((B, 100, 4), 0.5)
            for b in range(B):
                n = int(tb[b, 0, 0])
                boxes[b, :n] = torch.from_numpy(tb[b, 1:1 + n])
                logits[b, torch.arange(n), torch.from_numpy(tc[b, 1:1 + n, 0])] = 20.0
            return {"pred_logits": logits.cuda(), "pred_boxes": boxes.cuda()}

    echo = Echo()

    def stream():
        for im, tb, tc in batches:
            echo.cur = (tb, tc)
            yield im, tb, tc

    maps = eval_model(echo, cfg, [f"c{i}" for i in range(92)], stream())
    assert maps["box"]["all"] == 100.0 and maps["box"][50] == 100.0 and maps["box"][95] == 100.0 and maps["mask"]["all"] == 0.0
    assert "  box |" in capsys.readouterr().out
