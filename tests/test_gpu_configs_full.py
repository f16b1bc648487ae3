"""The BASELINE.json configurations at their OWN sizes (`configs_untested`).

  C3  DETR-R50 bf16 training step, B=8, 800x1333: the bf16 step against the fp32 step of the SAME HIP path (same weights, batch
      and dropout masks -- precision is the only difference): loss, matched index sets (flips counted), every gradient tensor.
      Measured (round 3): loss 5e-4, 5-6 % of the matched pairs flip (random-init predictions are near-degenerate), gradient
      relative L2 median 2.7-3.1 %, 90th percentile 3-4.5 %; bounds: 1e-3 / 12 % / 5 % / 10 %, named exceptions <= 40 %.
  C4  DETR-R101, 1000x1333: bf16 forward + set loss against the fp32 oracle (B=2), and the bf16 training step at B=8
      against the fp32 HIP step.
  C5  DETR-R50, 300 queries + 5 aux losses, B=16, 800x1333: properties of the training step (every assignment problem solved,
      matched count = number of targets, finite loss and gradients) and bf16 against fp32.
The oracle is the checker only where it finishes in seconds (C4 forward at B=2); at full batch the fp32 HIP path -- itself
pinned to the oracle at 1e-3 by test_gpu_model.py at C2's full size -- is the reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(train=True):
    from detr_tf.training_config import TrainingConfig
    cfg = TrainingConfig()
    cfg.background_class = 91
    cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = train
    cfg.target_batch = None
    return cfg


def _step(precision, params, images, t_bbox, t_class, *, backbone="resnet50", num_queries=100, dropout=0.1):
    """One training-mode step (forward, set loss, backward; no optimiser apply) -> loss, log, matching, flat gradient (CPU)."""
    from detr_tf import training
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.optimizers import setup_optimizers
    cfg = _cfg()
    model = get_detr_model(cfg, include_top=True, backbone=backbone, num_queries=num_queries, dropout=dropout, precision=precision)
    assert not model.load_weights(params)
    opt = setup_optimizers(model, cfg)
    out, total, log, _ = training.run_train_step(model, images, t_bbox, t_class, opt, cfg)
    torch.cuda.synchronize()
    m = out.set_loss.matcher
    res = dict(total=float(total), log={k: float(v) for k, v in log.items() if torch.is_tensor(v)},
               status=m.status.cpu().numpy().copy(), pred_for_tgt=m.pred_for_tgt.cpu().numpy().copy(),
               tgt_for_pred=m.tgt_for_pred.cpu().numpy().copy(), grad=model.engine.P.grad.cpu().clone(),
               offsets=dict(model.engine.P.offsets), logits=out["pred_logits"].cpu().clone(), boxes=out["pred_boxes"].cpu().clone())
    del model, opt, out
    torch.cuda.empty_cache()
    return res


# bounds of the fp32x3-vs-exact-fp32 comparison (measured values in the test's docstring / printed report)
# measured at this shape: loss equal to the last printed digit, 0 of 954 matched pairs differ, per-tensor gradient rel-L2 <= 2e-4 except the stem kernel
# (9e-4: the end of the longest chain, cancellation-dominated)
X3_LOSS_TOL, X3_FLIP_MAX, X3_MEDIAN_MAX, X3_P90_MAX, X3_WORST_MAX = 1e-6, 0.0, 1e-4, 5e-4, 5e-3


def _compare(a32, b16, t_bbox, levels, tag, *, loss_tol, flip_frac_max, median_max, p90_max, what="bf16", worst_max=None):
    """bf16 step `b16` against the fp32 step `a32` of the same path.  Returns the report dict (also printed)."""
    B = t_bbox.shape[0]
    n = t_bbox[:, 0, 0].astype(int)
    assert (a32["status"] == 0).all() and (b16["status"] == 0).all(), "an assignment problem was not solved"
    flips = matched = 0
    for lv in range(levels):
        for b in range(B):
            p = lv * B + b
            for res in (a32, b16):
                sel = res["tgt_for_pred"][p] >= 0
                assert int(sel.sum()) == n[b], (tag, lv, b, int(sel.sum()), n[b])          # matched count = number of targets
                assert len(set(res["pred_for_tgt"][p, :n[b]].tolist())) == n[b]            # ... one distinct query each
            flips += int((a32["pred_for_tgt"][p, :n[b]] != b16["pred_for_tgt"][p, :n[b]]).sum())
            matched += int(n[b])
    dloss = abs(a32["total"] - b16["total"]) / abs(a32["total"])
    l2 = []
    for name, (o, cnt) in a32["offsets"].items():
        ga, gb = a32["grad"][o:o + cnt].double(), b16["grad"][o:o + cnt].double()
        assert torch.isfinite(gb).all(), name
        if float(ga.norm()) > 1e-10:
            l2.append((float((ga - gb).norm() / ga.norm()), name))
    l2.sort(reverse=True)
    vals = np.array([v for v, _ in l2])
    rep = dict(tag=tag, loss_fp32=a32["total"], loss_bf16=b16["total"], loss_rel=dloss, flips=flips, matched=matched,
               grad_median=float(np.median(vals)), grad_p90=float(np.quantile(vals, 0.9)), grad_worst=l2[:6])
    print(f"[{tag}] {what} vs fp32 HIP step: loss {a32['total']:.5f} / {b16['total']:.5f} (rel {dloss:.2e}); matching flips {flips} of "
          f"{matched}; gradient rel-L2 median {rep['grad_median']:.3g} p90 {rep['grad_p90']:.3g}; worst {l2[:6]}")
    assert np.isfinite(b16["total"]) and dloss <= loss_tol, rep
    assert flips <= flip_frac_max * matched, rep
    assert rep["grad_median"] <= median_max and rep["grad_p90"] <= p90_max, rep
    # named exceptions (measured 0.10 .. 0.26): cancellation-dominated tensors -- the q / k projection of decoder layer 0's self
    # attention (its true gradient is ~0: tgt is the zero target, every query sees the same keys), query_embed (a sum of
    # nearly cancelling terms over layers), and the stem kernel (the end of the longest bf16 chain).  Everything else <= 12 %.
    loose = ("transformer/decoder/layer_0/self_attn/in_proj", "query_embed/kernel", "backbone/conv1/kernel")
    if worst_max is not None:             # (an fp32-accuracy mode: one bound for every tensor)
        assert l2[0][0] <= worst_max, (l2[0], rep)
        return rep
    for v, name in l2:
        assert v <= (0.40 if any(name.startswith(t) for t in loose) else 0.12), (name, v, rep)
    return rep


def test_c3_bf16_train_step_vs_fp32_step_at_b8_800x1333(hip):
    """C3 at its own shape: per-tensor gradient agreement of the bf16 step with the fp32 step (same path, same masks)."""
    from oracle import detr_ref as R, set_loss_ref as L
    params = R.make_params(0)
    images = np.random.default_rng(1234).normal(size=(8, 800, 1333, 3)).astype(np.float32)
    t_bbox, t_class = L.make_targets(8, seed=1235)
    a32 = _step("fp32", params, images, t_bbox, t_class)
    b16 = _step("bf16", params, images, t_bbox, t_class)
    # exceptions named by test_bf16_compute_mode_deviation_from_fp32_oracle: cancellation-dominated tensors (query_embed, the
    # zero-gradient q / k projections of decoder layer 0) sit far above the median; they are in the report, not in the bounds
    # (round 6, measured: loss 6.2e-4, 65 of 954 matched pairs differ = 6.8 %, gradient rel-L2 median 0.024, p90 0.029)
    _compare(a32, b16, t_bbox, 6, "C3 R50 B8 800x1333", loss_tol=1e-3, flip_frac_max=0.10, median_max=0.04, p90_max=0.06)


def test_c3_fp32x3_train_step_vs_exact_fp32_step_at_b8_800x1333(hip):
    """Round 6: the fp32x3 step (fp32 storage, bf16 matrix pipe, six exact partial products per product) against the exact-fp32 step at C3's own
    shape -- the same comparison as the bf16 test above with bounds three orders of magnitude tighter: fp32x3 is an fp32-accuracy mode."""
    from oracle import detr_ref as R, set_loss_ref as L
    params = R.make_params(0)
    images = np.random.default_rng(1234).normal(size=(8, 800, 1333, 3)).astype(np.float32)
    t_bbox, t_class = L.make_targets(8, seed=1235)
    a32 = _step("fp32", params, images, t_bbox, t_class)
    x3 = _step("fp32x3", params, images, t_bbox, t_class)
    _compare(a32, x3, t_bbox, 6, "C3 R50 B8 800x1333 fp32x3", loss_tol=X3_LOSS_TOL, flip_frac_max=X3_FLIP_MAX, median_max=X3_MEDIAN_MAX, p90_max=X3_P90_MAX,
             what="fp32x3", worst_max=X3_WORST_MAX)


def test_c4_r101_bf16_forward_loss_vs_fp32_oracle_at_1000x1333(hip):
    """C4's backbone and input size in C3/C4's compute mode: bf16 forward + set loss against the fp32 ORACLE (B=2)."""
    from detr_tf.loss.loss import get_losses
    from detr_tf.networks.detr import get_detr_model
    from oracle import detr_ref as R, set_loss_ref as L
    cfg = _cfg(train=False)
    params = R.make_params(7, blocks=R.RESNET101_BLOCKS)
    images = np.random.default_rng(8).normal(size=(2, 1000, 1333, 3)).astype(np.float32)
    t_bbox, t_class = L.make_targets(2, seed=9, force_full=False)
    model = get_detr_model(cfg, include_top=True, backbone="resnet101", dropout=0.0, precision="bf16")
    assert not model.load_weights(params)
    out = model(images, training=False)
    total, log = get_losses(out, t_bbox, t_class, cfg)
    P = R.to_torch(params)
    with torch.no_grad():
        refs = [R.detr_forward(torch.from_numpy(images[b:b + 1]), P, blocks=R.RESNET101_BLOCKS) for b in range(2)]
    ref = {"pred_logits": torch.cat([r["pred_logits"] for r in refs]), "pred_boxes": torch.cat([r["pred_boxes"] for r in refs]),
           "aux": [{"pred_logits": torch.cat([r["aux"][i]["pred_logits"] for r in refs]),
                    "pred_boxes": torch.cat([r["aux"][i]["pred_boxes"] for r in refs])} for i in range(5)]}
    ref_total, _ = L.get_losses(ref, torch.from_numpy(t_bbox), torch.from_numpy(t_class), 91)
    dl = float((out["pred_logits"].cpu() - ref["pred_logits"]).abs().max() / ref["pred_logits"].abs().max())
    db = float((out["pred_boxes"].cpu() - ref["pred_boxes"]).abs().max() / ref["pred_boxes"].abs().max())
    dloss = abs(float(total) - float(ref_total)) / abs(float(ref_total))
    print(f"[C4 R101 1000x1333 bf16 vs fp32 oracle] loss {dloss:.2e} logits {dl:.2e} boxes {db:.2e}")
    assert tuple(out["pred_logits"].shape) == (2, 100, 92)
    assert dloss <= 1e-3, (float(total), float(ref_total))
    assert dl < 2e-2 and db < 2e-2, (dl, db)


def test_c4_r101_bf16_train_step_at_b8_1000x1333(hip):
    """C4 at its own batch and size: the bf16 training step against the fp32 step of the same path."""
    from oracle import detr_ref as R, set_loss_ref as L
    params = R.make_params(7, blocks=R.RESNET101_BLOCKS)
    images = np.random.default_rng(18).normal(size=(8, 1000, 1333, 3)).astype(np.float32)
    t_bbox, t_class = L.make_targets(8, seed=19)
    a32 = _step("fp32", params, images, t_bbox, t_class, backbone="resnet101")
    b16 = _step("bf16", params, images, t_bbox, t_class, backbone="resnet101")
    _compare(a32, b16, t_bbox, 6, "C4 R101 B8 1000x1333", loss_tol=1e-3, flip_frac_max=0.12, median_max=0.05, p90_max=0.10)


def test_c5_300_queries_b16_800x1333_train_step(hip):
    """C5 at its own shape (B=16, 800x1333, 300 queries, 5 aux losses, one image forced to 99 targets): 96 assignment problems
    of 300 x n solved (status 0, one distinct query per target), finite loss / gradients, bf16 against fp32."""
    from oracle import detr_ref as R, set_loss_ref as L
    params = R.make_params(43, num_queries=300)
    images = np.random.default_rng(44).normal(size=(16, 800, 1333, 3)).astype(np.float32)
    t_bbox, t_class = L.make_targets(16, seed=45)
    a32 = _step("fp32", params, images, t_bbox, t_class, num_queries=300)
    assert a32["logits"].shape == (16, 300, 92) and np.isfinite(a32["total"])
    assert torch.isfinite(a32["grad"]).all() and float(a32["grad"].abs().max()) > 0
    assert float(a32["boxes"].min()) >= 0.0 and float(a32["boxes"].max()) <= 1.0
    b16 = _step("bf16", params, images, t_bbox, t_class, num_queries=300)
    _compare(a32, b16, t_bbox, 6, "C5 R50 Q300 B16 800x1333", loss_tol=1e-3, flip_frac_max=0.12, median_max=0.05, p90_max=0.10)


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_c3_step_is_bit_identical_run_to_run_at_full_size(precision):
    """Run-to-run determinism at the BENCH shape (B = 8, 800x1333, dropout 0.1, one image with 99 targets): the same step -- same
    weights, batch and dropout masks -- repeated four times must give bit-identical logits, boxes and EVERY gradient entry.  Round 5
    found the one exception this test now guards: the column sums behind input_proj/bias (and the heads' biases) were float atomics
    over 132 row chunks -- invisible at the small test shapes, where a column fits one chunk (scripts/experiments/determinism_full.py).
    The ring GEMM kernels (csrc/gemm_ring.h) take 80 launches of this step: a stage read before its DMA has landed would show
    up here as rare differing tiles."""
    from detr_tf import training
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.optimizers import setup_optimizers
    from oracle.set_loss_ref import make_targets
    cfg = _cfg()
    cfg.batch_size = 8
    model = get_detr_model(cfg, include_top=True, device="cuda:0", seed=0, dropout=0.1, precision=precision)
    opt = setup_optimizers(model, cfg)
    images = torch.from_numpy(np.random.default_rng(1234).normal(size=(8, 800, 1333, 3)).astype(np.float32)).cuda()
    tb, tc = make_targets(8, seed=5, force_full=True)
    tb, tc = torch.from_numpy(tb).cuda(), torch.from_numpy(tc).cuda()
    ref = None
    for rep in range(4):
        step0 = model.engine._step_no
        out, total, log, _ = training.run_train_step(model, images, tb, tc, opt, cfg)
        model.engine._step_no = step0                 # the same dropout masks every repetition
        torch.cuda.synchronize()
        cur = (out["pred_logits"].clone(), out["pred_boxes"].clone(), model.engine.P.grad.clone())
        if ref is None:
            ref = cur
            assert float(cur[2].abs().max()) > 0
            continue
        for what, a, b in zip(("logits", "boxes", "gradients"), ref, cur):
            nd = int((a != b).sum())
            if nd and what == "gradients":
                diff = a != b
                names = [k for k, (o, n) in model.engine.P.offsets.items() if bool(diff[o:o + n].any())]
                raise AssertionError(f"{precision} rep {rep}: {nd} gradient entries differ between two runs: {names[:8]}")
            assert nd == 0, f"{precision} rep {rep}: {nd} {what} entries differ between two runs"
    del model, opt
    torch.cuda.empty_cache()
