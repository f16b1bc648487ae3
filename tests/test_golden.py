"""Golden-fixture tests.  tests/golden/tiny_r50_e1d2_64x96.npz was captured from the CPU oracle by
tests/golden/make_golden.py (the reference ships no golden vectors).
 * CPU: the oracle still reproduces the fixture (guards the checker itself against drift);
 * GPU: the HIP path reproduces it through the drop-in API: output structure, the 36-key loss
   dict, matched index sets per (level, image), per-tensor gradient norms, parameter updates after
   two accumulated steps, get_model_inference in the three box formats."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "tiny_r50_e1d2_64x96.npz")


@pytest.fixture(scope="module")
def gold():
    with np.load(FIX) as z:
        return {k: z[k] for k in z.files}


def test_oracle_reproduces_golden(gold):
    from oracle import detr_ref as R, set_loss_ref as L
    seed, num_enc, num_dec, B, H, W = (int(v) for v in gold["meta"])
    P = R.to_torch(R.make_params(seed, num_enc=num_enc, num_dec=num_dec))
    with torch.no_grad():
        out = R.detr_forward(torch.from_numpy(gold["images"]), P, num_enc=num_enc, num_dec=num_dec)
        total, losses = L.get_losses(out, torch.from_numpy(gold["t_bbox"]), torch.from_numpy(gold["t_class"]), 91)
    assert np.allclose(out["pred_logits"].numpy(), gold["pred_logits"], rtol=0, atol=2e-5)
    assert np.allclose(out["pred_boxes"].numpy(), gold["pred_boxes"], rtol=0, atol=2e-6)
    assert list(losses.keys()) == [str(k) for k in gold["loss_keys"]]
    assert np.allclose([float(v) for v in losses.values()], gold["loss_vals"], rtol=1e-5, atol=1e-6)
    assert abs(float(total) - float(gold["total"])) < 1e-4


@pytest.mark.gpu
def test_hip_path_reproduces_golden(hip, gold):
    from detr_tf import training
    from detr_tf.inference import get_model_inference
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.optimizers import setup_optimizers
    from detr_tf.training_config import TrainingConfig
    from oracle import detr_ref as R
    seed, num_enc, num_dec, B, H, W = (int(v) for v in gold["meta"])
    cfg = TrainingConfig()
    cfg.background_class = 91
    cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
    cfg.batch_size, cfg.target_batch = 2, 4                       # gradient_aggregate = 2
    params = R.make_params(seed, num_enc=num_enc, num_dec=num_dec)
    model = get_detr_model(cfg, include_top=True, num_encoder_layers=num_enc, num_decoder_layers=num_dec, dropout=0.0)
    model.load_weights(params)
    opt = setup_optimizers(model, cfg)
    out, total, log, steps = training.run_train_step(model, gold["images"], gold["t_bbox"], gold["t_class"], opt, cfg)
    assert np.abs(out["pred_logits"].cpu().numpy() - gold["pred_logits"]).max() < 2e-4 * np.abs(gold["pred_logits"]).max()
    assert np.abs(out["pred_boxes"].cpu().numpy() - gold["pred_boxes"]).max() < 2e-4
    assert np.abs(out["aux"][0]["pred_logits"].cpu().numpy() - gold["aux0_logits"]).max() < 2e-4 * np.abs(gold["aux0_logits"]).max()
    assert [k for k in log if not k.endswith("_lr")] == [str(k) for k in gold["loss_keys"]]
    got = np.array([float(log[str(k)]) for k in gold["loss_keys"]])
    assert np.allclose(got, gold["loss_vals"], rtol=1e-3, atol=1e-4), (got, gold["loss_vals"])
    assert abs(float(total) * 2 - float(gold["total"])) < 1e-3 * float(gold["total"])     # total / gradient_aggregate
    # matched index sets: order main, aux0 ; engine levels are [aux0, main]
    # (near-duplicate predictions of this tiny random model make the optimum non-unique at the 1e-6 level, so
    #  the check here is: a valid one-to-one matching of every target whose total cost equals the stored optimum;
    #  index-for-index equality of the matched sets is asserted on the reference-produced fixtures, whose optima are
    #  unique: tests/test_refpy_fixtures.py::test_hip_set_loss_matching_inference_vs_reference_code_outputs)
    tfp = out.set_loss.matcher.tgt_for_pred.cpu().numpy().reshape(num_dec, B, 100)
    for g_idx, (lv, b) in enumerate([(num_dec - 1, bb) for bb in range(B)] + [(0, bb) for bb in range(B)]):
        n = int(gold["t_bbox"][b, 0, 0])
        mine, ref = tfp[lv, b], gold["matched"][g_idx]
        C = gold["costs"][g_idx]
        assert sorted(mine[mine >= 0].tolist()) == list(range(n))
        cost_mine = sum(C[q, mine[q]] for q in range(100) if mine[q] >= 0)
        cost_ref = sum(C[q, ref[q]] for q in range(100) if ref[q] >= 0)
        assert abs(cost_mine - cost_ref) < 1e-4 * max(1.0, abs(cost_ref)), (lv, b, cost_mine, cost_ref)
    # gradient norms (this step's gradient is d(total/2))
    names = [str(n) for n in gold["grad_names"]]
    gn = np.array([float(model.engine.P.gviews[n].norm()) * 2 for n in names])
    ref = gold["grad_norms"]
    big = ref > 1e-6 * ref.max()
    assert np.allclose(gn[big], ref[big], rtol=2e-3), np.abs(gn[big] / ref[big] - 1).max()
    # two accumulated steps on the same batch, then ONE Adam apply per group
    for step in range(2):
        if step == 1:
            out, total, log, steps = training.run_train_step(model, gold["images"], gold["t_bbox"], gold["t_class"], opt, cfg)
        for name in steps:
            training.aggregate_grad_and_apply(name, opt, steps[name]["gradients"], step, cfg)
    torch.cuda.synchronize()
    for key in ("class_embed/bias", "transformer/decoder/norm/gamma", "input_proj/bias"):
        upd_ref = gold["upd_" + key.replace("/", ".")]
        upd = model.engine.P.views[key].cpu().numpy() - params[key]
        lr = 1e-5 if key.startswith("input_proj") else 1e-4
        assert np.abs(upd - upd_ref).mean() < 0.03 * lr, (key, np.abs(upd - upd_ref).mean() / lr)
    # inference post-processing on the ORIGINAL weights
    model.load_weights(params)
    out = model(gold["images"])
    for fmt in ("xy_center", "xyxy", "yxyx"):
        b, l, s = get_model_inference(out, 91, fmt)
        assert np.array_equal(l.cpu().numpy(), gold[f"inf_{fmt}_labels"])
        if len(gold[f"inf_{fmt}_labels"]):
            assert np.abs(b.cpu().numpy() - gold[f"inf_{fmt}_boxes"]).max() < 2e-4
            assert np.abs(s.cpu().numpy() - gold[f"inf_{fmt}_scores"]).max() < 2e-4
