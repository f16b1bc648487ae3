"""Replay of tests/golden/refpy_*.npz: DATA-ONLY fixtures holding seeded inputs and the outputs of the REFERENCE'S OWN
Python (detr_tf/loss/loss.py, loss/hungarian_matching.py, bbox.py, inference.py, networks/*.py, optimizers.py, training.py
of Visual-Behavior/detr-tensorflow) executed in the build container under a torch-backed TensorFlow stand-in
(scripts/tf_shim.py + scripts/crosscheck_reference.py, both committed; TensorFlow itself is not installable).  The
fixtures pin the reference's control flow -- header stripping, the double name swap of the matcher's six-tuple, index
offsets, class weights, whole-batch normalisers, aux ordering, the three output modes, dropout placement, the variable
partition, the accumulate / apply cadence, the console line -- for the oracle (CPU tests) and for the HIP path (GPU tests).
They do not pin TensorFlow's kernels: parity stays "unpinned at the TF boundary" (DESIGN.md section 2)."""
import hashlib
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with np.load(os.path.join(GOLD, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def _hash(params):
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(k.encode())
        h.update(np.ascontiguousarray(params[k]).tobytes())
    return h.hexdigest()


def _rel(a, b):
    a = (a.detach().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).double()
    b = (b.detach().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b))).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


def _levels_to_outputs(logits, boxes, wrap):
    Lv = logits.shape[0]
    out = {"pred_logits": wrap(logits[Lv - 1]), "pred_boxes": wrap(boxes[Lv - 1])}
    if Lv > 1:
        out["aux"] = [{"pred_logits": wrap(logits[i]), "pred_boxes": wrap(boxes[i])} for i in range(Lv - 1)]
    return out


# ======================================================================================================
# CPU: the oracle against the reference-code outputs
# ======================================================================================================
def test_oracle_set_loss_matching_inference_equal_reference_code_outputs():
    from oracle import set_loss_ref as L
    fx = _load("refpy_setloss.npz")
    for ci in range(int(fx["n_cases"])):
        lg, bx, tb, tc, bg = (fx[f"c{ci}_{n}"] for n in ("logits", "boxes", "t_bbox", "t_class", "bg"))
        out = _levels_to_outputs(lg, bx, torch.from_numpy)
        total, losses = L.get_losses(out, torch.from_numpy(tb), torch.from_numpy(tc), int(bg))
        assert list(losses.keys()) == [str(k) for k in fx[f"c{ci}_keys"]]
        got = np.array([float(v) for v in losses.values()])
        assert np.allclose(got, fx[f"c{ci}_losses"], rtol=2e-6, atol=2e-6)
        assert abs(float(total) - float(fx[f"c{ci}_total"])) <= 2e-6 * abs(float(fx[f"c{ci}_total"]))
        Lv, B = lg.shape[0], lg.shape[1]
        for lv in range(Lv):
            for b in range(B):
                ti, pi, sel, _, _ = L.hungarian_matching(torch.from_numpy(tb[b]), torch.from_numpy(tc[b]), torch.from_numpy(bx[lv, b]),
                                                         torch.from_numpy(lg[lv, b]))
                n = int(tb[b, 0, 0])
                assert np.array_equal(ti.numpy(), fx[f"c{ci}_t_idx"][lv, b, :n]) and np.array_equal(pi.numpy(), fx[f"c{ci}_p_idx"][lv, b, :n])
        for fmt in ("xy_center", "xyxy", "yxyx"):
            b_, l_, s_ = L.get_model_inference(out, int(bg), fmt)
            assert np.array_equal(l_.numpy(), fx[f"c{ci}_inf_{fmt}_labels"])
            assert np.allclose(b_.numpy(), fx[f"c{ci}_inf_{fmt}_boxes"], rtol=1e-6, atol=1e-7)
            assert np.allclose(s_.numpy(), fx[f"c{ci}_inf_{fmt}_scores"], rtol=1e-6, atol=1e-7)


def test_oracle_forward_equals_reference_code_outputs():
    from oracle import detr_ref as R, dropout_ref as DR
    fx = _load("refpy_forward.npz")
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    params = R.make_params(int(fx["top_seed"]))
    assert _hash(params) == str(fx["top_hash"]), "numpy's Generator stream changed: regenerate the fixtures"
    with torch.no_grad():
        ref = R.detr_forward(torch.from_numpy(fx["top_images"]), R.to_torch(params))
    lg = torch.stack([a["pred_logits"] for a in ref["aux"]] + [ref["pred_logits"]])
    bx = torch.stack([a["pred_boxes"] for a in ref["aux"]] + [ref["pred_boxes"]])
    assert _rel(lg, fx["top_logits"]) < 1e-5 and _rel(bx, fx["top_boxes"]) < 1e-5
    p2 = R.make_params(int(fx["drop_seed"]), num_enc=2, num_dec=2)
    assert _hash(p2) == str(fx["drop_hash"])
    with torch.no_grad():
        r2 = R.detr_forward(torch.from_numpy(fx["drop_images"]), R.to_torch(p2), num_enc=2, num_dec=2,
                            drop=DR.Dropper(0.1, int(fx["drop_step_seed"])))
    lg2 = torch.stack([a["pred_logits"] for a in r2["aux"]] + [r2["pred_logits"]])
    assert _rel(lg2, fx["drop_logits"]) < 1e-5
    p3 = R.make_params(int(fx["ft_seed"]), num_enc=1, num_dec=6, nb_class=4)
    with torch.no_grad():
        r3 = R.detr_forward(torch.from_numpy(fx["ft_images"]), R.to_torch(p3), num_enc=1, num_dec=6)
    assert _rel(torch.stack([a["pred_boxes"] for a in r3["aux"]] + [r3["pred_boxes"]]), fx["ft_boxes"]) < 1e-5


def test_variable_partition_equals_reference_code_partition():
    """optimizers.py:10-64 as executed by the reference code: which tensor belongs to which Adam."""
    from detr_tf.params import GROUPS, variable_group
    fx = _load("refpy_training.npz")
    for names, groups, nlayers in ((fx["names"], fx["groups"], []), (fx["ft_names"], fx["ft_groups"], ["cls_layer", "pos_layer"])):
        for k, g in zip(names, groups):
            assert GROUPS[variable_group(str(k), nlayers)] == str(g), (k, g)


# ======================================================================================================
# GPU: the HIP path against the reference-code outputs
# ======================================================================================================
def _cfg(bg=91):
    from detr_tf.training_config import TrainingConfig
    cfg = TrainingConfig()
    cfg.background_class = bg
    cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
    cfg.target_batch = None
    cfg.batch_size = 2
    return cfg


@pytest.mark.gpu
def test_hip_set_loss_matching_inference_vs_reference_code_outputs(hip):
    from detr_tf.inference import get_model_inference
    from detr_tf.loss.loss import get_losses
    fx = _load("refpy_setloss.npz")
    dev = "cuda:0"
    for ci in range(int(fx["n_cases"])):
        lg, bx, tb, tc, bg = (fx[f"c{ci}_{n}"] for n in ("logits", "boxes", "t_bbox", "t_class", "bg"))
        out = _levels_to_outputs(lg, bx, lambda a: torch.from_numpy(a).to(dev))
        cfg = _cfg(int(bg))
        cfg.check_matching = True
        total, log = get_losses(out, tb, tc, cfg)
        assert list(log.keys()) == [str(k) for k in fx[f"c{ci}_keys"]]                     # same keys, same ORDER (main, then _0.._4)
        got = np.array([float(v) for v in log.values()])
        assert np.allclose(got, fx[f"c{ci}_losses"], rtol=2e-5, atol=2e-6), (ci, got, fx[f"c{ci}_losses"])
        assert abs(float(total) - float(fx[f"c{ci}_total"])) <= 1e-5 * abs(float(fx[f"c{ci}_total"]))
        # matched pairs per (level, image): the reference's (t_indices, p_indices) after the double swap
        from detr_tf.loss.loss import SetLoss
        Lv, B, Q, C = lg.shape
        sl = SetLoss.get(Lv, B, Q, C, tb.shape[1], torch.device(dev))
        pft = sl.matcher.pred_for_tgt.view(Lv, B, -1).cpu().numpy()
        for lv in range(Lv):
            for b in range(B):
                n = int(tb[b, 0, 0])
                t_idx, p_idx = fx[f"c{ci}_t_idx"][lv, b, :n], fx[f"c{ci}_p_idx"][lv, b, :n]
                assert np.array_equal(pft[lv, b, t_idx], p_idx), (ci, lv, b)
        for fmt in ("xy_center", "xyxy", "yxyx"):
            b_, l_, s_ = get_model_inference(out, int(bg), fmt)
            assert np.array_equal(l_.cpu().numpy(), fx[f"c{ci}_inf_{fmt}_labels"])
            assert np.allclose(b_.cpu().numpy(), fx[f"c{ci}_inf_{fmt}_boxes"], rtol=1e-6, atol=1e-7)
            assert np.allclose(s_.cpu().numpy(), fx[f"c{ci}_inf_{fmt}_scores"], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_hip_forward_three_modes_vs_reference_code_outputs(hip):
    from detr_tf.networks.detr import get_detr_model
    from oracle import detr_ref as R
    fx = _load("refpy_forward.npz")

    def levels(out):
        return (torch.stack([a["pred_logits"] for a in out["aux"]] + [out["pred_logits"]]),
                torch.stack([a["pred_boxes"] for a in out["aux"]] + [out["pred_boxes"]]))

    cfg = _cfg()
    m = get_detr_model(cfg, include_top=True)
    assert not m.load_weights(R.make_params(int(fx["top_seed"])))
    lg, bx = levels(m(fx["top_images"], training=False))
    assert _rel(lg, fx["top_logits"]) < 2e-4 and _rel(bx, fx["top_boxes"]) < 2e-4
    # training=True: the dropout sites / order are the reference code's, the bits the shared counter hash
    m2 = get_detr_model(cfg, include_top=True, num_encoder_layers=2, num_decoder_layers=2)
    assert not m2.load_weights(R.make_params(int(fx["drop_seed"]), num_enc=2, num_dec=2))
    lg2, bx2 = levels(m2(fx["drop_images"], training=True))
    assert m2.engine._drop[1] == int(fx["drop_step_seed"])
    assert _rel(lg2, fx["drop_logits"]) < 2e-4 and _rel(bx2, fx["drop_boxes"]) < 2e-4
    # finetune heads and the headless mode
    cfg3 = _cfg()
    m3 = get_detr_model(cfg3, include_top=False, nb_class=4, num_encoder_layers=1, num_decoder_layers=6)
    p3 = R.make_params(int(fx["ft_seed"]), num_enc=1, num_dec=6, nb_class=4)
    assert not m3.load_weights(p3) and cfg3.nlayers == ["cls_layer", "pos_layer"]
    lg3, bx3 = levels(m3(fx["ft_images"], training=False))
    assert _rel(lg3, fx["ft_logits"]) < 2e-4 and _rel(bx3, fx["ft_boxes"]) < 2e-4
    m4 = get_detr_model(_cfg(), include_top=False, num_encoder_layers=1, num_decoder_layers=6)
    m4.load_weights({k: v for k, v in p3.items() if not k.startswith(("cls_layer", "pos_layer"))})
    assert _rel(m4(fx["ft_images"], training=False), fx["ft_hs"]) < 2e-4


@pytest.mark.gpu
def test_hip_training_steps_vs_reference_code_outputs(hip, capsys):
    """The reference's run_train_step + aggregate_grad_and_apply (dropout on, masks shared): per-tensor gradient norms of
    both steps, the parameters after 2 plain steps and after 4 steps with target_batch = 4, and fit()'s console line."""
    from detr_tf import training
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.optimizers import setup_optimizers
    from oracle import detr_ref as R
    fx = _load("refpy_training.npz")
    names = [str(k) for k in fx["names"]]
    small = [str(k) for k in fx["small_names"]]
    params = R.make_params(int(fx["seed"]), num_enc=1, num_dec=2)
    assert _hash(params) == str(fx["hash"])
    batches = [(fx[f"images{i}"], fx[f"t_bbox{i}"], fx[f"t_class{i}"]) for i in range(4)]

    def fresh(target_batch):
        cfg = _cfg()
        cfg.target_batch = target_batch
        cfg.use_graph = False
        model = get_detr_model(cfg, include_top=True, num_encoder_layers=1, num_decoder_layers=2)
        assert not model.load_weights(params)
        return cfg, model, setup_optimizers(model, cfg)

    # ---- two plain steps
    cfg, model, opt = fresh(None)
    loss_keys = [str(k) for k in fx["plain_loss_keys"]]
    for s in range(2):
        im, tb, tc = batches[s]
        out, total, log, steps = training.run_train_step(model, im, tb, tc, opt, cfg)
        gn = np.array([float(model.engine.P.gviews[k].double().norm()) for k in names])
        ref = fx["plain_grad_norms"][s]
        bad = [(k, a, b) for k, a, b in zip(names, gn, ref) if abs(a - b) > 1e-2 * b + 1e-6 * ref.max()]
        assert not bad, bad[:8]
        assert list(log.keys()) == loss_keys
        got = np.array([float(log[k]) for k in loss_keys])
        assert np.allclose(got, fx["plain_losses"][s], rtol=1e-3, atol=1e-5), (got, fx["plain_losses"][s])
        for name in steps:
            training.aggregate_grad_and_apply(name, opt, steps[name]["gradients"], s, cfg)
    after = {k: model.engine.P.views[k].detach().cpu().numpy() for k in names}
    dn = np.array([float(np.linalg.norm((after[k] - params[k]).astype(np.float64))) for k in names])
    assert np.allclose(dn, fx["plain_delta_norms"], rtol=2e-2), [(k, a, b) for k, a, b in zip(names, dn, fx["plain_delta_norms"]) if abs(a - b) > 2e-2 * b][:8]
    for k in small:
        d_ref, d_got = fx["plain_after/" + k] - params[k], after[k] - params[k]
        assert np.linalg.norm(d_got - d_ref) <= 5e-2 * np.linalg.norm(d_ref) + 1e-12, k
    # ---- gradient accumulation through fit(): apply after batches 1 and 3
    cfg, model, opt = fresh(4)
    training.fit(model, batches, opt, cfg, epoch_nb=3, class_names=[])
    after = {k: model.engine.P.views[k].detach().cpu().numpy() for k in names}
    dn = np.array([float(np.linalg.norm((after[k] - params[k]).astype(np.float64))) for k in names])
    assert np.allclose(dn, fx["accum_delta_norms"], rtol=2e-2)
    assert cfg.global_step == 4
    # ---- console line of fit() on 3 plain batches: character for character up to the time field
    capsys.readouterr()
    cfg, model, opt = fresh(None)
    training.fit(model, batches[:3], opt, cfg, epoch_nb=3, class_names=[])
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    ref_lines = [str(l) for l in fx["fit_stdout"]]
    assert len(lines) == len(ref_lines) == 1
    assert lines[0].split("time :")[0] == ref_lines[0].split("time :")[0], (lines[0], ref_lines[0])
