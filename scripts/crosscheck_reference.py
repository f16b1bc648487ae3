#!/usr/bin/env python
"""BUILD-CONTAINER ONLY: runs the reference's OWN Python (from /root/reference) under the torch-backed TensorFlow stand-in
of scripts/tf_shim.py, asserts that oracle/*.py computes the same numbers, and writes DATA-ONLY fixtures
(inputs + the reference code's outputs) to tests/golden/refpy_*.npz for the `-m gpu` replay tests
(tests/test_refpy_fixtures.py).  Nothing of the reference travels: no source text, only arrays.

What it pins (and what not): the shim is the build's code, so this is a CONSISTENCY check of the restatement against the
reference's control flow -- header stripping, the double name swap of hungarian_matching.py:163-203, index offsets,
0.1/1.0 class weights, whole-batch normalisers, 1/2/5 loss weights, aux ordering, padding / stride / dilation choices
of the backbone, scale-after-bias in MultiHeadAttention, sequence-first reshapes, the three output modes of
get_detr_model, the dropout sites and their order, the variable partition of optimizers.py and the accumulate / apply
cadence of aggregate_grad_and_apply -- NOT an independent TensorFlow run (TF is not installable here).

  python scripts/crosscheck_reference.py            # check + (re)write fixtures
"""
import contextlib
import hashlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import tf_shim                                                    # noqa: E402

tf_shim.install()
from tf_shim import STATE, tft                                    # noqa: E402

from detr_tf import bbox as ref_bbox                              # noqa: E402  (the REFERENCE's modules, /root/reference)
from detr_tf import inference as ref_inference                    # noqa: E402
from detr_tf import optimizers as ref_optimizers                  # noqa: E402
from detr_tf import training as ref_training                      # noqa: E402
from detr_tf.data import processing as ref_processing             # noqa: E402
from detr_tf.loss import compute_map as ref_map                    # noqa: E402
from detr_tf.loss import loss as ref_loss                         # noqa: E402
from detr_tf.loss.hungarian_matching import hungarian_matching as ref_hungarian      # noqa: E402
from detr_tf.networks import detr as ref_detr                     # noqa: E402
from detr_tf.training_config import TrainingConfig as RefConfig   # noqa: E402

assert ref_loss.__file__.startswith("/root/reference/"), ref_loss.__file__

from oracle import detr_ref as R, dropout_ref as DR, input_ref as I, optim_ref as O, set_loss_ref as L      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def rel(a, b):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def params_hash(params):
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(k.encode())
        h.update(np.ascontiguousarray(params[k]).tobytes())
    return h.hexdigest()


# =====================================================================================================
# A. set loss + matching + inference post-processing
# =====================================================================================================
def make_set_case(seed, B, Q, C, n_list, levels=6, dup=False):
    rng = np.random.default_rng(seed)
    logits = rng.normal(size=(levels, B, Q, C)).astype(np.float32) * 1.5
    boxes = np.concatenate([rng.uniform(0.05, 0.95, (levels, B, Q, 2)), rng.uniform(0.02, 0.6, (levels, B, Q, 2))], -1).astype(np.float32)
    t_bbox = np.zeros((B, 100, 4), np.float32)
    t_class = np.zeros((B, 100, 1), np.int64)
    for b, n in enumerate(n_list):
        t_bbox[b, 0, 0] = n
        t_bbox[b, 1:1 + n, :2] = rng.uniform(0.2, 0.8, (n, 2))
        t_bbox[b, 1:1 + n, 2:] = rng.uniform(0.05, 0.4, (n, 2))
        t_class[b, 1:1 + n, 0] = rng.integers(1, C - 1, n)
    if dup:      # some predictions sit exactly on targets / outside [0,1] (the clip of bbox.py:180-183 matters)
        boxes[:, 0, :3] = t_bbox[0, 1:4][None]
        boxes[:, :, 5, :] = np.array([0.02, 0.97, 0.3, 0.4], np.float32)
    return logits, boxes, t_bbox, t_class


def m_outputs_from_levels(logits, boxes, wrap):
    Lv = logits.shape[0]
    out = {"pred_logits": wrap(logits[Lv - 1]), "pred_boxes": wrap(boxes[Lv - 1])}
    out["aux"] = [{"pred_logits": wrap(logits[i]), "pred_boxes": wrap(boxes[i])} for i in range(Lv - 1)]
    return out


def check_set_loss():
    cases = [dict(seed=1, B=2, Q=100, C=92, n_list=[7, 3], bg=91),
             dict(seed=2, B=3, Q=100, C=92, n_list=[99, 1, 20], bg=91),
             dict(seed=3, B=2, Q=100, C=5, n_list=[4, 12], bg=0, dup=True),          # finetune-style: few classes, background 0
             dict(seed=4, B=1, Q=100, C=92, n_list=[50], bg=91, levels=1)]           # no aux
    fx = {}
    for ci, c in enumerate(cases):
        logits, boxes, t_bbox, t_class = make_set_case(c["seed"], c["B"], c["Q"], c["C"], c["n_list"], c.get("levels", 6), c.get("dup", False))
        cfg = RefConfig()
        cfg.background_class = c["bg"]
        # ---- the reference's own code
        m_ref = m_outputs_from_levels(logits, boxes, tft)
        if logits.shape[0] == 1:
            m_ref.pop("aux")
        total_ref, losses_ref = ref_loss.get_losses(m_ref, tft(t_bbox), tft(t_class), cfg)
        # ---- the oracle
        m_or = m_outputs_from_levels(logits, boxes, torch.from_numpy)
        if logits.shape[0] == 1:
            m_or.pop("aux")
        total_or, losses_or = L.get_losses(m_or, torch.from_numpy(t_bbox), torch.from_numpy(t_class), c["bg"])
        assert list(losses_ref.keys()) == list(losses_or.keys()), (list(losses_ref), list(losses_or))
        for k in losses_ref:
            assert abs(float(losses_ref[k]) - float(losses_or[k])) <= 2e-6 * max(1.0, abs(float(losses_or[k]))), (ci, k, float(losses_ref[k]), float(losses_or[k]))
        assert abs(float(total_ref) - float(total_or)) <= 2e-6 * abs(float(total_or))
        # ---- matching: the six-tuple of the reference per (level, image) vs the oracle and vs SciPy
        Lv, B = logits.shape[0], logits.shape[1]
        tis, pis = [], []
        for lv in range(Lv):
            for b in range(B):
                six = ref_hungarian(tft(t_bbox[b]), tft(t_class[b]), tft(boxes[lv, b]), tft(logits[lv, b]), slice_preds=True)
                assert len(six) == 6
                t_idx, p_idx, t_sel, p_sel, tb, tc = six            # what the caller (loss.py:118) names them
                oti, opi, osel, otb, otc = L.hungarian_matching(torch.from_numpy(t_bbox[b]), torch.from_numpy(t_class[b]),
                                                                torch.from_numpy(boxes[lv, b]), torch.from_numpy(logits[lv, b]))
                assert np.array_equal(npy(t_idx), npy(oti)) and np.array_equal(npy(p_idx), npy(opi))
                assert np.array_equal(npy(p_sel), npy(osel)) and bool(npy(t_sel).all()) and len(npy(t_sel)) == c["n_list"][b]
                assert np.array_equal(npy(tb), npy(otb)) and np.array_equal(npy(tc), npy(otc))
                pad_t, pad_p = np.full(99, -1, np.int64), np.full(99, -1, np.int64)
                pad_t[:len(npy(t_idx))], pad_p[:len(npy(p_idx))] = npy(t_idx), npy(p_idx)
                tis.append(pad_t)
                pis.append(pad_p)
        fx[f"c{ci}_logits"], fx[f"c{ci}_boxes"], fx[f"c{ci}_t_bbox"], fx[f"c{ci}_t_class"] = logits, boxes, t_bbox, t_class
        fx[f"c{ci}_bg"] = np.int64(c["bg"])
        fx[f"c{ci}_keys"] = np.array(list(losses_ref.keys()))
        fx[f"c{ci}_losses"] = np.array([float(losses_ref[k]) for k in losses_ref], np.float64)
        fx[f"c{ci}_total"] = np.float64(float(total_ref))
        fx[f"c{ci}_t_idx"], fx[f"c{ci}_p_idx"] = np.stack(tis).reshape(Lv, B, 99), np.stack(pis).reshape(Lv, B, 99)
        # ---- inference post-processing on the main level (inference.py:68-95: batch element 0 only)
        for fmt in ("xy_center", "xyxy", "yxyx"):
            rb, rl, rs = ref_inference.get_model_inference(m_ref, c["bg"], bbox_format=fmt)
            ob, ol, os_ = L.get_model_inference(m_or, c["bg"], fmt)
            assert np.array_equal(npy(rl), npy(ol)) and rel(npy(rb), npy(ob)) < 1e-6 and rel(npy(rs), npy(os_)) < 1e-6
            fx[f"c{ci}_inf_{fmt}_boxes"], fx[f"c{ci}_inf_{fmt}_labels"], fx[f"c{ci}_inf_{fmt}_scores"] = npy(rb), npy(rl), npy(rs)
    fx["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(GOLD, "refpy_setloss.npz"), **fx)
    print(f"[A] set loss / matching / inference: reference == oracle on {len(cases)} cases; wrote refpy_setloss.npz")


def check_bbox():
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.uniform(-0.1, 1.1, (13, 2)), rng.uniform(0.01, 0.7, (13, 2))], 1).astype(np.float32)
    b = np.concatenate([rng.uniform(0.0, 1.0, (7, 2)), rng.uniform(0.01, 0.5, (7, 2))], 1).astype(np.float32)
    ra, rb = ref_bbox.xcycwh_to_xy_min_xy_max(tft(a)), ref_bbox.xcycwh_to_xy_min_xy_max(tft(b))
    oa, ob = L.xcycwh_to_xy_min_xy_max(torch.from_numpy(a)), L.xcycwh_to_xy_min_xy_max(torch.from_numpy(b))
    assert np.array_equal(npy(ra), npy(oa))
    iou_r, un_r = ref_bbox.jaccard(ra, rb, return_union=True)
    iou_o, un_o = L.jaccard(oa, ob)
    assert rel(npy(iou_r), npy(iou_o)) < 1e-6 and rel(npy(un_r), npy(un_o)) < 1e-6
    assert np.array_equal(npy(ref_bbox.xcycwh_to_yx_min_yx_max(tft(a))), npy(L.xcycwh_to_yx_min_yx_max(torch.from_numpy(a))))
    print("[A] bbox helpers: reference == oracle")


# =====================================================================================================
# B. network forward (three output modes, eval and training mode)
# =====================================================================================================
def shim_params(params, requires_grad=False):
    return {k: tft(v, requires_grad=requires_grad and R.trainable(k)) for k, v in params.items()}


def drop_hook_for(step_seed_value):
    dropper_cache = {}

    def hook(owner, k, x, scope, rate):
        kind, idx = owner.split("/")[-2], int(owner.rsplit("_", 1)[1])
        site = (16 * idx if kind == "encoder" else 16 * (32 + idx)) + k
        layout = "attn" if scope[-1] in ("self_attn", "multihead_attn") else "lbc"
        d = dropper_cache.setdefault(rate, DR.Dropper(rate, step_seed_value))
        return d(site, x, layout)
    return hook


class ReplayModel:
    """The reference's functional model re-built on every call (tf_shim executes the functional API eagerly on a concrete
    batch); the variables are the SAME tensor objects across re-builds, so optimizers.py sees a persistent model."""

    def __init__(self, P, first_batch, step_seed_fn=None, **kw):
        self.P, self.kw, self.step_seed_fn, self.calls = P, kw, step_seed_fn, 0
        self.config = kw.pop("config")
        self._build(first_batch, False)

    def _build(self, images, training):
        hook = None
        if training and self.step_seed_fn is not None:
            hook = drop_hook_for(self.step_seed_fn(self.calls))
        tf_shim.reset(params=self.P, input=images, training=training, drop_hook=hook)
        self.cur = ref_detr.get_detr_model(self.config, **self.kw)
        return self.cur

    def __call__(self, images, training=False):
        images = images if isinstance(images, tf_shim.TFTensor) else tft(images)
        if training:
            self.calls += 1
        m = self._build(images, training)
        return m(images)

    def get_layer(self, name):
        return self.cur.get_layer(name)

    @property
    def layers(self):
        return self.cur.layers

    @property
    def name(self):
        return self.cur.name


def check_forward():
    fx = {}
    rng = np.random.default_rng(21)
    # ---- include_top=True, full 6+6, eval
    params = R.make_params(31)
    images = rng.normal(size=(2, 64, 96, 3)).astype(np.float32)
    cfg = RefConfig()
    model = ReplayModel(shim_params(params), tft(images), config=cfg, include_top=True)
    assert model.name == "detr_finetuning" and [l.name for l in model.layers][:2] == ["input", "detr"]
    out = model(images, training=False)
    ref = R.detr_forward(torch.from_numpy(images), R.to_torch(params))
    assert len(out["aux"]) == 5
    worst = max([rel(npy(out["pred_logits"]), npy(ref["pred_logits"])), rel(npy(out["pred_boxes"]), npy(ref["pred_boxes"]))] +
                [rel(npy(out["aux"][i]["pred_logits"]), npy(ref["aux"][i]["pred_logits"])) for i in range(5)] +
                [rel(npy(out["aux"][i]["pred_boxes"]), npy(ref["aux"][i]["pred_boxes"])) for i in range(5)])
    assert worst < 2e-5, worst
    fx["top_seed"], fx["top_hash"], fx["top_images"] = np.int64(31), np.array(params_hash(params)), images
    fx["top_logits"] = np.stack([npy(a["pred_logits"]) for a in out["aux"]] + [npy(out["pred_logits"])])
    fx["top_boxes"] = np.stack([npy(a["pred_boxes"]) for a in out["aux"]] + [npy(out["pred_boxes"])])
    print(f"[B] get_detr_model(include_top=True) eval forward: reference vs oracle max rel {worst:.1e}")
    # ---- the DETR class itself (detr.py:71-92, incl. downsample_masks) on the same weights: main output only
    tf_shim.reset(params=shim_params(params), input=None, training=False)
    d = ref_detr.DETR()
    tf_shim.STATE.scope = []
    # DETR.call runs inside the scope of the model's own name; the checkpoint names have no such prefix -> call the body directly
    o2 = ref_detr.DETR.call(d, (tft(images), torch.zeros(2, 64, 96, dtype=torch.bool)), training=False)
    assert rel(npy(o2["pred_logits"]), npy(ref["pred_logits"])) < 2e-5 and rel(npy(o2["pred_boxes"]), npy(ref["pred_boxes"])) < 2e-5
    print("[B] DETR.call (class form, with downsample_masks): == oracle")
    # ---- training=True with the shared dropout masks (sites and order decided by the reference's control flow)
    seed = DR.step_seed(0x5EED, 1, 0)
    params2 = R.make_params(32, num_enc=2, num_dec=2)
    images2 = rng.normal(size=(2, 64, 96, 3)).astype(np.float32)
    model2 = ReplayModel(shim_params(params2), tft(images2), step_seed_fn=lambda call: seed, config=RefConfig(), include_top=True,
                         num_encoder_layers=2, num_decoder_layers=2)
    out2 = model2(images2, training=True)
    ref2 = R.detr_forward(torch.from_numpy(images2), R.to_torch(params2), num_enc=2, num_dec=2, drop=DR.Dropper(0.1, seed))
    w2 = max(rel(npy(out2["pred_logits"]), npy(ref2["pred_logits"])), rel(npy(out2["pred_boxes"]), npy(ref2["pred_boxes"])),
             rel(npy(out2["aux"][0]["pred_logits"]), npy(ref2["aux"][0]["pred_logits"])))
    assert w2 < 2e-5, w2
    ev = model2(images2, training=False)
    assert rel(npy(ev["pred_logits"]), npy(out2["pred_logits"])) > 1e-3          # dropout really was active
    fx["drop_seed"], fx["drop_hash"], fx["drop_images"], fx["drop_step_seed"] = np.int64(32), np.array(params_hash(params2)), images2, np.int64(seed)
    fx["drop_logits"] = np.stack([npy(a["pred_logits"]) for a in out2["aux"]] + [npy(out2["pred_logits"])])
    fx["drop_boxes"] = np.stack([npy(a["pred_boxes"]) for a in out2["aux"]] + [npy(out2["pred_boxes"])])
    print(f"[B] training=True forward with the shared dropout masks: reference vs oracle max rel {w2:.1e}")
    # ---- include_top=False + nb_class (finetune heads, detr.py:94-114) and the headless mode
    params3 = R.make_params(33, num_enc=1, num_dec=6, nb_class=4)
    sp3 = shim_params(params3)
    images3 = rng.normal(size=(1, 64, 96, 3)).astype(np.float32)
    cfg3 = RefConfig()
    model3 = ReplayModel(sp3, tft(images3), config=cfg3, include_top=False, nb_class=4, num_encoder_layers=1, num_decoder_layers=6)
    assert cfg3.nlayers == ["cls_layer", "pos_layer"], cfg3.nlayers
    out3 = model3(images3)
    ref3 = R.detr_forward(torch.from_numpy(images3), R.to_torch(params3), num_enc=1, num_dec=6)
    assert len(out3["aux"]) == 5
    w3 = max(rel(npy(out3["pred_logits"]), npy(ref3["pred_logits"])), rel(npy(out3["pred_boxes"]), npy(ref3["pred_boxes"])),
             rel(npy(out3["aux"][4]["pred_boxes"]), npy(ref3["aux"][4]["pred_boxes"])))
    assert w3 < 2e-5, w3
    fx["ft_seed"], fx["ft_hash"], fx["ft_images"] = np.int64(33), np.array(params_hash(params3)), images3
    fx["ft_logits"] = np.stack([npy(a["pred_logits"]) for a in out3["aux"]] + [npy(out3["pred_logits"])])
    fx["ft_boxes"] = np.stack([npy(a["pred_boxes"]) for a in out3["aux"]] + [npy(out3["pred_boxes"])])
    hs_model = ReplayModel(sp3, tft(images3), config=RefConfig(), include_top=False, num_encoder_layers=1, num_decoder_layers=6)
    hs = hs_model(images3)
    hs_or = R.detr_hs(torch.from_numpy(images3), R.to_torch(params3), num_enc=1, num_dec=6)
    assert tuple(hs.shape) == (6, 1, 100, 256) and rel(npy(hs), npy(hs_or)) < 2e-5
    fx["ft_hs"] = npy(hs)
    print(f"[B] finetune heads (nb_class=4) and headless mode: reference vs oracle max rel {w3:.1e}")
    np.savez_compressed(os.path.join(GOLD, "refpy_forward.npz"), **fx)
    print("    wrote refpy_forward.npz")


# =====================================================================================================
# C. training step: variable partition, gradients, clipnorm + Adam, accumulate cadence, fit() console
# =====================================================================================================
def group_of_variables(optimizers_dict):
    out = {}
    for g in ("backbone", "transformers", "nlayers"):
        for v in optimizers_dict[f"{g}_variables"]:
            out[v._shim_name] = g
    return out


def check_training():
    fx = {}
    ne, nd = 1, 2
    params = R.make_params(41, num_enc=ne, num_dec=nd)
    rng = np.random.default_rng(42)
    batches = []
    for i in range(4):
        tb, tc = L.make_targets(2, seed=300 + i, force_full=False)
        batches.append((rng.normal(size=(2, 64, 96, 3)).astype(np.float32), tb, tc))
    names = [k for k in params if R.trainable(k)]

    def run_reference(target_batch, n_steps, capture_fit=False):
        P = shim_params(params, requires_grad=True)
        cfg = RefConfig()
        cfg.background_class = 91
        cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
        cfg.batch_size, cfg.target_batch = 2, target_batch
        model = ReplayModel(P, tft(batches[0][0]), step_seed_fn=lambda call: DR.step_seed(0x5EED, call, 0), config=cfg, include_top=True,
                            num_encoder_layers=ne, num_decoder_layers=nd)
        opts = ref_optimizers.setup_optimizers(model, cfg)
        groups = group_of_variables(opts)
        rec = dict(groups=groups, grad_norms=[], losses=[], cfg=cfg)
        if capture_fit:
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ref_training.fit(model, [(tft(im), tft(tb), tft(tc)) for im, tb, tc in batches[:n_steps]], opts, cfg, epoch_nb=3, class_names=[])
            rec["stdout"] = buf.getvalue()
        else:
            for step, (im, tb, tc) in enumerate(batches[:n_steps]):
                m_out, total, log, gsteps = ref_training.run_train_step(model, tft(im), tft(tb), tft(tc), opts, cfg)
                gn = {}
                for g in ("backbone", "transformers", "nlayers"):
                    for v, gr in zip(opts[f"{g}_variables"], gsteps[g]["gradients"]):
                        gn[v._shim_name] = 0.0 if gr is None else float(gr.double().norm())
                rec["grad_norms"].append(gn)
                rec["losses"].append({k: float(v) for k, v in log.items()})
                rec["total"] = float(total)
                for name in gsteps:
                    ref_optimizers.aggregate_grad_and_apply(name, opts, gsteps[name]["gradients"], step, cfg)
        rec["params"] = {k: npy(P[k]).copy() for k in names}
        return rec

    def run_oracle(target_batch, n_steps):
        agg = 1 if target_batch is None else target_batch // 2
        cur = {k: v.copy() for k, v in params.items()}
        opts = {g: O.Adam(lr, clipnorm=0.1) for g, lr in (("backbone", 1e-5), ("transformers", 1e-4), ("nlayers", 1e-4))}
        acc, gnorms = None, []
        for step, (im, tb, tc) in enumerate(batches[:n_steps]):
            P = R.to_torch(cur, requires_grad=True)
            out = R.detr_forward(torch.from_numpy(im), P, num_enc=ne, num_dec=nd, drop=DR.Dropper(0.1, DR.step_seed(0x5EED, step + 1, 0)))
            total, _ = L.get_losses(out, torch.from_numpy(tb), torch.from_numpy(tc), 91)
            (total / agg).backward()
            grads = {k: (P[k].grad.numpy().copy() if P[k].grad is not None else np.zeros_like(cur[k])) for k in names}
            gnorms.append({k: float(np.linalg.norm(grads[k].astype(np.float64))) for k in names})
            if step % agg == 0:
                acc = {k: np.zeros_like(v) for k, v in grads.items()}
            acc = {k: acc[k] + grads[k] for k in names}
            if (step + 1) % agg == 0:
                for g in opts:
                    opts[g].apply({k: v for k, v in acc.items() if O.variable_group(k) == g}, cur)
        return cur, gnorms

    # ---- partition of the variables (optimizers.py:10-64) vs the oracle's rule
    r1 = run_reference(None, 2)
    assert set(r1["groups"]) == set(names), (set(names) ^ set(r1["groups"]))
    for k in names:
        assert r1["groups"][k] == O.variable_group(k), (k, r1["groups"][k], O.variable_group(k))
    n_by = {g: sum(1 for k in names if r1["groups"][k] == g) for g in ("backbone", "transformers", "nlayers")}
    print(f"[C] variable partition reference == oracle: {n_by}")
    # ---- the finetune model (include_top=False, nb_class): cls_layer / pos_layer form the 'nlayers' group
    pf = R.make_params(43, num_enc=1, num_dec=6, nb_class=4)
    cfgf = RefConfig()
    mf = ReplayModel(shim_params(pf, requires_grad=True), tft(batches[0][0]), config=cfgf, include_top=False, nb_class=4,
                     num_encoder_layers=1, num_decoder_layers=6)
    gf = group_of_variables(ref_optimizers.setup_optimizers(mf, cfgf))
    names_f = [k for k in pf if R.trainable(k)]
    assert set(gf) == set(names_f)
    for k in names_f:
        assert gf[k] == O.variable_group(k, tuple(cfgf.nlayers)), (k, gf[k])
    assert sum(1 for k in names_f if gf[k] == "nlayers") == 8
    fx["ft_names"], fx["ft_groups"] = np.array(names_f), np.array([gf[k] for k in names_f])
    print("[C] finetune-model partition (nlayers = cls_layer + pos_layer, 8 tensors): reference == oracle")
    # ---- two plain steps: gradients (per-tensor norms) and the parameters after clipnorm + Adam
    cur, gn = run_oracle(None, 2)
    for s in range(2):
        for k in names:
            a, b = r1["grad_norms"][s][k], gn[s][k]
            assert abs(a - b) <= 2e-3 * max(b, 1e-7) + 1e-9, (s, k, a, b)
    # (Adam divides by sqrt(v): entries whose gradient is at rounding-noise level move by +-lr whatever the noise says, so
    #  the max-abs metric is loose and the relative L2 over each tensor tight)
    def rel_l2(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float(np.linalg.norm(a - b)) / (float(np.linalg.norm(b)) + 1e-30)
    worst = max(rel(r1["params"][k] - params[k], cur[k] - params[k]) for k in names)
    worst_l2 = max(rel_l2(r1["params"][k] - params[k], cur[k] - params[k]) for k in names)
    assert worst < 2e-2 and worst_l2 < 2e-3, (worst, worst_l2)
    print(f"[C] 2 training steps (dropout on, shared masks): gradient norms agree, parameter updates max rel {worst:.1e}")
    fx["names"] = np.array(names)
    fx["groups"] = np.array([r1["groups"][k] for k in names])
    fx["seed"], fx["hash"] = np.int64(41), np.array(params_hash(params))
    for i, (im, tb, tc) in enumerate(batches):
        fx[f"images{i}"], fx[f"t_bbox{i}"], fx[f"t_class{i}"] = im, tb, tc
    fx["plain_grad_norms"] = np.array([[r1["grad_norms"][s][k] for k in names] for s in range(2)])
    fx["plain_delta_norms"] = np.array([float(np.linalg.norm((r1["params"][k] - params[k]).astype(np.float64))) for k in names])
    fx["plain_delta_sums"] = np.array([float((r1["params"][k].astype(np.float64) - params[k]).sum()) for k in names])
    fx["plain_loss_keys"] = np.array(list(r1["losses"][0].keys()))
    fx["plain_losses"] = np.array([[r1["losses"][s][k] for k in r1["losses"][0]] for s in range(2)])
    small = [k for k in names if params[k].size <= 4096]
    fx["small_names"] = np.array(small)
    for k in small:
        fx["plain_after/" + k] = r1["params"][k]
    # ---- gradient accumulation: target_batch = 4 with batch 2 -> apply after steps 1 and 3 (optimizers.py:137-163)
    r2 = run_reference(4, 4)
    cur2, _ = run_oracle(4, 4)
    worst2 = max(rel_l2(r2["params"][k] - params[k], cur2[k] - params[k]) for k in names)
    assert worst2 < 2e-3, worst2
    fx["accum_delta_norms"] = np.array([float(np.linalg.norm((r2["params"][k] - params[k]).astype(np.float64))) for k in names])
    for k in small:
        fx["accum_after/" + k] = r2["params"][k]
    print(f"[C] 4 steps with target_batch=4 (accumulate 2): parameter updates max rel {worst2:.1e}")
    # ---- the console lines of training.fit (training.py:57-60)
    r3 = run_reference(None, 3, capture_fit=True)
    lines = [l for l in r3["stdout"].splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("Epoch: [3], \t Step: [0], \t ce: ["), lines
    assert r3["cfg"].global_step == 3
    fx["fit_stdout"] = np.array(lines)
    print(f"[C] training.fit console line: {lines[0]!r}")
    np.savez_compressed(os.path.join(GOLD, "refpy_training.npz"), **fx)
    print("    wrote refpy_training.npz")


# =====================================================================================================
# D. mAP accumulation (eval.py:30-61 / compute_map.py) -- fixture for the vectorised accumulator of the package
# =====================================================================================================
def make_map_case(seed, n_images, nb_class):
    rng = np.random.default_rng(seed)
    images = []
    for _ in range(n_images):
        n_gt = int(rng.integers(0, 9))
        yx = rng.uniform(0.0, 0.7, (n_gt, 2))
        hw = rng.uniform(0.05, 0.3, (n_gt, 2))
        t_bbox = np.concatenate([yx, np.minimum(yx + hw, 1.0)], 1).astype(np.float32)
        t_cls = rng.integers(0, nb_class, n_gt).astype(np.int64)
        preds, pcls, pscore = [], [], []
        for j in range(n_gt):                       # perturbed copies of the ground truths (some duplicated, some mislabelled)
            for _ in range(int(rng.integers(0, 3))):
                jit = rng.normal(0, 0.03, 4)
                preds.append(np.clip(t_bbox[j] + jit, 0, 1))
                pcls.append(t_cls[j] if rng.uniform() < 0.8 else rng.integers(0, nb_class))
                pscore.append(round(float(rng.uniform(0.05, 1.0)), 2))      # two decimals: score ties do occur
        for _ in range(int(rng.integers(0, 6))):    # false positives
            a = rng.uniform(0, 0.8, 2)
            preds.append(np.concatenate([a, np.minimum(a + rng.uniform(0.05, 0.2, 2), 1.0)]))
            pcls.append(rng.integers(0, nb_class))
            pscore.append(round(float(rng.uniform(0.05, 1.0)), 2))
        images.append((np.asarray(preds, np.float32).reshape(-1, 4), np.asarray(pcls, np.int64), np.asarray(pscore, np.float32), t_bbox, t_cls))
    return images


def check_map():
    fx = {}
    for ci, (seed, n_images, nb_class) in enumerate([(71, 12, 6), (72, 30, 3), (73, 5, 10)]):
        images = make_map_case(seed, n_images, nb_class)
        thresholds = [x / 100.0 for x in range(50, 100, 5)]                   # eval.py:33
        class_names = [f"class_{i}" for i in range(nb_class)]
        ap_data = {"box": [[ref_map.APDataObject() for _ in class_names] for _ in thresholds],
                   "mask": [[ref_map.APDataObject() for _ in class_names] for _ in thresholds]}
        for p_bbox, p_cls, p_score, t_bbox, t_cls in images:                  # eval.py:54 (dummy all-zero masks)
            with np.errstate(divide="ignore", invalid="ignore"):
                ref_map.cal_map(p_bbox, p_cls, p_score, np.zeros((138, 138, len(p_bbox))), t_bbox, t_cls, np.zeros((138, 138, len(t_bbox))),
                                ap_data, thresholds)
        per_class = np.array([[ap_data["box"][t][c].get_ap() if not ap_data["box"][t][c].is_empty() else -1.0 for c in range(nb_class)]
                              for t in range(len(thresholds))], np.float64)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            all_maps = ref_map.calc_map(ap_data, thresholds, class_names, print_result=True)
        fx[f"m{ci}_nb_class"], fx[f"m{ci}_n_images"] = np.int64(nb_class), np.int64(n_images)
        for i, (p_bbox, p_cls, p_score, t_bbox, t_cls) in enumerate(images):
            fx[f"m{ci}_{i}_p_bbox"], fx[f"m{ci}_{i}_p_cls"], fx[f"m{ci}_{i}_p_score"] = p_bbox, p_cls, p_score
            fx[f"m{ci}_{i}_t_bbox"], fx[f"m{ci}_{i}_t_cls"] = t_bbox, t_cls
        fx[f"m{ci}_per_class_ap"] = per_class
        fx[f"m{ci}_box_keys"] = np.array([str(k) for k in all_maps["box"].keys()])
        fx[f"m{ci}_box_vals"] = np.array(list(all_maps["box"].values()), np.float64)
        fx[f"m{ci}_mask_vals"] = np.array(list(all_maps["mask"].values()), np.float64)
        fx[f"m{ci}_table"] = np.array(buf.getvalue())
        print(f"[D] reference cal_map / calc_map on {n_images} images, {nb_class} classes: box mAP {all_maps['box']['all']}, mask mAP {all_maps['mask']['all']}")
    fx["n_cases"] = np.int64(3)
    np.savez_compressed(os.path.join(GOLD, "refpy_map.npz"), **fx)
    print("    wrote refpy_map.npz")


# =====================================================================================================
# E. input stage: normalized_images / pad_labels (data/processing.py) -- the resize is third-party (imgaug / cv2)
# =====================================================================================================
def check_input():
    fx = {}
    rng = np.random.default_rng(81)
    img = rng.integers(0, 256, (2, 37, 53, 3)).astype(np.uint8)
    ramp = np.repeat(np.arange(256, dtype=np.uint8)[:, None, None], 3, 2)             # every pixel value in every channel
    for method in ("torch_resnet", "tf_resnet"):
        cfg = RefConfig()
        cfg.normalized_method = method
        for name, x in (("img", img), ("ramp", ramp)):
            ref = ref_processing.normalized_images(x, cfg)
            assert ref.dtype == np.float32 and np.array_equal(ref, I.normalized_images(x, method))
            fx[f"norm_{method}_{name}"] = ref
    fx["img"], fx["ramp"] = img, ramp
    for ci, n in enumerate((0, 1, 7, 99)):
        tb = rng.uniform(0.1, 0.9, (n, 4)).astype(np.float32)
        tc = rng.integers(1, 91, (n, 1)).astype(np.int64)
        _, rb, rc = ref_processing.pad_labels(None, tft(tb), tft(tc))
        ob, oc = I.pad_labels(tb, tc)
        assert tuple(rb.shape) == (100, 4) and tuple(rc.shape) == (100, 1) and npy(rb).dtype == np.float32 and npy(rc).dtype == np.int64
        assert np.array_equal(npy(rb), ob) and np.array_equal(npy(rc), oc)
        fx[f"pad{ci}_in_bbox"], fx[f"pad{ci}_in_class"], fx[f"pad{ci}_bbox"], fx[f"pad{ci}_class"] = tb, tc, npy(rb), npy(rc)
    fx["n_pad"] = np.int64(4)
    np.savez_compressed(os.path.join(GOLD, "refpy_input.npz"), **fx)
    print("[E] normalized_images (both methods, every pixel value) and pad_labels: reference == oracle; wrote refpy_input.npz")


if __name__ == "__main__":
    check_input()
    check_map()
    check_bbox()
    check_set_loss()
    check_forward()
    check_training()
    print("crosscheck_reference: all checks passed")
