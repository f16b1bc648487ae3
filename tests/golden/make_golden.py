"""Regenerates tests/golden/*.npz from the CPU oracle (restatement of the TF reference -- the
reference itself ships no golden vectors and TensorFlow is not installable).
The fixtures are DATA (seeds, inputs, expected outputs); they pin the Python caller rows of
SURVEY.md 8a (output structure, 36-key loss dict, matched index sets, gradient norms, parameters
after accumulated Adam steps, get_model_inference outputs).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import detr_ref as R, optim_ref as O, set_loss_ref as L  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def tiny_case():
    """1 encoder + 2 decoder layers, 2 images of 64x96, full R50 backbone: a few MB of weights are
    regenerated from the seed, only inputs and expected outputs are stored."""
    seed, num_enc, num_dec, B, H, W = 42, 1, 2, 2, 64, 96
    params = R.make_params(seed, num_enc=num_enc, num_dec=num_dec)
    rng = np.random.default_rng(7)
    images = rng.normal(size=(B, H, W, 3)).astype(np.float32)
    t_bbox, t_class = L.make_targets(B, seed=8, force_full=False)
    P = R.to_torch(params, requires_grad=True)
    out = R.detr_forward(torch.from_numpy(images), P, num_enc=num_enc, num_dec=num_dec)
    total, losses = L.get_losses(out, torch.from_numpy(t_bbox), torch.from_numpy(t_class), 91)
    total.backward()
    matched, costs = [], []
    for lvl in [out] + out["aux"]:
        for b in range(B):
            tbs, tcs = L.strip_header(torch.from_numpy(t_bbox[b]), torch.from_numpy(t_class[b]))
            C = L.cost_matrix(tbs, tcs, lvl["pred_boxes"][b].detach(), lvl["pred_logits"][b].detach()).numpy()
            rows, cols, _ = L.lsap(C)
            m = np.full(100, -1, np.int32)
            m[rows] = cols
            matched.append(m)
            cp = np.zeros((100, 99), np.float32)
            cp[:, :C.shape[1]] = C
            costs.append(cp)
    names = [k for k in params if R.trainable(k)]
    gnorm = np.array([float(P[k].grad.norm()) for k in names], np.float64)
    # two accumulated steps (target_batch // batch_size = 2) then one Adam apply per group
    p2 = {k: v.copy() for k, v in params.items()}
    opts = {g: O.Adam(lr, clipnorm=0.1) for g, lr in (("backbone", 1e-5), ("transformers", 1e-4), ("nlayers", 1e-4))}
    state = {}
    grads = {k: (P[k].grad.numpy() / 2) for k in names}        # loss / gradient_aggregate (training.py:20)
    for step in range(2):
        for g in opts:
            O.aggregate_and_apply(state, g, opts[g], {k: v for k, v in grads.items() if O.variable_group(k) == g}, p2,
                                  step, 2, True)
    upd = {k: (p2[k] - params[k]) for k in ("class_embed/bias", "transformer/decoder/norm/gamma", "input_proj/bias",
                                             "backbone/layer4/2/conv3/kernel")}
    inf = {}
    with torch.no_grad():
        for fmt in ("xy_center", "xyxy", "yxyx"):
            b, l, s = L.get_model_inference(out, 91, fmt)
            inf[f"inf_{fmt}_boxes"], inf[f"inf_{fmt}_labels"], inf[f"inf_{fmt}_scores"] = b.numpy(), l.numpy(), s.numpy()
    np.savez_compressed(
        os.path.join(HERE, "tiny_r50_e1d2_64x96.npz"),
        meta=np.array([seed, num_enc, num_dec, B, H, W]), images=images, t_bbox=t_bbox, t_class=t_class,
        pred_logits=out["pred_logits"].detach().numpy(), pred_boxes=out["pred_boxes"].detach().numpy(),
        aux0_logits=out["aux"][0]["pred_logits"].detach().numpy(), aux0_boxes=out["aux"][0]["pred_boxes"].detach().numpy(),
        loss_keys=np.array(list(losses.keys())), loss_vals=np.array([float(v) for v in losses.values()], np.float64),
        total=np.float64(float(total)), matched=np.stack(matched), costs=np.stack(costs),
        grad_names=np.array(names), grad_norms=gnorm,
        **{"upd_" + k.replace("/", "."): v for k, v in upd.items()}, **inf)


if __name__ == "__main__":
    tiny_case()
    print("wrote", os.listdir(HERE))
