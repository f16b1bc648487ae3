"""ORACLE (test infrastructure only) -- CPU restatement of the reference's optimiser step.

Follows detr_tf/optimizers.py (variable partition :10-64, three Keras Adam(clipnorm) :86-88,
accumulate/apply cadence :137-163) and the semantics of the third-party Keras/TF ops those
lines call (tf.clip_by_norm; Keras OptimizerV2 Adam, beta1 .9, beta2 .999, epsilon 1e-7,
bias-corrected step size, no weight decay).  PARITY UNPINNED (no reference tests; TF absent).
"""
import math

import numpy as np


def variable_group(name, nlayers=("cls_layer", "pos_layer")):
    """optimizers.py:10-43: 'backbone' = every layer of the inner model "detr" except the
    transformer (ResNet convs + input_proj + query_embed); 'transformers' = transformer vars +
    outer layers not listed in config.nlayers (class_embed / bbox_embed_* with include_top);
    'nlayers' = the layers named in config.nlayers."""
    top = name.split("/", 1)[0]
    if top in nlayers:
        return "nlayers"
    if top == "transformer" or top in ("class_embed", "bbox_embed_0", "bbox_embed_1", "bbox_embed_2"):
        return "transformers"
    return "backbone"


def clip_by_norm(g, clip):
    """tf.clip_by_norm: g * clip / max(||g||_2, clip)."""
    n = math.sqrt(float((g.astype(np.float64) ** 2).sum()))
    return (g * (clip / max(n, clip))).astype(g.dtype), n


class Adam:
    """Keras OptimizerV2 Adam applied to a list of tensors with per-tensor clipnorm."""

    def __init__(self, lr, clipnorm=0.1, beta1=0.9, beta2=0.999, eps=1e-7):
        self.lr, self.clipnorm, self.b1, self.b2, self.eps = lr, clipnorm, beta1, beta2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def apply(self, grads, params):
        """grads/params: dict name -> np.ndarray (params updated in place)."""
        self.t += 1
        lr = float(self.lr() if callable(self.lr) else self.lr)
        lr_t = lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, g in grads.items():
            if g is None:
                continue
            if self.clipnorm is not None:
                g, _ = clip_by_norm(g, self.clipnorm)
            m = self.m.setdefault(k, np.zeros_like(params[k]))
            v = self.v.setdefault(k, np.zeros_like(params[k]))
            m[...] = self.b1 * m + (1 - self.b1) * g
            v[...] = self.b2 * v + (1 - self.b2) * g * g
            params[k][...] = params[k] - lr_t * m / (np.sqrt(v) + self.eps)


def aggregate_and_apply(state, name, opt, grads, params, step, gradient_aggregate, train_flag):
    """optimizers.py:137-163 for one group; `state` holds the '<name>_gradients' accumulators."""
    if not train_flag:
        return
    key = f"{name}_gradients"
    if gradient_aggregate is not None and step % gradient_aggregate == 0:
        state[key] = {k: np.zeros_like(params[k]) for k in grads}
    if gradient_aggregate is not None:
        state[key] = {k: (state[key][k] + g) if g is not None else None for k, g in grads.items()}
    else:
        state[key] = grads
    if gradient_aggregate is None or (step + 1) % gradient_aggregate == 0:
        opt.apply(state[key], params)
