"""Optimiser side of the drop-in API (reference detr_tf/optimizers.py:10-163).

Three Adam(clipnorm) optimisers -- backbone / transformers / nlayers -- over disjoint groups of a
flat fp32 parameter buffer; per-TENSOR clip-by-norm and the Adam update are two HIP launches per
group (csrc/optim.hip).  Gradients come from the engine's hand-written backward.
"""
import math
from ctypes import c_float

import torch

from . import _hip as hip
from .params import GROUPS

CHUNK = 8192


class GroupAdam:
    """tf.keras.optimizers.Adam(learning_rate=<callable>, clipnorm=...) for one variable group
    (optimizers.py:86-88): beta1 .9, beta2 .999, epsilon 1e-7, bias-corrected step size."""

    def __init__(self, store, group_id, lr_getter, clipnorm, nlayers, engine=None):
        self.store, self.group_id, self.lr_getter, self.clipnorm = store, group_id, lr_getter, clipnorm
        self.engine = engine          # its derived weight copies (BN-folded kernels, bf16 shadow) go stale with every apply
        self.iterations = 0
        self.beta_1, self.beta_2, self.epsilon = 0.9, 0.999, 1e-7
        dev = store.device
        tables = store.build_tables(nlayers, CHUNK)
        grp = tables["group_host"]
        keep = [i for i, t in enumerate(tables["chunk_tensor"].tolist()) if grp[t] == group_id]
        self.names = [n for n, g in zip(tables["names"], grp) if g == group_id]
        self.n_chunks = len(keep)
        idx = torch.tensor(keep, dtype=torch.int64, device=dev)
        self.chunk_tensor = tables["chunk_tensor"][idx].contiguous() if keep else None
        self.chunk_start = tables["chunk_start"][idx].contiguous() if keep else None
        self.seg_end = tables["seg_end"]
        self.tensor_group = torch.tensor(grp, dtype=torch.int32, device=dev)
        self.sumsq = torch.zeros(max(1, self.n_chunks), dtype=torch.float32, device=dev)     # one partial sum of squares per chunk
        if not hasattr(store, "adam_m"):            # the groups own disjoint slices of shared moment buffers
            store.adam_m = torch.zeros_like(store.flat)
            store.adam_v = torch.zeros_like(store.flat)
        self.m, self.v = store.adam_m, store.adam_v
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)

    def learning_rate(self):
        return float(self.lr_getter())

    def set_step_hyper(self):
        """Advances the Adam step counter and writes this step's hyper-parameters (bias-corrected step size, clipnorm,
        betas, epsilon) to DEVICE memory.  Separate from the kernels that consume them, so that a captured training
        step (training.GraphedTrainStep) can be replayed with fresh values."""
        self.iterations += 1
        t = self.iterations
        lr_t = self.learning_rate() * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        clip = self.clipnorm if self.clipnorm is not None else 0.0
        hip.call("detr_hip_set_floats8_f32", self.hyper.data_ptr(), lr_t, lr_t, lr_t, clip, self.beta_1, self.beta_2,
                 self.epsilon, 0.0)

    def apply_gradients(self, grad_flat, hyper_ready=False):
        """One Keras Adam step on this group's tensors using the flat gradient buffer."""
        if not hyper_ready:
            self.set_step_hyper()
        if self.engine is not None:
            self.engine.bump_weights_version()
        if self.n_chunks == 0:
            return
        hip.call("detr_hip_sumsq_segments_f32", grad_flat.data_ptr(), self.chunk_tensor.data_ptr(),
                 self.chunk_start.data_ptr(), self.seg_end.data_ptr(), self.n_chunks, CHUNK, self.sumsq.data_ptr())
        hip.call("detr_hip_clip_adam_f32", self.store.flat.data_ptr(), grad_flat.data_ptr(), self.m.data_ptr(),
                 self.v.data_ptr(), self.chunk_tensor.data_ptr(), self.chunk_start.data_ptr(), self.seg_end.data_ptr(),
                 self.tensor_group.data_ptr(), self.sumsq.data_ptr(), self.hyper.data_ptr(), self.n_chunks, CHUNK)


def setup_optimizers(model, config):
    """optimizers.py:67-107: returns the dict the training loop threads through."""
    store = model.engine.P
    out = {}
    for gid, name in enumerate(GROUPS):
        # resolved at every step: `config.backbone_lr = 1e-4` (rebinding) works like `.assign(1e-4)` on the cell
        getter = (lambda n=name: float(getattr(config, f"{n}_lr")))
        opt = GroupAdam(store, gid, getter, config.gradient_norm_clipping, config.nlayers, engine=model.engine)
        out[f"{name}_optimizer"] = opt
        out[f"{name}_variables"] = [store.views[n] for n in opt.names]
    out["_store"] = store
    out["_engine"] = model.engine
    return out


class GroupGradients:
    """Handle on one group's slice of the flat gradient buffer (what the reference's
    gradient_steps[name]["gradients"] list is)."""

    def __init__(self, store, opt):
        self.store, self.opt = store, opt

    def tensors(self):
        return [self.store.gviews[n] for n in self.opt.names]


def gather_gradient(model, optimizers, total_loss, m_outputs, config, log, loss_scale=1.0):
    """optimizers.py:110-133.  The reference calls tape.gradient here; this runs the set-loss
    gradient kernel and the engine's backward into the (zeroed) flat gradient buffer."""
    eng = model.engine
    eng.zero_grad()
    d_logits, d_boxes = m_outputs.set_loss.grad(loss_scale)
    need_backbone = bool(getattr(config, "train_backbone", True))
    dp = model.dp
    eng.backward(d_logits, d_boxes, backbone=need_backbone, on_bucket=(dp.on_bucket if dp is not None else None))
    if dp is not None:
        dp.finish()
    steps = {}
    for name in GROUPS:
        steps[name] = {"gradients": GroupGradients(eng.P, optimizers[f"{name}_optimizer"])}
        log[f"{name}_lr"] = optimizers[f"{name}_optimizer"].learning_rate()
    return steps


def aggregate_grad_and_apply(name, optimizers, gradients, step, config):
    """optimizers.py:137-163: zero the accumulator at step % agg == 0, add this step's gradient,
    apply at (step+1) % agg == 0; skipped when config.train_<name> is falsy."""
    gradient_aggregate = None
    if config.target_batch is not None:
        gradient_aggregate = int(config.target_batch // config.batch_size)
    if not getattr(config, f"train_{name}"):
        return
    store = optimizers["_store"]
    opt = optimizers[f"{name}_optimizer"]
    gname = f"{name}_gradients"
    if gradient_aggregate is None or gradient_aggregate <= 1:
        optimizers[gname] = store.grad
        opt.apply_gradients(store.grad)
        model_dirty(optimizers)
        return
    acc = optimizers.get(gname)
    if not isinstance(acc, torch.Tensor) or acc is store.grad:
        acc = optimizers[gname] = torch.zeros_like(store.grad)
    elif step % gradient_aggregate == 0:
        hip.zero_(acc)
    # only this group's tensors are accumulated (the others are untouched slices of the same buffer)
    for n in opt.names:
        o, cnt = store.offsets[n]
        hip.call("detr_hip_axpy_f32", acc.data_ptr() + 4 * o, store.grad.data_ptr() + 4 * o, c_float(1.0), cnt)
    if (step + 1) % gradient_aggregate == 0:
        opt.apply_gradients(acc)
        model_dirty(optimizers)


def model_dirty(optimizers):
    eng = optimizers.get("_engine")
    if eng is not None:
        eng.bump_weights_version()
