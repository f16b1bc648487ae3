"""Training / evaluation loop of the drop-in API (reference detr_tf/training.py:9-87)."""
import time

import torch

from .loss.loss import get_losses
from .optimizers import aggregate_grad_and_apply, gather_gradient


def _gradient_aggregate(config):
    if config.target_batch is not None:
        return max(1, int(config.target_batch // config.batch_size))
    return 1


def run_train_step(model, images, t_bbox, t_class, optimizers, config):
    """training.py:9-25: forward (training=True), set loss / gradient_aggregate, gradients."""
    gradient_aggregate = _gradient_aggregate(config)
    optimizers["_engine"] = model.engine
    m_outputs = model(images, training=True)
    total_loss, log = get_losses(m_outputs, t_bbox, t_class, config)
    total_loss = total_loss / gradient_aggregate
    gradient_steps = gather_gradient(model, optimizers, total_loss, m_outputs, config, log,
                                     loss_scale=1.0 / gradient_aggregate)
    return m_outputs, total_loss, log, gradient_steps


def run_val_step(model, images, t_bbox, t_class, config):
    """training.py:28-32."""
    m_outputs = model(images, training=False)
    total_loss, log = get_losses(m_outputs, t_bbox, t_class, config)
    return m_outputs, total_loss, log


def fit(model, train_dt, optimizers, config, epoch_nb, class_names):
    """Train the model for one epoch (training.py:35-65); same console line every 100 steps."""
    t = None
    for epoch_step, (images, t_bbox, t_class) in enumerate(train_dt):
        m_outputs, total_loss, log, gradient_steps = run_train_step(model, images, t_bbox, t_class, optimizers, config)
        for name in gradient_steps:
            aggregate_grad_and_apply(name, optimizers, gradient_steps[name]["gradients"], epoch_step, config)
        if epoch_step % 100 == 0:
            t = t if t is not None else time.time()
            elapsed = time.time() - t
            print(f"Epoch: [{epoch_nb}], \t Step: [{epoch_step}], \t ce: [{float(log['label_cost']):.2f}] \t "
                  f"giou : [{float(log['giou_loss']):.2f}] \t l1 : [{float(log['l1_loss']):.2f}] \t time : [{elapsed:.2f}]")
            t = time.time()
        config.global_step += 1


def eval(model, valid_dt, config, class_name, evaluation_step=200):
    """Evaluate on the validation set (training.py:68-87)."""
    t = None
    for val_step, (images, t_bbox, t_class) in enumerate(valid_dt):
        m_outputs, total_loss, log = run_val_step(model, images, t_bbox, t_class, config)
        if val_step % 10 == 0:
            t = t if t is not None else time.time()
            elapsed = time.time() - t
            print(f"Validation step: [{val_step}], \t ce: [{float(log['label_cost']):.2f}] \t "
                  f"giou : [{float(log['giou_loss']):.2f}] \t l1 : [{float(log['l1_loss']):.2f}] \t time : [{elapsed:.2f}]")
        if val_step + 1 >= evaluation_step:
            break
