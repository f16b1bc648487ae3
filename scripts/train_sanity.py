"""Trains the same model on one fixed synthetic batch for a few steps in the exact-fp32 mode and in precision="bf16"
(bf16 MFMA + weight shadow + bf16 backbone activation storage) and prints both loss curves: evidence that the bf16 mode
optimises like the fp32 mode (same seeds, same dropout masks).  Run on the GPU box: python scripts/train_sanity.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
import numpy as np
import torch

from detr_tf import training
from detr_tf.networks.detr import get_detr_model
from detr_tf.optimizers import setup_optimizers
from detr_tf.training_config import TrainingConfig
from oracle.set_loss_ref import make_targets

B, H, W, STEPS = 4, 384, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = TrainingConfig()
cfg.background_class = 91
cfg.batch_size = B
cfg.target_batch = None
cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
images = torch.from_numpy(np.random.default_rng(7).normal(size=(B, H, W, 3)).astype(np.float32)).cuda()
tb, tc = make_targets(B, seed=8, force_full=False)
tb, tc = torch.from_numpy(tb).cuda(), torch.from_numpy(tc).cuda()
curves = {}
for prec in ("fp32", "bf16"):
    model = get_detr_model(cfg, include_top=True, device="cuda:0", seed=0, dropout=0.1, precision=prec)
    opt = setup_optimizers(model, cfg)
    losses = []
    for i in range(STEPS):
        out, total, log, steps = training.run_train_step(model, images, tb, tc, opt, cfg)
        for name in steps:
            training.aggregate_grad_and_apply(name, opt, steps[name]["gradients"], i, cfg)
        losses.append(float(total))
    curves[prec] = losses
    del model, opt
    torch.cuda.empty_cache()
print("step   fp32-loss   bf16-loss   rel.diff")
for i in range(STEPS):
    a, b = curves["fp32"][i], curves["bf16"][i]
    print(f"{i:4d}  {a:10.5f}  {b:10.5f}  {abs(a - b) / abs(a):.2e}")
print(f"fp32: {curves['fp32'][0]:.4f} -> {curves['fp32'][-1]:.4f}   bf16: {curves['bf16'][0]:.4f} -> {curves['bf16'][-1]:.4f}")
