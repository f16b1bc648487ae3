"""Reports the deviation of precision="bf16" (bf16 MFMA, bf16 weight shadow, bf16 backbone activation storage) from the
exact-fp32 HIP path on the same weights / batch at the bench size: forward logits / boxes, set loss, one training step's
gradients.  Run on the GPU box: python scripts/bf16_deviation.py [batch]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = TrainingConfig()
cfg.background_class = 91
cfg.batch_size = B
cfg.target_batch = None
cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
images = torch.from_numpy(np.random.default_rng(1234).normal(size=(B, 800, 1333, 3)).astype(np.float32)).cuda()
tb, tc = make_targets(B, np.random.default_rng(1235))
tb, tc = torch.from_numpy(tb).cuda(), torch.from_numpy(tc).cuda()
res = {}
for prec in ("fp32", "bf16"):
    model = get_detr_model(cfg, include_top=True, device="cuda:0", seed=0, dropout=0.0, precision=prec)
    opt = setup_optimizers(model, cfg)
    out, total, log, steps = training.run_train_step(model, images, tb, tc, opt, cfg)
    torch.cuda.synchronize()
    res[prec] = dict(logits=out["pred_logits"].float().cpu(), boxes=out["pred_boxes"].float().cpu(), loss=float(total),
                     grad=model.engine.P.grad.detach().cpu().clone(), offsets=dict(model.engine.P.offsets))
    del model, opt
    torch.cuda.empty_cache()
a, b = res["fp32"], res["bf16"]
rel = lambda x, y: float((x - y).abs().max() / (y.abs().max() + 1e-30))
print(f"logits max-abs dev / scale: {rel(b['logits'], a['logits']):.3e}   boxes: {rel(b['boxes'], a['boxes']):.3e}")
print(f"loss fp32 {a['loss']:.6f}  bf16 {b['loss']:.6f}  rel dev {abs(a['loss'] - b['loss']) / abs(a['loss']):.3e}")
l2 = []
for name, (o, n) in a["offsets"].items():
    ra, rb = a["grad"][o:o + n].double(), b["grad"][o:o + n].double()
    if float(ra.norm()) > 1e-10:
        l2.append(float((rb - ra).norm() / ra.norm()))
l2 = np.array(sorted(l2))
print(f"per-tensor gradient relative L2 dev: median {np.median(l2):.3e}  p90 {l2[int(0.9 * len(l2))]:.3e}  max {l2[-1]:.3e}  (n={len(l2)})")
print(f"whole-gradient relative L2 dev: {float((b['grad'].double() - a['grad'].double()).norm() / a['grad'].double().norm()):.3e}")
