"""Run-to-run determinism of the bf16 training step at the BENCH shape (B = 8, 800x1333, dropout 0.1): the same step (no optimiser apply, same
dropout step) repeated N times must give bit-identical forward outputs and gradients.  usage: python scripts/experiments/determinism_full.py [N]
Environment switches (e.g. DETR_HIP_GEMM_RING=2) select the kernels under test."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from detr_tf import training
from detr_tf.networks.detr import get_detr_model
from detr_tf.optimizers import setup_optimizers
from detr_tf.training_config import TrainingConfig
from oracle.set_loss_ref import make_targets

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, H, W = int(os.environ.get("DET_B", 8)), int(os.environ.get("DET_H", 800)), int(os.environ.get("DET_W", 1333))
cfg = TrainingConfig()
cfg.background_class = 91
cfg.batch_size = B
cfg.target_batch = None
cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
model = get_detr_model(cfg, include_top=True, device="cuda:0", seed=0, dropout=0.1, precision=os.environ.get("DET_PREC", "bf16"))
opt = setup_optimizers(model, cfg)
images = torch.from_numpy(np.random.default_rng(1234).normal(size=(B, H, W, 3)).astype(np.float32)).cuda()
tb, tc = make_targets(B, seed=5, force_full=True)
tb, tc = torch.from_numpy(tb).cuda(), torch.from_numpy(tc).cuda()
ref = None
bad = 0
for rep in range(N):
    step0 = model.engine._step_no
    out, total, log, steps = training.run_train_step(model, images, tb, tc, opt, cfg)
    model.engine._step_no = step0                     # the same dropout masks every repetition
    torch.cuda.synchronize()
    cur = (out["pred_logits"].clone(), out["pred_boxes"].clone(), model.engine.P.grad.clone())
    if ref is None:
        ref = cur
        print(f"rep 0: loss {float(total):.6f} |grad| {float(cur[2].abs().sum()):.6e}")
        continue
    nd = [int((a != b).sum()) for a, b in zip(ref, cur)]
    md = float((ref[2] - cur[2]).abs().max())
    print(f"rep {rep}: loss {float(total):.6f}  differing entries: logits {nd[0]} boxes {nd[1]} grads {nd[2]} (max |d grad| {md:.3e})")
    bad += sum(nd) > 0
    if nd[2] and rep == 1:                            # which parameters: name -> differing entries
        diff = (ref[2] != cur[2])
        for name, (o, n) in model.engine.P.offsets.items():
            c = int(diff[o:o + n].sum())
            if c:
                print(f"    {name}: {c} of {n} entries differ, max |d| {float((ref[2][o:o + n] - cur[2][o:o + n]).abs().max()):.3e}, max |g| {float(ref[2][o:o + n].abs().max()):.3e}")
print("DETERMINISTIC" if bad == 0 else f"NOT deterministic: {bad} of {N - 1} repetitions differ")
