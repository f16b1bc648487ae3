"""GPU-side diagnostic: per-block backward intermediates of the engine vs oracle autograd."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
import numpy as np, torch
from detr_tf import training
from detr_tf.networks.detr import get_detr_model
from detr_tf.optimizers import setup_optimizers
from detr_tf.training_config import TrainingConfig
from oracle import detr_ref as R, set_loss_ref as L

cfg = TrainingConfig(); cfg.background_class = 91
cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True; cfg.target_batch = None
params = R.make_params(5, num_enc=1, num_dec=2)
model = get_detr_model(cfg, include_top=True, num_encoder_layers=1, num_decoder_layers=2)
model.load_weights(params)
opt = setup_optimizers(model, cfg)
images = np.random.default_rng(2).normal(size=(2, 96, 128, 3)).astype(np.float32)
t_bbox, t_class = L.make_targets(2, seed=30, force_full=False)
training.run_train_step(model, images, t_bbox, t_class, opt, cfg)
torch.cuda.synchronize()
P = R.to_torch(params, requires_grad=True)
taps = {"_blocks": True}
ref_out = R.detr_forward(torch.from_numpy(images), P, num_enc=1, num_dec=2, taps=taps)
tot, _ = L.get_losses(ref_out, torch.from_numpy(t_bbox), torch.from_numpy(t_class), 91)
tot.backward()
eng = model.engine
def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().double()
    d = (a - b).abs()
    return float(d.max()) / (float(b.abs().max()) + 1e-30), int((d > 1e-3 * b.abs().max()).sum()), d
for m in eng._block_meta:
    p = m["p"]
    y1r, y2r = taps[f"{p}:y1"], taps[f"{p}:y2"]
    r1 = rel(m["y1"], y1r); r2 = rel(m["y2"], y2r); rx = rel(m["x"], taps[f"{p}:x"])
    line = f"{p:22s} fwd x {rx[0]:.1e} y1 {r1[0]:.1e} y2 {r2[0]:.1e}"
    k1 = f"scratch:dz1:{m['d1']}:{m['h']}"; k2 = f"scratch:dz2:{m['d1']}:{m['ho']}"
    # engine dz = dL/d(bn out, pre-relu) ; oracle: dL/dy * (y > 0)
    for nm, key, yr in (("dz1", k1, y1r), ("dz2", k2, y2r)):
        if key in eng._bufs and tuple(eng._bufs[key].shape) == tuple(yr.shape):
            ref = yr.grad * (yr.detach() > 0)
            r = rel(eng._bufs[key], ref)
            line += f" | {nm} rel {r[0]:.1e} nbad {r[1]}"
            if r[0] > 1e-3:
                idx = torch.nonzero(r[2] > 1e-3 * ref.abs().max())
                pix = sorted(set((int(i[0]), int(i[1]), int(i[2])) for i in idx))
                line += f" badpix {pix[:12]} (of {len(pix)})"
    print(line)
names = ["backbone/layer3/0/conv1/kernel", "backbone/layer3/0/downsample_0/kernel", "backbone/layer3/0/conv2/kernel"]
for n in names:
    gv = eng.P.gviews[n].cpu().double(); ref = P[n].grad.double()
    d = (gv - ref).abs()
    print(n, "rel", float(d.max() / ref.abs().max()), "n>1e-3", int((d > 1e-3 * ref.abs().max()).sum()), "of", d.numel())
    if d.dim() == 4 and d.shape[0] == 1:
        dd = d[0, 0]
        print("   worst rows(ci)", torch.topk(dd.max(1).values, 5).indices.tolist(), "worst cols(co)", torch.topk(dd.max(0).values, 5).indices.tolist(),
              "row-mean err", float(dd.mean()), )
