#!/usr/bin/env python
"""BUILD-CONTAINER ONLY (needs `transformers`; never imported by the package or the GPU box): an INDEPENDENT cross-check of
the network side of the oracle.  HuggingFace's `DetrForObjectDetection` (PyTorch, HF ResNet backbone, random weights with
randomised frozen-BN statistics) is NOT the reference, but it implements the same architecture family; after mapping its
state-dict with detr_tf.networks.weights.convert_state_dict the oracle's forward (oracle/detr_ref.py, a restatement of the
TF reference) must produce HF's logits / boxes on every decoder level.  Known deviations between the TF reference and the
PyTorch lineage are outside the forward graph (box clipping in the loss / post-processing, target header layout, loss
normalisation) or vanish here (no padding mask: the reference feeds an all-False mask, detr.py:172).

Also checks the weight mapper itself: HF layout -> reference names -> facebookresearch/detr layout -> reference names.

  python scripts/crosscheck_hf.py [--size 480 640]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
from detr_tf.networks import weights as W                    # noqa: E402  (pure numpy module of the package)
from oracle import detr_ref as R                               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=[224, 320])
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    from transformers import DetrConfig, DetrForObjectDetection, ResNetConfig
    torch.manual_seed(0)
    cfg = DetrConfig(use_timm_backbone=False, use_pretrained_backbone=False, backbone_config=ResNetConfig(out_features=["stage4"]),
                     num_labels=91, auxiliary_loss=False)
    # (auxiliary_loss=True makes HF's decoder feed the layer-NORMED intermediate state into the next layer -- a deviation of the
    #  HF port from facebookresearch/detr and from the reference (transformer.py:121-125 only appends norm(x)); the per-level
    #  states are taken from forward hooks on the decoder layers instead)
    model = DetrForObjectDetection(cfg).eval()
    with torch.no_grad():      # give the frozen BN layers and LayerNorms non-trivial statistics, keep activations O(1)
        for name, buf in model.named_buffers():
            if name.endswith("running_var"):
                buf.uniform_(0.5, 1.5)
            elif name.endswith("running_mean"):
                buf.normal_(0.0, 0.1)
        for name, p in model.named_parameters():
            if "normalization.weight" in name:
                p.uniform_(0.2, 0.4) if ".layer.2." in name else p.uniform_(0.5, 1.5)
            elif "normalization.bias" in name or name.endswith("layer_norm.bias") or name.endswith("layernorm.bias"):
                p.normal_(0.0, 0.1)
            elif name.endswith("layer_norm.weight") or name.endswith("layernorm.weight"):
                p.uniform_(0.8, 1.2)
            elif name.endswith("query_position_embeddings.weight"):
                p.normal_(0.0, 1.0)
    sd = model.state_dict()
    params = W.convert_state_dict(sd)
    want = R.param_shapes()
    assert set(params) == set(want), (set(want) - set(params), set(params) - set(want))
    for k, shp in want.items():
        assert tuple(params[k].shape) == tuple(shp), (k, params[k].shape, shp)
    # mapper round trip through the facebookresearch/detr layout
    back = W.convert_state_dict(W.export_state_dict(params))
    assert set(back) == set(params) and all(np.array_equal(back[k], params[k]) for k in params)
    H, Wd = args.size
    x = torch.randn(args.batch, 3, H, Wd)
    with torch.no_grad():
        mask = torch.ones(args.batch, H, Wd, dtype=torch.long)
        states = []
        hooks = [l.register_forward_hook(lambda mod, inp, out: states.append(out[0] if isinstance(out, tuple) else out))
                 for l in model.model.decoder.layers]
        hf = model(pixel_values=x, pixel_mask=mask)
        for h in hooks:
            h.remove()
        assert len(states) == 6
        inter = [model.model.decoder.layernorm(s_) for s_ in states]
        hf_aux = [{"logits": model.class_labels_classifier(inter[i]), "pred_boxes": model.bbox_predictor(inter[i]).sigmoid()} for i in range(5)]
        ours = R.detr_forward(x.permute(0, 2, 3, 1).contiguous(), R.to_torch(params))

    def rel(a, b):
        return float((a.double() - b.double()).abs().max()) / float(b.double().abs().max())

    worst = max(rel(ours["pred_logits"], hf.logits), rel(ours["pred_boxes"], hf.pred_boxes))
    for i, aux in enumerate(hf_aux):
        worst = max(worst, rel(ours["aux"][i]["pred_logits"], aux["logits"]), rel(ours["aux"][i]["pred_boxes"], aux["pred_boxes"]))
    print(f"oracle forward vs HuggingFace DetrForObjectDetection ({args.batch}x{H}x{Wd}, 6 levels): max rel deviation {worst:.2e}")
    assert worst < 1e-4, worst
    print("crosscheck_hf: ok")


if __name__ == "__main__":
    main()
