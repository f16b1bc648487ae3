"""Weight import: PyTorch DETR state-dicts -> the reference's layer names (detr_tf/networks/weights.py),
round trips, and -- with mapped weights -- an INDEPENDENT implementation (HuggingFace DetrForObjectDetection, random init) as
a second opinion on the network side of the oracle (CPU) and of the HIP path (GPU)."""
import numpy as np
import pytest
import torch


def _hf_model(seed=0):
    transformers = pytest.importorskip("transformers")
    torch.manual_seed(seed)
    cfg = transformers.DetrConfig(use_timm_backbone=False, use_pretrained_backbone=False,
                                  backbone_config=transformers.ResNetConfig(out_features=["stage4"]), num_labels=91,
                                  auxiliary_loss=False)
    model = transformers.DetrForObjectDetection(cfg).eval()
    with torch.no_grad():      # non-trivial frozen-BN statistics / LayerNorm vectors, activations O(1)
        for name, buf in model.named_buffers():
            if name.endswith("running_var"):
                buf.uniform_(0.5, 1.5)
            elif name.endswith("running_mean"):
                buf.normal_(0.0, 0.1)
            elif name.endswith("normalization.weight"):
                buf.uniform_(0.2, 0.4) if ".layer.2." in name else buf.uniform_(0.5, 1.5)
            elif name.endswith("normalization.bias"):
                buf.normal_(0.0, 0.1)
        for name, p in model.named_parameters():
            if name.endswith("layer_norm.bias") or name.endswith("layernorm.bias"):
                p.normal_(0.0, 0.1)
            elif name.endswith("layer_norm.weight") or name.endswith("layernorm.weight"):
                p.uniform_(0.8, 1.2)
    return model


def _hf_levels(model, x_nchw):
    """logits / boxes of all six decoder levels (per-layer states from forward hooks + the shared decoder LayerNorm: HF's own
    auxiliary_loss=True path feeds the NORMED state into the next layer, unlike facebookresearch/detr and the reference)."""
    states = []
    hooks = [l.register_forward_hook(lambda m, i, o: states.append(o[0] if isinstance(o, tuple) else o)) for l in model.model.decoder.layers]
    with torch.no_grad():
        out = model(pixel_values=x_nchw, pixel_mask=torch.ones(x_nchw.shape[0], x_nchw.shape[2], x_nchw.shape[3], dtype=torch.long))
        for h in hooks:
            h.remove()
        inter = [model.model.decoder.layernorm(s) for s in states]
        lg = torch.stack([model.class_labels_classifier(t) for t in inter])
        bx = torch.stack([model.bbox_predictor(t).sigmoid() for t in inter])
    assert float((lg[-1] - out.logits).abs().max()) < 1e-5
    return lg, bx


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()) / float(b.abs().max())


def test_detr_layout_round_trip_and_shapes():
    from detr_tf.networks import weights as W
    from oracle import detr_ref as R
    for blocks in (R.RESNET50_BLOCKS, R.RESNET101_BLOCKS):
        params = R.make_params(3, blocks=blocks, num_enc=2, num_dec=3)
        sd = W.export_state_dict(params)
        assert W.detect_format(sd) == "detr"
        assert sd["backbone.0.body.layer2.0.downsample.0.weight"].shape == (512, 256, 1, 1)
        assert sd["transformer.decoder.layers.2.multihead_attn.in_proj_weight"].shape == (768, 256)
        back = W.convert_state_dict({"model": {k: torch.from_numpy(v) for k, v in sd.items()}})     # checkpoint wrapper + tensors
        assert set(back) == set(params)
        assert all(np.array_equal(back[k], params[k]) for k in params)
    with pytest.raises(KeyError):
        W.export_state_dict(R.make_params(0, num_enc=1, num_dec=1, nb_class=3))      # Keras Dense heads have no PyTorch twin
    with pytest.raises(ValueError):
        W.detect_format({"foo.weight": np.zeros(1)})


def test_oracle_forward_equals_huggingface_detr_on_mapped_weights():
    from detr_tf.networks import weights as W
    from oracle import detr_ref as R
    model = _hf_model()
    params = W.convert_state_dict(model.state_dict())
    want = R.param_shapes()
    assert set(params) == set(want) and all(tuple(params[k].shape) == tuple(want[k]) for k in want)
    torch.set_num_threads(8)
    x = torch.randn(1, 3, 96, 128)
    lg, bx = _hf_levels(model, x)
    with torch.no_grad():
        ours = R.detr_forward(x.permute(0, 2, 3, 1).contiguous(), R.to_torch(params))
    lo = torch.stack([a["pred_logits"] for a in ours["aux"]] + [ours["pred_logits"]])
    bo = torch.stack([a["pred_boxes"] for a in ours["aux"]] + [ours["pred_boxes"]])
    assert _rel(lo, lg) < 1e-4 and _rel(bo, bx) < 1e-4, (_rel(lo, lg), _rel(bo, bx))


@pytest.mark.gpu
def test_hip_forward_equals_huggingface_detr_on_mapped_weights(hip, tmp_path):
    """HIP path vs an independent PyTorch implementation, weights through the N1 mapper and an .npz round trip."""
    from detr_tf.networks import weights as W
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.training_config import TrainingConfig
    model = _hf_model(seed=1)
    path = str(tmp_path / "hf_detr")                      # no suffix: save / load add ".npz" symmetrically
    np.savez(path + ".npz", **W.convert_state_dict(model.state_dict()))
    m = get_detr_model(TrainingConfig(), include_top=True, weights=path + ".npz")
    x = torch.randn(2, 3, 160, 224)
    lg, bx = _hf_levels(model, x)
    out = m(x.permute(0, 2, 3, 1).contiguous().numpy(), training=False)
    lo = torch.stack([a["pred_logits"] for a in out["aux"]] + [out["pred_logits"]])
    bo = torch.stack([a["pred_boxes"] for a in out["aux"]] + [out["pred_boxes"]])
    assert _rel(lo, lg) < 3e-4 and _rel(bo, bx) < 3e-4, (_rel(lo, lg), _rel(bo, bx))
    m.save_weights(path)
    m2 = get_detr_model(TrainingConfig(), include_top=True)
    assert not m2.load_weights(path)
    assert torch.equal(m2.engine.P.flat, m.engine.P.flat)
    with pytest.raises(FileNotFoundError, match="detr.ckpt.index"):      # (the reference's own checkpoint: read when the files are there,
        get_detr_model(TrainingConfig(), include_top=True, weights="detr")   #  tests/test_tf_checkpoint.py; never downloaded)


def test_oracle_keras_resnet50_wiring_equals_huggingface_resnet_v1():
    """tf_backbone=True: the oracle's restatement of tf.keras.applications.ResNet50 (third-party; ResNet v1 with the stride on
    the FIRST 1x1 conv of a stage) against an independent implementation of the same wiring -- HuggingFace ResNetModel with
    downsample_in_bottleneck=True (no conv biases there: the oracle's are set to zero; BN eps 1e-5 vs Keras' 1.001e-5)."""
    transformers = pytest.importorskip("transformers")
    from oracle import detr_ref as R
    torch.manual_seed(3)
    cfg = transformers.ResNetConfig(downsample_in_bottleneck=True, out_features=["stage4"])
    m = transformers.ResNetModel(cfg).eval()
    with torch.no_grad():
        for name, buf in m.named_buffers():
            if name.endswith("running_var"):
                buf.uniform_(0.5, 1.5)
            elif name.endswith("running_mean"):
                buf.normal_(0.0, 0.1)
        for name, p in m.named_parameters():
            if name.endswith("normalization.weight"):
                p.uniform_(0.2, 0.4) if ".layer.2." in name else p.uniform_(0.5, 1.5)
            elif name.endswith("normalization.bias"):
                p.normal_(0.0, 0.1)
    sd = m.state_dict()
    P = {}

    def put(src_conv, src_bn, dst):
        P[f"resnet50/{dst}_conv/kernel"] = sd[src_conv + ".weight"].permute(2, 3, 1, 0).contiguous()
        P[f"resnet50/{dst}_conv/bias"] = torch.zeros(sd[src_conv + ".weight"].shape[0])
        for a, b in (("weight", "gamma"), ("bias", "beta"), ("running_mean", "moving_mean"), ("running_var", "moving_variance")):
            P[f"resnet50/{dst}_bn/{b}"] = sd[f"{src_bn}.{a}"]

    put("embedder.embedder.convolution", "embedder.embedder.normalization", "conv1")
    for s_, nb in enumerate((3, 4, 6, 3)):
        for b in range(nb):
            pre, q = f"encoder.stages.{s_}.layers.{b}", f"conv{s_ + 2}_block{b + 1}"
            if b == 0:
                put(f"{pre}.shortcut.convolution", f"{pre}.shortcut.normalization", f"{q}_0")
            for k in range(3):
                put(f"{pre}.layer.{k}.convolution", f"{pre}.layer.{k}.normalization", f"{q}_{k + 1}")
    assert set(P) == {k for k in R.param_shapes(tf_backbone=True) if k.startswith("resnet50/")}
    x = torch.randn(2, 3, 96, 128)
    with torch.no_grad():
        ref = m(x).last_hidden_state                       # [2, 2048, 3, 4]
        ours = R.backbone_tf(x.permute(0, 2, 3, 1).contiguous(), P).permute(0, 3, 1, 2)
    assert ours.shape == ref.shape
    assert _rel(ours, ref) < 1e-4, _rel(ours, ref)
