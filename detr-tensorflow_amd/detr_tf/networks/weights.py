"""Weight import / export for the HIP DETR path.

The reference loads a converted TensorFlow checkpoint from GCS (`weights="detr"`, detr_tf/networks/weights.py:5-37) whose
variables carry the Keras layer names of SURVEY.md A.6 -- the names this package's `.npz` format uses.  That checkpoint was
itself produced from the original PyTorch release of DETR, so the route to real weights here is the PyTorch state-dict:

    convert_state_dict(state_dict) -> {reference layer name: np.float32 array}      (then model.load_weights(dict) / np.savez)

understands three source layouts:
  * "detr"      facebookresearch/detr checkpoints (detr-r50-e632da11.pth, key "model"): backbone.0.body.*, transformer.*,
                class_embed, bbox_embed.layers.N, query_embed, input_proj
  * "hf"        HuggingFace `DetrForObjectDetection` with the HF ResNet backbone (use_timm_backbone=False):
                model.backbone[.conv_encoder].model.{embedder,encoder.stages}..., q_proj / k_proj / v_proj / o_proj, [mlp.]fc1/fc2
  * "hf-timm"   HuggingFace DETR with the timm backbone: model.backbone.conv_encoder.model.{conv1,bn1,layerN.M...}
Layout conversions: conv kernels OIHW -> HWIO; Linear kernels stay (out, in) (custom_layers.py:31-54 keeps PyTorch's layout);
q / k / v projections are packed into `in_proj_kernel` (768, 256) rows [Q; K; V] (transformer.py:253-268); LayerNorm
weight / bias -> gamma / beta; frozen-BN vectors keep their four names (custom_layers.py:11-18).

`export_state_dict` is the inverse for the "detr" layout (round-trip tested).  CLI:

    python -m detr_tf.networks.weights detr-r50-e632da11.pth detr-r50.npz
"""
import re
import sys

import numpy as np

BN_LEAVES = ("weight", "bias", "running_mean", "running_var")


def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v), dtype=np.float32)


def _conv(w):
    return np.ascontiguousarray(_np(w).transpose(2, 3, 1, 0))          # OIHW -> HWIO


def detect_format(sd):
    keys = list(sd.keys())
    if any(k.startswith("backbone.0.body.") for k in keys):
        return "detr"
    if any(".embedder.embedder.convolution.weight" in k for k in keys):
        return "hf"
    if any(k.endswith("conv_encoder.model.conv1.weight") for k in keys):
        return "hf-timm"
    raise ValueError("unrecognised DETR state-dict layout (expected facebookresearch/detr or HuggingFace DetrForObjectDetection names)")


def _backbone_torchvision(sd, prefix, out):
    """torchvision-style ResNet names under `prefix` (conv1, bn1, layerN.M.convK / bnK / downsample.{0,1})."""
    out["backbone/conv1/kernel"] = _conv(sd[prefix + "conv1.weight"])
    for leaf in BN_LEAVES:
        out[f"backbone/bn1/{leaf}"] = _np(sd[f"{prefix}bn1.{leaf}"])
    pat = re.compile(re.escape(prefix) + r"layer(\d)\.(\d+)\.(conv\d|bn\d|downsample\.0|downsample\.1)\.(\w+)$")
    for k, v in sd.items():
        m = pat.match(k)
        if not m or m.group(4) == "num_batches_tracked":
            continue
        li, b, what, leaf = m.group(1), m.group(2), m.group(3), m.group(4)
        p = f"backbone/layer{li}/{b}"
        if what.startswith("conv"):
            out[f"{p}/{what}/kernel"] = _conv(v)
        elif what == "downsample.0":
            out[f"{p}/downsample_0/kernel"] = _conv(v)
        elif what == "downsample.1":
            out[f"{p}/downsample_1/{leaf}"] = _np(v)
        else:
            out[f"{p}/{what}/{leaf}"] = _np(v)


def _backbone_hf(sd, prefix, out):
    """HF ResNetModel names under `prefix`: embedder.embedder.*, encoder.stages.S.layers.B.{layer.K,shortcut}.*"""
    out["backbone/conv1/kernel"] = _conv(sd[prefix + "embedder.embedder.convolution.weight"])
    for leaf in BN_LEAVES:
        out[f"backbone/bn1/{leaf}"] = _np(sd[f"{prefix}embedder.embedder.normalization.{leaf}"])
    pat = re.compile(re.escape(prefix) + r"encoder\.stages\.(\d)\.layers\.(\d+)\.(layer\.(\d)|shortcut)\.(convolution|normalization)\.(\w+)$")
    for k, v in sd.items():
        m = pat.match(k)
        if not m or m.group(6) == "num_batches_tracked":
            continue
        s, b, which, idx, kind, leaf = m.groups()
        p = f"backbone/layer{int(s) + 1}/{b}"
        if which == "shortcut":
            if kind == "convolution":
                out[f"{p}/downsample_0/kernel"] = _conv(v)
            else:
                out[f"{p}/downsample_1/{leaf}"] = _np(v)
        else:
            n = int(idx) + 1
            if kind == "convolution":
                out[f"{p}/conv{n}/kernel"] = _conv(v)
            else:
                out[f"{p}/bn{n}/{leaf}"] = _np(v)


def _ln(sd, src, dst, out):
    out[f"{dst}/gamma"], out[f"{dst}/beta"] = _np(sd[src + ".weight"]), _np(sd[src + ".bias"])


def _lin(sd, src, dst, out):
    out[f"{dst}/kernel"], out[f"{dst}/bias"] = _np(sd[src + ".weight"]), _np(sd[src + ".bias"])


def _convert_detr(sd, out):
    _backbone_torchvision(sd, "backbone.0.body.", out)
    out["input_proj/kernel"], out["input_proj/bias"] = _conv(sd["input_proj.weight"]), _np(sd["input_proj.bias"])
    out["query_embed/kernel"] = _np(sd["query_embed.weight"])
    n_enc = 1 + max([int(m.group(1)) for k in sd for m in [re.match(r"transformer\.encoder\.layers\.(\d+)\.", k)] if m], default=-1)
    n_dec = 1 + max([int(m.group(1)) for k in sd for m in [re.match(r"transformer\.decoder\.layers\.(\d+)\.", k)] if m], default=-1)

    def mha(src, dst):
        out[f"{dst}/in_proj_kernel"], out[f"{dst}/in_proj_bias"] = _np(sd[src + ".in_proj_weight"]), _np(sd[src + ".in_proj_bias"])
        out[f"{dst}/out_proj_kernel"], out[f"{dst}/out_proj_bias"] = _np(sd[src + ".out_proj.weight"]), _np(sd[src + ".out_proj.bias"])

    for i in range(n_enc):
        s, d = f"transformer.encoder.layers.{i}", f"transformer/encoder/layer_{i}"
        mha(f"{s}.self_attn", f"{d}/self_attn")
        _lin(sd, f"{s}.linear1", f"{d}/linear1", out)
        _lin(sd, f"{s}.linear2", f"{d}/linear2", out)
        _ln(sd, f"{s}.norm1", f"{d}/norm1", out)
        _ln(sd, f"{s}.norm2", f"{d}/norm2", out)
    for i in range(n_dec):
        s, d = f"transformer.decoder.layers.{i}", f"transformer/decoder/layer_{i}"
        mha(f"{s}.self_attn", f"{d}/self_attn")
        mha(f"{s}.multihead_attn", f"{d}/multihead_attn")
        _lin(sd, f"{s}.linear1", f"{d}/linear1", out)
        _lin(sd, f"{s}.linear2", f"{d}/linear2", out)
        for n in (1, 2, 3):
            _ln(sd, f"{s}.norm{n}", f"{d}/norm{n}", out)
    _ln(sd, "transformer.decoder.norm", "transformer/decoder/norm", out)
    _lin(sd, "class_embed", "class_embed", out)
    for n in range(3):
        _lin(sd, f"bbox_embed.layers.{n}", f"bbox_embed_{n}", out)


def _convert_hf(sd, out, timm):
    if timm:
        pre = next(k for k in sd if k.endswith("conv_encoder.model.conv1.weight"))[:-len("conv1.weight")]
        _backbone_torchvision(sd, pre, out)
    else:
        pre = next(k for k in sd if k.endswith("embedder.embedder.convolution.weight"))[:-len("embedder.embedder.convolution.weight")]
        _backbone_hf(sd, pre, out)
    out["input_proj/kernel"], out["input_proj/bias"] = _conv(sd["model.input_projection.weight"]), _np(sd["model.input_projection.bias"])
    out["query_embed/kernel"] = _np(sd["model.query_position_embeddings.weight"])
    n_enc = 1 + max([int(m.group(1)) for k in sd for m in [re.match(r"model\.encoder\.layers\.(\d+)\.", k)] if m], default=-1)
    n_dec = 1 + max([int(m.group(1)) for k in sd for m in [re.match(r"model\.decoder\.layers\.(\d+)\.", k)] if m], default=-1)

    def mha(src, dst):
        oproj = "o_proj" if f"{src}.o_proj.weight" in sd else "out_proj"
        out[f"{dst}/in_proj_kernel"] = np.concatenate([_np(sd[f"{src}.{n}_proj.weight"]) for n in "qkv"], 0)
        out[f"{dst}/in_proj_bias"] = np.concatenate([_np(sd[f"{src}.{n}_proj.bias"]) for n in "qkv"], 0)
        out[f"{dst}/out_proj_kernel"], out[f"{dst}/out_proj_bias"] = _np(sd[f"{src}.{oproj}.weight"]), _np(sd[f"{src}.{oproj}.bias"])

    def ffn(src, dst):
        mlp = f"{src}.mlp." if f"{src}.mlp.fc1.weight" in sd else f"{src}."
        _lin(sd, mlp + "fc1", f"{dst}/linear1", out)
        _lin(sd, mlp + "fc2", f"{dst}/linear2", out)

    for i in range(n_enc):
        s, d = f"model.encoder.layers.{i}", f"transformer/encoder/layer_{i}"
        mha(f"{s}.self_attn", f"{d}/self_attn")
        ffn(s, d)
        _ln(sd, f"{s}.self_attn_layer_norm", f"{d}/norm1", out)
        _ln(sd, f"{s}.final_layer_norm", f"{d}/norm2", out)
    for i in range(n_dec):
        s, d = f"model.decoder.layers.{i}", f"transformer/decoder/layer_{i}"
        mha(f"{s}.self_attn", f"{d}/self_attn")
        mha(f"{s}.encoder_attn", f"{d}/multihead_attn")
        ffn(s, d)
        _ln(sd, f"{s}.self_attn_layer_norm", f"{d}/norm1", out)
        _ln(sd, f"{s}.encoder_attn_layer_norm", f"{d}/norm2", out)
        _ln(sd, f"{s}.final_layer_norm", f"{d}/norm3", out)
    _ln(sd, "model.decoder.layernorm", "transformer/decoder/norm", out)
    _lin(sd, "class_labels_classifier", "class_embed", out)
    for n in range(3):
        _lin(sd, f"bbox_predictor.layers.{n}", f"bbox_embed_{n}", out)


def convert_state_dict(state_dict):
    """PyTorch DETR state-dict (any of the three layouts; a {"model": ...} checkpoint wrapper is unwrapped) ->
    {reference layer name: float32 array} accepted by `model.load_weights` / `get_detr_model(weights=<path>.npz)`."""
    sd = state_dict.get("model", state_dict) if isinstance(state_dict, dict) and "model" in state_dict and isinstance(state_dict["model"], dict) else state_dict
    fmt = detect_format(sd)
    out = {}
    if fmt == "detr":
        _convert_detr(sd, out)
    else:
        _convert_hf(sd, out, timm=(fmt == "hf-timm"))
    return out


def export_state_dict(params):
    """Inverse of convert_state_dict for the facebookresearch/detr layout: {reference name: array} -> {pytorch name: float32 array}."""
    out = {}
    for k, v in params.items():
        v = np.asarray(v, dtype=np.float32)
        parts = k.split("/")
        leaf = parts[-1]
        if parts[0] == "backbone":
            body = "backbone.0.body."
            mid = ".".join(parts[1:-1]).replace("downsample_0", "downsample.0").replace("downsample_1", "downsample.1")
            if leaf == "kernel":
                out[f"{body}{mid}.weight"] = np.ascontiguousarray(v.transpose(3, 2, 0, 1))
            else:
                out[f"{body}{mid}.{leaf}"] = v
        elif k == "input_proj/kernel":
            out["input_proj.weight"] = np.ascontiguousarray(v.transpose(3, 2, 0, 1))
        elif k == "input_proj/bias":
            out["input_proj.bias"] = v
        elif k == "query_embed/kernel":
            out["query_embed.weight"] = v
        elif parts[0] == "transformer":
            mid = ".".join(parts[1:-1])
            mid = re.sub(r"(encoder|decoder)\.layer_(\d+)", r"\1.layers.\2", mid)
            name = {"in_proj_kernel": "in_proj_weight", "in_proj_bias": "in_proj_bias", "out_proj_kernel": "out_proj.weight",
                    "out_proj_bias": "out_proj.bias", "kernel": "weight", "bias": "bias", "gamma": "weight", "beta": "bias"}[leaf]
            out[f"transformer.{mid}.{name}"] = v
        elif parts[0] == "class_embed":
            out[f"class_embed.{'weight' if leaf == 'kernel' else 'bias'}"] = v
        elif parts[0].startswith("bbox_embed_"):
            out[f"bbox_embed.layers.{parts[0][-1]}.{'weight' if leaf == 'kernel' else 'bias'}"] = v
        else:
            raise KeyError(f"no PyTorch DETR counterpart for '{k}' (finetune heads are Keras Dense layers of the reference only)")
    return out


def map_tf_variables(variables, wanted):
    """TensorFlow variable names -> this package's parameter names.  `variables`: {name: array} from
    tf_checkpoint.load_tf_checkpoint (Keras names such as `detr/transformer/encoder/layer_0/linear1/kernel`, whatever model /
    scope prefix the saving program put in front); `wanted`: {parameter name: shape} (ParamStore.shapes + the frozen-BN
    vectors).  A variable is taken for parameter K when its name is K or ends with "/" + K -- the LONGEST such K wins, so
    `.../layer1/0/conv1/kernel` is never mistaken for the stem's `backbone/conv1/kernel` -- and the shape agrees.
    Returns (params, unused variable names)."""
    by_len = sorted(wanted, key=len, reverse=True)
    params, unused = {}, []
    for name, arr in variables.items():
        n = name[:-2] if name.endswith(":0") else name
        hit = next((k for k in by_len if n == k or n.endswith("/" + k)), None)
        if hit is None or tuple(np.shape(arr)) != tuple(wanted[hit]):
            unused.append(name)
            continue
        if hit in params:
            raise ValueError(f"two checkpoint variables map to {hit}: {name} and another one")
        params[hit] = np.ascontiguousarray(arr, dtype=np.float32)
    return params, unused


def load_tf_checkpoint_params(prefix, wanted):
    """Parameters of a TensorFlow checkpoint `<prefix>.index` / `.data-*` (the reference's `weights="detr"` files,
    weights.py:5-11,33) under this package's names; no TensorFlow needed (networks/tf_checkpoint.py)."""
    from .tf_checkpoint import load_tf_checkpoint
    dups = []
    variables = load_tf_checkpoint(prefix, duplicates=dups)
    # a variable name that occurs twice is only a problem when it is one of the wanted parameters (optimizer / bookkeeping
    # variables of a user checkpoint may repeat freely)
    by_len = sorted(wanted, key=len, reverse=True)
    for var, key in dups:
        hit = next((k for k in by_len if var == k or var.endswith("/" + k)), None)
        if hit is not None:
            raise ValueError(f"{prefix}: two checkpoint entries carry the variable name {var!r} (parameter {hit}; second entry {key})")
    return map_tf_variables(variables, wanted)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        raise SystemExit("usage: python -m detr_tf.networks.weights <detr checkpoint .pth | HF pytorch_model.bin | .safetensors> <out.npz>")
    src, dst = argv
    if src.endswith(".safetensors"):
        from safetensors.numpy import load_file
        sd = load_file(src)
    else:
        import torch
        sd = torch.load(src, map_location="cpu", weights_only=True)
    params = convert_state_dict(sd)
    np.savez(dst, **params)
    print(f"wrote {len(params)} tensors ({sum(v.size for v in params.values()):,} scalars) to {dst}")


if __name__ == "__main__":
    main()
