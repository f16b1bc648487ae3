"""ORACLE (test infrastructure only) -- CPU restatement of the reference's DETR forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker.  The shipped path (detr-tensorflow_amd/) never does.

PARITY UNPINNED at the TensorFlow boundary: the reference has no tests / golden vectors
for this path and TensorFlow is not installable in this image, so
this file is a line-faithful torch-CPU fp32 (or fp64) restatement of the reference's
graph, nothing more.  Every function cites the reference lines it follows
(paths relative to /root/reference).

Layout conventions follow the reference: images NHWC, conv kernels HWIO,
`Linear.kernel` is (out, in), MHA `in_proj_kernel` is (3*256, 256) rows [Q;K;V],
sequence-first [L, B, 256] inside the transformer.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # detr_tf/networks/custom_layers.py:5
KERAS_BN_EPS = 1.001e-5  # tf.keras.applications.resnet (third-party): BatchNormalization(epsilon=1.001e-5)
LN_EPS = 1e-5          # detr_tf/networks/transformer.py:151-152,200-202
RESNET50_BLOCKS = (3, 4, 6, 3)      # detr_tf/networks/resnet_backbone.py:39-48
RESNET101_BLOCKS = (3, 4, 23, 3)    # detr_tf/networks/resnet_backbone.py:56-65


# --------------------------------------------------------------------------------------
# parameter construction (seeded; SURVEY.md 8d "Weights")
# --------------------------------------------------------------------------------------
def tf_backbone_shapes(s, blocks=RESNET50_BLOCKS):
    """tf.keras.applications.ResNet50(include_top=False) (third-party Keras code the reference instantiates when
    tf_backbone=True, detr.py:146-148): every conv has a bias, BatchNormalization layers carry gamma / beta /
    moving_mean / moving_variance, block k of stack s is `conv{s}_block{k}_{0=shortcut,1,2,3}_{conv,bn}`."""
    def conv(name, kh, ci, co):
        s[f"resnet50/{name}_conv/kernel"] = (kh, kh, ci, co)
        s[f"resnet50/{name}_conv/bias"] = (co,)
        for n in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[f"resnet50/{name}_bn/{n}"] = (co,)

    conv("conv1", 7, 3, 64)
    cin = 64
    for li, nb in enumerate(blocks):
        f = 64 * 2 ** li
        for b in range(nb):
            q = f"conv{li + 2}_block{b + 1}"
            if b == 0:
                conv(f"{q}_0", 1, cin, 4 * f)
            conv(f"{q}_1", 1, cin, f)
            conv(f"{q}_2", 3, f, f)
            conv(f"{q}_3", 1, f, 4 * f)
            cin = 4 * f


def param_shapes(blocks=RESNET50_BLOCKS, num_enc=6, num_dec=6, num_queries=100,
                 num_classes=92, model_dim=256, ff=2048, nb_class=None, tf_backbone=False):
    """Ordered dict name -> shape, names per SURVEY.md A.6 (Keras layer/weight names)."""
    s = {}

    def bn(prefix, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}/{n}"] = (c,)

    if tf_backbone:
        tf_backbone_shapes(s, blocks)
    else:
        s["backbone/conv1/kernel"] = (7, 7, 3, 64)
        bn("backbone/bn1", 64)
    cin = 64
    for li, nb in enumerate(blocks if not tf_backbone else ()):
        d1 = 64 * 2 ** li
        d2 = 4 * d1
        for b in range(nb):
            p = f"backbone/layer{li + 1}/{b}"
            s[f"{p}/conv1/kernel"] = (1, 1, cin, d1)
            bn(f"{p}/bn1", d1)
            s[f"{p}/conv2/kernel"] = (3, 3, d1, d1)
            bn(f"{p}/bn2", d1)
            s[f"{p}/conv3/kernel"] = (1, 1, d1, d2)
            bn(f"{p}/bn3", d2)
            if b == 0:   # resnet_backbone.py:131-132 -- only block 0 uses the downsample
                s[f"{p}/downsample_0/kernel"] = (1, 1, cin, d2)
                bn(f"{p}/downsample_1", d2)
            cin = d2
    s["input_proj/kernel"] = (1, 1, 2048, model_dim)
    s["input_proj/bias"] = (model_dim,)
    s["query_embed/kernel"] = (num_queries, model_dim)

    def mha(p):
        s[f"{p}/in_proj_kernel"] = (3 * model_dim, model_dim)
        s[f"{p}/in_proj_bias"] = (3 * model_dim,)
        s[f"{p}/out_proj_kernel"] = (model_dim, model_dim)
        s[f"{p}/out_proj_bias"] = (model_dim,)

    def lin(p, o, i):
        s[f"{p}/kernel"] = (o, i)
        s[f"{p}/bias"] = (o,)

    def ln(p):
        s[f"{p}/gamma"] = (model_dim,)
        s[f"{p}/beta"] = (model_dim,)

    for i in range(num_enc):
        p = f"transformer/encoder/layer_{i}"
        mha(f"{p}/self_attn")
        lin(f"{p}/linear1", ff, model_dim)
        lin(f"{p}/linear2", model_dim, ff)
        ln(f"{p}/norm1")
        ln(f"{p}/norm2")
    for i in range(num_dec):
        p = f"transformer/decoder/layer_{i}"
        mha(f"{p}/self_attn")
        mha(f"{p}/multihead_attn")
        lin(f"{p}/linear1", ff, model_dim)
        lin(f"{p}/linear2", model_dim, ff)
        ln(f"{p}/norm1")
        ln(f"{p}/norm2")
        ln(f"{p}/norm3")
    ln("transformer/decoder/norm")
    if nb_class is None:
        lin("class_embed", num_classes, model_dim)
        lin("bbox_embed_0", model_dim, model_dim)
        lin("bbox_embed_1", model_dim, model_dim)
        lin("bbox_embed_2", 4, model_dim)
    else:
        # finetune heads are Keras Dense: kernel (in, out)  (detr.py:97-102)
        s["cls_layer/kernel"] = (model_dim, nb_class)
        s["cls_layer/bias"] = (nb_class,)
        s["pos_layer/dense_0/kernel"] = (model_dim, 256)
        s["pos_layer/dense_0/bias"] = (256,)
        s["pos_layer/dense_1/kernel"] = (256, 256)
        s["pos_layer/dense_1/bias"] = (256,)
        s["pos_layer/dense_2/kernel"] = (256, 4)
        s["pos_layer/dense_2/bias"] = (4,)
    return s


def make_params(seed=0, **kw):
    """Seeded synthetic weights: He-normal convs, BN weight~U(.5,1.5),
    bias/mean~N(0,.1), var~U(.5,1.5), Glorot-uniform linears.  Returns name -> np.float32."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in param_shapes(**kw).items():
        leaf = name.rsplit("/", 1)[1]
        is_bn = ("/bn" in name) or ("/downsample_1/" in name) or ("_bn/" in name)
        if is_bn and leaf in ("weight", "gamma") and "/norm" not in name:
            # the last BN of a residual branch gets a small gain so that 16 stacked blocks keep O(1) activations
            v = rng.uniform(0.2, 0.4, shp) if ("/bn3/" in name or "_3_bn/" in name) else rng.uniform(0.5, 1.5, shp)
        elif is_bn and leaf in ("running_var", "moving_variance"):
            v = rng.uniform(0.5, 1.5, shp)
        elif is_bn and leaf in ("running_mean", "bias", "moving_mean", "beta"):
            v = rng.normal(0.0, 0.1, shp)
        elif leaf == "gamma":
            v = rng.uniform(0.8, 1.2, shp)
        elif leaf == "beta":
            v = rng.normal(0.0, 0.05, shp)
        elif len(shp) == 4:
            fan_in = shp[0] * shp[1] * shp[2]
            gain = 1.0 if ("conv3" in name or "downsample_0" in name or "input_proj" in name or "_3_conv" in name or "_0_conv" in name) else math.sqrt(2.0)
            v = rng.normal(0.0, gain / math.sqrt(fan_in), shp)
        elif len(shp) == 2:
            lim = math.sqrt(6.0 / (shp[0] + shp[1]))
            v = rng.uniform(-lim, lim, shp)
            if name == "query_embed/kernel":
                v = rng.normal(0.0, 1.0, shp)
        else:
            v = rng.uniform(-0.05, 0.05, shp)
        out[name] = v.astype(np.float32)
    return out


def to_torch(params, dtype=torch.float32, requires_grad=False):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad and trainable(k))
            for k, v in params.items()}


def trainable(name):
    """FrozenBatchNorm2D vectors are trainable=False (custom_layers.py:11-18); with tf_backbone=True the Keras
    BatchNormalization layers are set trainable=False by optimizers.disable_batchnorm_training (optimizers.py:3-8)."""
    return not (("/bn" in name) or ("/downsample_1/" in name) or ("_bn/" in name))


# --------------------------------------------------------------------------------------
# backbone  (detr_tf/networks/resnet_backbone.py, custom_layers.py)
# --------------------------------------------------------------------------------------
def frozen_bn(x_nhwc, P, prefix):
    """custom_layers.py:21-24."""
    scale = P[f"{prefix}/weight"] * torch.rsqrt(P[f"{prefix}/running_var"] + BN_EPS)
    shift = P[f"{prefix}/bias"] - P[f"{prefix}/running_mean"] * scale
    return x_nhwc * scale + shift


def conv2d_valid(x_nhwc, k_hwio, stride=1, pad=0, dilation=1):
    """ZeroPadding2D(pad) followed by Conv2D(padding='valid') (resnet_backbone.py:11-13,98-105)."""
    x = x_nhwc.permute(0, 3, 1, 2)
    w = k_hwio.permute(3, 2, 0, 1)
    y = F.conv2d(x, w, None, stride=stride, padding=pad, dilation=dilation)
    return y.permute(0, 2, 3, 1)


def bottleneck(x, P, p, stride, downsample, taps=None):
    """BottleNeck.call resnet_backbone.py:116-137 (dilation is always 1: :36,:80-85)."""
    y1 = torch.relu(frozen_bn(conv2d_valid(x, P[f"{p}/conv1/kernel"]), P, f"{p}/bn1"))
    y2 = torch.relu(frozen_bn(conv2d_valid(y1, P[f"{p}/conv2/kernel"], stride=stride, pad=1), P, f"{p}/bn2"))
    if taps is not None and taps.get("_blocks"):
        taps[f"{p}:x"], taps[f"{p}:y1"], taps[f"{p}:y2"] = x, y1, y2
        for t in (y1, y2):
            if t.requires_grad:
                t.retain_grad()
    out = frozen_bn(conv2d_valid(y2, P[f"{p}/conv3/kernel"]), P, f"{p}/bn3")
    if downsample:
        identity = frozen_bn(conv2d_valid(x, P[f"{p}/downsample_0/kernel"], stride=stride), P, f"{p}/downsample_1")
    else:
        identity = x
    return torch.relu(out + identity)


def backbone(images_nhwc, P, blocks=RESNET50_BLOCKS, taps=None):
    """ResNetBase.call resnet_backbone.py:20-32."""
    x = conv2d_valid(images_nhwc, P["backbone/conv1/kernel"], stride=2, pad=3)
    x = torch.relu(frozen_bn(x, P, "backbone/bn1"))
    if taps is not None:
        taps["stem_conv"] = x
    # pad2 = ZeroPadding2D(1) then MaxPool2D(3, 2, 'valid'): the zero padding takes part in the max
    xp = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1), value=0.0)
    x = F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1)
    if taps is not None:
        taps["stem_pool"] = x
    for li, nb in enumerate(blocks):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 0) else 1     # resnet_backbone.py:39-48,80-81
            x = bottleneck(x, P, f"backbone/layer{li + 1}/{b}", stride, b == 0, taps)
        if taps is not None:
            taps[f"layer{li + 1}"] = x
    return x


def keras_bn(x, P, prefix):
    """tf.keras BatchNormalization in inference mode (trainable=False): gamma * (x - mean) / sqrt(var + eps) + beta."""
    inv = P[f"{prefix}/gamma"] * torch.rsqrt(P[f"{prefix}/moving_variance"] + KERAS_BN_EPS)
    return x * inv + (P[f"{prefix}/beta"] - P[f"{prefix}/moving_mean"] * inv)


def backbone_tf(images_nhwc, P, blocks=RESNET50_BLOCKS, taps=None):
    """tf.keras.applications.resnet.ResNet50(include_top=False) as the reference uses it for tf_backbone=True
    (detr.py:146-148; third-party Keras code, restated from its published definition): ZeroPadding2D(3) + 7x7/2 conv WITH
    bias + BN + ReLU + ZeroPadding2D(1) + 3x3/2 max pool; bottleneck `block1`: shortcut 1x1 conv (stride s) + BN on the
    first block of a stack, then 1x1 conv (stride s -- ResNet v1: the stride sits on the FIRST 1x1) + BN + ReLU, 3x3
    'same' conv + BN + ReLU, 1x1 conv + BN, add, ReLU; stacks (64,3,s1) (128,4,s2) (256,6,s2) (512,3,s2)."""
    def conv(x, name, stride=1, pad=0):
        return conv2d_valid(x, P[f"resnet50/{name}_conv/kernel"], stride=stride, pad=pad) + P[f"resnet50/{name}_conv/bias"]

    x = torch.relu(keras_bn(conv(images_nhwc, "conv1", stride=2, pad=3), P, "resnet50/conv1_bn"))
    if taps is not None:
        taps["stem_conv"] = x
    xp = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1), value=0.0)
    x = F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1)
    for li, nb in enumerate(blocks):
        for b in range(nb):
            q = f"conv{li + 2}_block{b + 1}"
            stride = 2 if (b == 0 and li > 0) else 1
            shortcut = keras_bn(conv(x, f"{q}_0", stride=stride), P, f"resnet50/{q}_0_bn") if b == 0 else x
            y = torch.relu(keras_bn(conv(x, f"{q}_1", stride=stride), P, f"resnet50/{q}_1_bn"))
            y = torch.relu(keras_bn(conv(y, f"{q}_2", pad=1), P, f"resnet50/{q}_2_bn"))
            y = keras_bn(conv(y, f"{q}_3"), P, f"resnet50/{q}_3_bn")
            x = torch.relu(shortcut + y)
        if taps is not None:
            taps[f"layer{li + 1}"] = x
    return x


# --------------------------------------------------------------------------------------
# positional encoding (detr_tf/networks/position_embeddings.py:23-50, zero mask detr.py:172)
# --------------------------------------------------------------------------------------
def position_embedding_sine(B, H, W, num_pos_features=128, temperature=10000.0, eps=1e-6,
                            dtype=torch.float32):
    not_mask = torch.ones(B, H, W, dtype=dtype)
    y_embed = torch.cumsum(not_mask, 1)
    x_embed = torch.cumsum(not_mask, 2)
    scale = 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_features, dtype=dtype)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_features)
    pos_x = x_embed[..., None] / dim_t
    pos_y = y_embed[..., None] / dim_t
    pos_x = torch.stack([pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()], dim=4).reshape(B, H, W, -1)
    pos_y = torch.stack([pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()], dim=4).reshape(B, H, W, -1)
    return torch.cat([pos_y, pos_x], dim=3)


# --------------------------------------------------------------------------------------
# transformer  (detr_tf/networks/transformer.py)
# --------------------------------------------------------------------------------------
def linear(x, P, p):
    """custom_layers.py:49-50: x @ kernel^T + bias, kernel (out, in)."""
    return x @ P[f"{p}/kernel"].t() + P[f"{p}/bias"]


def layer_norm(x, P, p):
    return F.layer_norm(x, (x.shape[-1],), P[f"{p}/gamma"], P[f"{p}/beta"], LN_EPS)


def multi_head_attention(query, key, value, P, p, num_heads=8, drop=None, seed=0):
    """MultiHeadAttention.call transformer.py:285-356 (attn_mask None, key-padding branch
    disabled :322-337, dropout = identity)."""
    T, B, D = query.shape
    S = key.shape[0]
    hd = D // num_heads
    W, b = P[f"{p}/in_proj_kernel"], P[f"{p}/in_proj_bias"]
    WQ = query @ W[:D].t() + b[:D]
    WK = key @ W[D:2 * D].t() + b[D:2 * D]
    WV = value @ W[2 * D:].t() + b[2 * D:]
    WQ = WQ * float(hd) ** -0.5                               # :307 scale AFTER the bias
    WQ = WQ.reshape(T, B * num_heads, hd).transpose(0, 1)     # :308-309
    WK = WK.reshape(S, B * num_heads, hd).transpose(0, 1)
    WV = WV.reshape(S, B * num_heads, hd).transpose(0, 1)
    w = torch.softmax(WQ @ WK.transpose(1, 2), dim=-1)        # :317,340
    if drop is not None:
        w = drop(seed + 0, w, "attn")                         # :341 dropout on the attention weights
    o = (w @ WV).transpose(0, 1).reshape(T, B, D)             # :343-345
    out = o @ P[f"{p}/out_proj_kernel"].t() + P[f"{p}/out_proj_bias"]   # :346-347
    if drop is not None:
        out = drop(seed + 1, out, "lbc")                      # the caller's self.dropout(attn_output) :169,215,226
    return out


def _ffn(x, P, p, drop, seed):
    h = torch.relu(linear(x, P, f"{p}/linear1"))
    if drop is not None:
        h = drop(seed + 0, h, "lbc")                          # :174 / :230
    y = linear(h, P, f"{p}/linear2")
    if drop is not None:
        y = drop(seed + 1, y, "lbc")                          # :176 / :232
    return y


def encoder_layer(src, pos, P, p, drop=None, seed=0):
    """EncoderLayer.call transformer.py:157-179 (post-norm); dropout sites seed+0..3."""
    q = k = src + pos
    src = layer_norm(src + multi_head_attention(q, k, src, P, f"{p}/self_attn", drop=drop, seed=seed), P, f"{p}/norm1")
    return layer_norm(src + _ffn(src, P, p, drop, seed + 2), P, f"{p}/norm2")


def decoder_layer(tgt, memory, pos, qpos, P, p, drop=None, seed=0):
    """DecoderLayer.call transformer.py:207-234; dropout sites seed+0..5."""
    q = k = tgt + qpos
    tgt = layer_norm(tgt + multi_head_attention(q, k, tgt, P, f"{p}/self_attn", drop=drop, seed=seed), P, f"{p}/norm1")
    tgt = layer_norm(tgt + multi_head_attention(tgt + qpos, memory + pos, memory, P, f"{p}/multihead_attn", drop=drop,
                                                seed=seed + 2), P, f"{p}/norm2")
    return layer_norm(tgt + _ffn(tgt, P, p, drop, seed + 4), P, f"{p}/norm3")


def transformer(src_nhwc, pos_nhwc, query_embed, P, num_enc=6, num_dec=6, taps=None, drop=None):
    """Transformer.call transformer.py:29-57; returns hs [num_dec, B, Q, 256]."""
    B, H, W, D = src_nhwc.shape
    src = src_nhwc.reshape(B, H * W, D).transpose(0, 1)
    pos = pos_nhwc.reshape(B, H * W, D).transpose(0, 1)
    qpos = query_embed[:, None, :].expand(-1, B, -1)
    tgt = torch.zeros_like(qpos)
    x = src
    for i in range(num_enc):
        x = encoder_layer(x, pos, P, f"transformer/encoder/layer_{i}", drop, 16 * i)
    memory = x
    if taps is not None:
        taps["memory"] = memory
    inter = []
    x = tgt
    for i in range(num_dec):
        x = decoder_layer(x, memory, pos, qpos, P, f"transformer/decoder/layer_{i}", drop, 16 * (32 + i))
        inter.append(layer_norm(x, P, "transformer/decoder/norm"))     # :121-125
    hs = torch.stack(inter, 0)                                          # [num_dec, Q, B, D]
    return hs.transpose(1, 2)                                           # :53


# --------------------------------------------------------------------------------------
# full model  (detr_tf/networks/detr.py:116-204)
# --------------------------------------------------------------------------------------
def detr_hs(images_nhwc, P, blocks=RESNET50_BLOCKS, num_enc=6, num_dec=6, taps=None, drop=None):
    """The inner Keras model "detr": images -> hs  (detr.py:170-177)."""
    x = backbone_tf(images_nhwc, P, blocks, taps) if "resnet50/conv1_conv/kernel" in P else backbone(images_nhwc, P, blocks, taps)
    B, H, W, _ = x.shape
    pos = position_embedding_sine(B, H, W, dtype=x.dtype)
    proj = conv2d_valid(x, P["input_proj/kernel"]) + P["input_proj/bias"]
    if taps is not None:
        taps["input_proj"] = proj
        taps["pos"] = pos
    return transformer(proj, pos, P["query_embed/kernel"], P, num_enc, num_dec, taps, drop)


def detr_forward(images_nhwc, P, blocks=RESNET50_BLOCKS, num_enc=6, num_dec=6, taps=None, drop=None):
    """get_detr_model(include_top=True) output dict (detr.py:181-204).  `drop`: optional
    oracle.dropout_ref.Dropper for training-mode parity (None = dropout off / inference)."""
    hs = detr_hs(images_nhwc, P, blocks, num_enc, num_dec, taps, drop)
    if taps is not None:
        taps["hs"] = hs
    if "class_embed/kernel" in P:
        logits = linear(hs, P, "class_embed")
        t = torch.relu(linear(hs, P, "bbox_embed_0"))
        t = torch.relu(linear(t, P, "bbox_embed_1"))
        boxes = torch.sigmoid(linear(t, P, "bbox_embed_2"))
        n_aux = num_dec - 1                                            # detr.py:195
    else:
        # add_heads_nlayers detr.py:94-114 (Keras Dense kernels are (in, out))
        logits = hs @ P["cls_layer/kernel"] + P["cls_layer/bias"]
        t = torch.relu(hs @ P["pos_layer/dense_0/kernel"] + P["pos_layer/dense_0/bias"])
        t = torch.relu(t @ P["pos_layer/dense_1/kernel"] + P["pos_layer/dense_1/bias"])
        boxes = torch.sigmoid(t @ P["pos_layer/dense_2/kernel"] + P["pos_layer/dense_2/bias"])
        n_aux = 5                                                      # detr.py:111 hard-coded range(0,5)
    out = {"pred_logits": logits[-1], "pred_boxes": boxes[-1]}
    out["aux"] = [{"pred_logits": logits[i], "pred_boxes": boxes[i]} for i in range(n_aux)]
    return out
