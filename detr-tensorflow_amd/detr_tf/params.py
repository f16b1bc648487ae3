"""Flat fp32 parameter store of the DETR hot path.

All TRAINABLE tensors live in ONE contiguous fp32 device buffer (plus same-shaped flat buffers
for gradients and the Adam moments) so that the optimiser and the data-parallel all-reduce are
a handful of large, HBM-friendly launches / collectives.  The order is reverse-forward (heads,
decoder, encoder, input_proj, query_embed, layer4 .. layer1, stem): gradients become final in
exactly that order during the backward pass, so contiguous prefixes of the gradient buffer can
be all-reduced over RCCL while the rest of the backward still runs.

Names and shapes follow the reference's Keras layer/weight names:
conv kernels HWIO, `Linear.kernel` (out, in), MHA `in_proj_kernel` (768, 256) rows [Q;K;V],
finetune heads are Keras Dense (in, out).  Frozen-BN vectors (custom_layers.py:11-18,
trainable=False) are kept outside the flat buffer.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

RESNET50_BLOCKS = (3, 4, 6, 3)      # resnet_backbone.py:39-48
RESNET101_BLOCKS = (3, 4, 23, 3)    # resnet_backbone.py:56-65
GROUPS = ("backbone", "transformers", "nlayers")


def variable_group(name, nlayers):
    """detr_tf/optimizers.py:10-43 of the reference: backbone = every layer of the inner model
    except the transformer (ResNet + input_proj + query_embed); transformers = transformer + outer
    layers not in config.nlayers (class_embed, bbox_embed_*); nlayers = config.nlayers."""
    top = name.split("/", 1)[0]
    if top in nlayers:
        return 2
    if top == "transformer" or top.startswith("class_embed") or top.startswith("bbox_embed"):
        return 1
    return 0


def block_names(li, b, tf_backbone=False):
    """Layer names of bottleneck b of stage li: conv1/bn1 (1x1), conv2/bn2 (3x3), conv3/bn3 (1x1), down/bnd (shortcut).
    Reference backbone: resnet_backbone.py:94-137 (`layer{n}/{b}/convK`, `bnK`, `downsample_0/1`); tf_backbone=True:
    tf.keras.applications ResNet50 (`conv{s}_block{k}_{1,2,3,0}_{conv,bn}`)."""
    if tf_backbone:
        q = f"resnet50/conv{li + 2}_block{b + 1}"
        return dict(conv1=f"{q}_1_conv", bn1=f"{q}_1_bn", conv2=f"{q}_2_conv", bn2=f"{q}_2_bn", conv3=f"{q}_3_conv", bn3=f"{q}_3_bn",
                    down=f"{q}_0_conv", bnd=f"{q}_0_bn", tag=f"{q}")
    p = f"backbone/layer{li + 1}/{b}"
    return dict(conv1=f"{p}/conv1", bn1=f"{p}/bn1", conv2=f"{p}/conv2", bn2=f"{p}/bn2", conv3=f"{p}/conv3", bn3=f"{p}/bn3",
                down=f"{p}/downsample_0", bnd=f"{p}/downsample_1", tag=p)


def stem_names(tf_backbone=False):
    return dict(conv="resnet50/conv1_conv", bn="resnet50/conv1_bn") if tf_backbone else dict(conv="backbone/conv1", bn="backbone/bn1")


BN_LEAVES = {False: ("weight", "bias", "running_mean", "running_var"),          # FrozenBatchNorm2D custom_layers.py:11-18
             True: ("gamma", "beta", "moving_mean", "moving_variance")}         # tf.keras BatchNormalization


def trainable_shapes(blocks, num_enc, num_dec, num_queries, num_classes, nb_class, model_dim=256, ff=2048, tf_backbone=False):
    """Ordered (reverse-forward) dict name -> shape of the trainable tensors."""
    s = OrderedDict()

    def lin(p, o, i):
        s[f"{p}/kernel"] = (o, i)
        s[f"{p}/bias"] = (o,)

    def mha(p):
        s[f"{p}/in_proj_kernel"] = (3 * model_dim, model_dim)
        s[f"{p}/in_proj_bias"] = (3 * model_dim,)
        s[f"{p}/out_proj_kernel"] = (model_dim, model_dim)
        s[f"{p}/out_proj_bias"] = (model_dim,)

    def ln(p):
        s[f"{p}/gamma"] = (model_dim,)
        s[f"{p}/beta"] = (model_dim,)

    if nb_class is None:
        lin("class_embed", num_classes, model_dim)
        lin("bbox_embed_0", model_dim, model_dim)
        lin("bbox_embed_1", model_dim, model_dim)
        lin("bbox_embed_2", 4, model_dim)
    else:
        s["cls_layer/kernel"] = (model_dim, nb_class)
        s["cls_layer/bias"] = (nb_class,)
        for i, (a, b) in enumerate([(model_dim, 256), (256, 256), (256, 4)]):
            s[f"pos_layer/dense_{i}/kernel"] = (a, b)
            s[f"pos_layer/dense_{i}/bias"] = (b,)
    ln("transformer/decoder/norm")
    for i in reversed(range(num_dec)):
        p = f"transformer/decoder/layer_{i}"
        mha(f"{p}/self_attn")
        mha(f"{p}/multihead_attn")
        lin(f"{p}/linear1", ff, model_dim)
        lin(f"{p}/linear2", model_dim, ff)
        ln(f"{p}/norm1")
        ln(f"{p}/norm2")
        ln(f"{p}/norm3")
    for i in reversed(range(num_enc)):
        p = f"transformer/encoder/layer_{i}"
        mha(f"{p}/self_attn")
        lin(f"{p}/linear1", ff, model_dim)
        lin(f"{p}/linear2", model_dim, ff)
        ln(f"{p}/norm1")
        ln(f"{p}/norm2")
    s["input_proj/kernel"] = (1, 1, 2048, model_dim)
    s["input_proj/bias"] = (model_dim,)
    s["query_embed/kernel"] = (num_queries, model_dim)
    cins = [64]
    for li in range(4):
        cins.append(256 * 2 ** li)
    def conv(name, shape):
        s[f"{name}/kernel"] = shape
        if tf_backbone:                       # keras.applications convs carry a (trainable) bias
            s[f"{name}/bias"] = (shape[3],)

    for li in reversed(range(4)):
        d1 = 64 * 2 ** li
        d2 = 4 * d1
        for b in reversed(range(blocks[li])):
            cin = cins[li] if b == 0 else d2
            n = block_names(li, b, tf_backbone)
            conv(n["conv3"], (1, 1, d1, d2))
            conv(n["conv2"], (3, 3, d1, d1))
            conv(n["conv1"], (1, 1, cin, d1))
            if b == 0:
                conv(n["down"], (1, 1, cin, d2))
    conv(stem_names(tf_backbone)["conv"], (7, 7, 3, 64))
    return s


def bn_names(blocks, tf_backbone=False):
    """name prefix -> channels of every frozen batch norm used by the graph."""
    out = OrderedDict()
    out[stem_names(tf_backbone)["bn"]] = 64
    for li in range(4):
        d1 = 64 * 2 ** li
        d2 = 4 * d1
        for b in range(blocks[li]):
            n = block_names(li, b, tf_backbone)
            out[n["bn1"]] = d1
            out[n["bn2"]] = d1
            out[n["bn3"]] = d2
            if b == 0:
                out[n["bnd"]] = d2
    return out


def bn_conv_pairs(blocks, tf_backbone=False):
    """[(bn prefix, conv prefix)] of every conv + frozen BN pair of the backbone, stem first."""
    st = stem_names(tf_backbone)
    out = [(st["bn"], st["conv"])]
    for li in range(4):
        for b in range(blocks[li]):
            n = block_names(li, b, tf_backbone)
            out += [(n["bn1"], n["conv1"]), (n["bn2"], n["conv2"]), (n["bn3"], n["conv3"])]
            if b == 0:
                out.append((n["bnd"], n["down"]))
    return out


class ParamStore:
    def __init__(self, device, blocks=RESNET50_BLOCKS, num_enc=6, num_dec=6, num_queries=100, num_classes=92,
                 nb_class=None, seed=0, tf_backbone=False):
        self.device = device
        self.tf_backbone = bool(tf_backbone)
        self.blocks = tuple(blocks)
        self.bn_leaves = BN_LEAVES[self.tf_backbone]
        self.shapes = trainable_shapes(blocks, num_enc, num_dec, num_queries, num_classes, nb_class, tf_backbone=tf_backbone)
        self.bn = bn_names(blocks, tf_backbone)
        self.offsets = OrderedDict()
        off = 0
        for k, shp in self.shapes.items():
            n = int(np.prod(shp))
            self.offsets[k] = (off, n)
            off += (n + 7) // 8 * 8          # every tensor 16-byte aligned in the fp32 buffer AND in its bf16 shadow
        self.total = off
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.views = {k: self.flat[o:o + n].view(self.shapes[k]) for k, (o, n) in self.offsets.items()}
        self.gviews = {k: self.grad[o:o + n].view(self.shapes[k]) for k, (o, n) in self.offsets.items()}
        self.flat16, self.views16 = None, None      # bf16 shadow of `flat` (bf16 compute mode; refreshed by the engine)
        self.on_change = None            # called after every mutation of the parameters (the engine's weights-version bump)
        # frozen BN raw vectors [4, C] per layer: weight, bias, running_mean, running_var
        self.bn_raw = {p: torch.zeros(4, c, dtype=torch.float32, device=device) for p, c in self.bn.items()}
        for p in self.bn_raw:
            self.bn_raw[p][3].fill_(1.0)
        self.init_random(seed)

    def shadow16(self):
        """bf16 twin of the flat parameter buffer (same offsets): the weight operand of the bf16-compute kernels."""
        if self.flat16 is None:
            self.flat16 = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
            self.views16 = {k: self.flat16[o:o + n].view(self.shapes[k]) for k, (o, n) in self.offsets.items()}
        return self.flat16

    # ---- initialisation / IO --------------------------------------------------------------
    def init_random(self, seed):
        """Keras defaults of the reference layers: GlorotUniform kernels/biases
        (custom_layers.py:11-18,41-47,63-64; transformer.py:253-268), BN mean 0 / var 1."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.total, np.float32)
        for k, shp in self.shapes.items():
            o, n = self.offsets[k]
            if len(shp) == 4:
                fan_in, fan_out = shp[0] * shp[1] * shp[2], shp[0] * shp[1] * shp[3]
            elif len(shp) == 2:
                fan_in, fan_out = shp[1], shp[0]
            else:
                fan_in = fan_out = shp[0]
            if k.endswith("gamma"):
                v = np.ones(shp, np.float32)
            elif k.endswith("beta"):
                v = np.zeros(shp, np.float32)
            else:
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                v = rng.uniform(-lim, lim, shp).astype(np.float32)
            host[o:o + n] = v.ravel()
        self.flat.copy_(torch.from_numpy(host))
        for p, c in self.bn.items():
            lim = math.sqrt(6.0 / (2 * c))
            raw = np.stack([rng.uniform(-lim, lim, c), rng.uniform(-lim, lim, c), np.zeros(c), np.ones(c)]).astype(np.float32)
            self.bn_raw[p].copy_(torch.from_numpy(raw))

    def load_dict(self, params):
        """params: name -> array, names per SURVEY.md A.6 (BN vectors as <prefix>/{weight,bias,running_mean,running_var})."""
        missing = []
        # every shape is checked BEFORE the first copy: a mismatch must not leave the store half overwritten (no BN refold, no
        # weights-version bump -> stale recorded graphs / bf16 shadow / folded kernels)
        staged = []
        for k in self.shapes:
            if k in params:
                v = torch.as_tensor(np.asarray(params[k]), dtype=torch.float32)
                if tuple(v.shape) != tuple(self.shapes[k]):
                    raise ValueError(f"{k}: shape {tuple(v.shape)} != {self.shapes[k]}")
                staged.append((self.views[k], v))
            else:
                missing.append(k)
        for p in self.bn:
            for i, leaf in enumerate(self.bn_leaves):
                if f"{p}/{leaf}" in params:
                    v = torch.as_tensor(np.asarray(params[f"{p}/{leaf}"]), dtype=torch.float32)
                    if v.numel() != self.bn_raw[p][i].numel():
                        raise ValueError(f"{p}/{leaf}: {v.numel()} values != {self.bn_raw[p][i].numel()}")
                    staged.append((self.bn_raw[p][i], v.reshape(self.bn_raw[p][i].shape)))
                else:
                    missing.append(f"{p}/{leaf}")
        try:
            for dst, v in staged:
                dst.copy_(v)
        finally:
            if self.on_change is not None:
                self.on_change()
        return missing

    def state_dict(self):
        out = {k: v.detach().cpu().numpy().copy() for k, v in self.views.items()}
        for p in self.bn:
            for i, leaf in enumerate(self.bn_leaves):
                out[f"{p}/{leaf}"] = self.bn_raw[p][i].cpu().numpy().copy()
        return out

    def save(self, path):
        np.savez(path, **self.state_dict())

    def load(self, path):
        with np.load(path) as z:
            return self.load_dict({k: z[k] for k in z.files})

    # ---- optimiser tables -------------------------------------------------------------------
    def build_tables(self, nlayers, chunk=8192):
        names = list(self.shapes)
        ct, cs, seg_end, grp = [], [], [], []
        for t, k in enumerate(names):
            o, n = self.offsets[k]
            seg_end.append(o + n)
            grp.append(variable_group(k, nlayers))
            for st in range(0, n, chunk):
                ct.append(t)
                cs.append(o + st)
        dev = self.device
        return {
            "names": names,
            "chunk": chunk,
            "n_chunks": len(ct),
            "chunk_tensor": torch.tensor(ct, dtype=torch.int32, device=dev),
            "chunk_start": torch.tensor(cs, dtype=torch.int64, device=dev),
            "seg_end": torch.tensor(seg_end, dtype=torch.int64, device=dev),
            "group_host": grp,
        }

    def bucket_bounds(self):
        """Element ranges of the gradient buffer that become final together during backward:
        [heads+decoder+encoder], [input_proj, query_embed, layer4], [layer3], [layer2, layer1, stem]."""
        def start(name):
            return self.offsets[name][0]
        b1 = start("input_proj/kernel")
        # the first tensor (in reverse-forward order) of stage 3 / stage 2: conv3 of their last block
        l3 = block_names(2, self.blocks[2] - 1, self.tf_backbone)["conv3"] + "/kernel"
        l2 = block_names(1, self.blocks[1] - 1, self.tf_backbone)["conv3"] + "/kernel"
        return [(0, b1), (b1, start(l3)), (start(l3), start(l2)), (start(l2), self.total)]
