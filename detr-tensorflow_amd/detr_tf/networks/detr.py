"""Model factory of the drop-in API: `get_detr_model(...)` (reference detr_tf/networks/detr.py:116-204).

The returned object is callable like the reference's Keras model -- `model(images, training=bool)`
with fp32 NHWC images -- and returns the same output structure
`{"pred_logits": [B,Q,C], "pred_boxes": [B,Q,4], "aux": [{...}] * (levels-1)}` (detr.py:190-204);
underneath every layer is a HIP kernel launch of `engine.DetrEngine`.
"""
import os

import numpy as np
import torch

from ..engine import DetrEngine
from ..params import RESNET50_BLOCKS, RESNET101_BLOCKS


class DetrOutputs(dict):
    """Output dict + handles used by get_losses / run_train_step (levels tensors, loss state)."""
    levels_logits = None
    levels_boxes = None
    set_loss = None
    reduce_sums = None
    model = None


class _Layer:
    def __init__(self, name, variables):
        self.name = name
        self.trainable_variables = variables


class DetrModel:
    def __init__(self, include_top=True, nb_class=None, num_decoder_layers=6, num_encoder_layers=6, num_queries=100,
                 backbone="resnet50", device=None, seed=0, dropout=0.1, precision="fp32", tf_backbone=False):
        device = device or (f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else None)
        if device is None:
            raise RuntimeError("DETR HIP model needs a GPU: the hot path has no CPU fallback")
        blocks = {"resnet50": RESNET50_BLOCKS, "resnet101": RESNET101_BLOCKS}[backbone]
        self.include_top = include_top
        self.headless = (not include_top) and nb_class is None
        self.name = "detr" if self.headless else "detr_finetuning"
        self.tf_backbone = bool(tf_backbone)
        self.eval_graph, self._eval_graph, self._eval_seen = True, None, None     # hipGraph replay of the eval forward (_eval_forward)
        self.engine = DetrEngine(device, blocks, num_encoder_layers, num_decoder_layers, num_queries, 92, nb_class, seed,
                                 tf_backbone=tf_backbone)
        if precision not in ("fp32", "bf16", "fp32x3"):
            raise ValueError("precision must be 'fp32' (exact fp32 MFMA, parity mode), 'fp32x3' (fp32 storage and accuracy on the bf16 matrix "
                             "pipe: 3-way operand split, detr_gemm_desc.compute = 2) or 'bf16' (bf16 MFMA, fp32 accumulate)")
        self.engine.compute = 1 if precision == "bf16" else 0
        self.engine.f32_split = precision == "fp32x3"
        self.precision = precision
        self.engine.dropout_p = float(dropout)  # Transformer(dropout=0.1), applied when called with training=True
        self.dp = None                        # parallel.DataParallel when training on several GPUs
        self.device = self.engine.device

    # ---- Keras-like surface used by the reference's scripts / optimizers.py ----------------
    @property
    def trainable_variables(self):
        return list(self.engine.P.views.values())

    @property
    def layers(self):
        tops = []
        for k in self.engine.P.shapes:
            t = k.split("/", 1)[0]
            if t not in tops:
                tops.append(t)
        return [_Layer(t, [v for k, v in self.engine.P.views.items() if k.split("/", 1)[0] == t]) for t in tops]

    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError(f"No such layer: {name}")

    def summary(self):
        P = self.engine.P
        n_train = sum(n for _, n in P.offsets.values())
        n_frozen = sum(4 * c for c in P.bn.values())
        print(f'Model: "{self.name}"  trainable params: {n_train:,}  non-trainable (frozen BN): {n_frozen:,}')
        for l in self.layers:
            print(f"  {l.name:28s} {sum(v.numel() for v in l.trainable_variables):>12,}")

    @staticmethod
    def _npz(path):
        return path if path.endswith(".npz") else path + ".npz"      # np.savez appends the suffix: keep save / load symmetric

    def wanted_shapes(self):
        """{parameter name: shape} of everything a weight file may provide (trainable tensors + frozen-BN vectors)."""
        P = self.engine.P
        out = dict(P.shapes)
        for p, c in P.bn.items():
            for leaf in P.bn_leaves:
                out[f"{p}/{leaf}"] = (c,)
        return out

    def load_weights(self, path_or_dict):
        """A parameter dict, this package's `.npz` file, or a TensorFlow checkpoint prefix (`x.ckpt` with `x.ckpt.index` next to
        it: the reference's own weight files, weights.py:33 -- read without TensorFlow).  Returns the names left unset."""
        # (ParamStore.load / load_dict notify the engine: frozen-BN refold + weights-version bump)
        if isinstance(path_or_dict, str) and os.path.exists(path_or_dict + ".index"):
            from .weights import load_tf_checkpoint_params
            params, unused = load_tf_checkpoint_params(path_or_dict, self.wanted_shapes())
            missing = self.engine.load_params(params)
            # `.expect_partial()` of the reference (weights.py:35) covers the fine-tuning heads only: everything else of the
            # network must come out of the file, or the model silently keeps its random initialisation
            heads = ("class_embed", "bbox_embed", "cls_layer", "pos_layer")
            lost = sorted(n for n in (missing or ()) if not n.startswith(heads))
            self.last_load_report = dict(missing=sorted(missing or ()), unused=sorted(unused))
            if lost:
                shown = ", ".join(lost[:8]) + (" ..." if len(lost) > 8 else "")
                extra = ", ".join(sorted(unused)[:8]) + (" ..." if len(unused) > 8 else "")
                raise ValueError(f"{path_or_dict}: {len(lost)} network parameters were not found in the checkpoint ({shown}); "
                                 f"{len(unused)} checkpoint variables matched no parameter by name and shape ({extra})")
            if unused:
                import warnings
                warnings.warn(f"{path_or_dict}: {len(unused)} checkpoint variables were not used: " +
                              ", ".join(sorted(unused)[:8]) + (" ..." if len(unused) > 8 else ""))
            return missing
        return self.engine.P.load(self._npz(path_or_dict)) if isinstance(path_or_dict, str) else self.engine.load_params(path_or_dict)

    def save_weights(self, path):
        self.engine.P.save(self._npz(path))

    # ---- forward ----------------------------------------------------------------------------
    def _eval_forward(self, images):
        """Eval-mode forward.  The launch sequence of a fixed input shape is static, so the second call with the same shape
        (and the same weights version) records it as a hipGraph and later calls replay it: a single 480x640 image is ~300
        launches whose host cost (ctypes, ~10 us each) exceeds their GPU time.  DETR_HIP_GRAPH=0 / model.eval_graph = False:
        always eager.  Outputs are views of the engine's buffers either way (the caller clones)."""
        eng = self.engine
        if not (self.eval_graph and os.environ.get("DETR_HIP_GRAPH", "1") != "0" and images.is_cuda):
            return eng.forward(images, training=False)
        key = (tuple(images.shape), eng._weights_version, eng.compute)
        g = self._eval_graph
        # a graph addresses the engine's buffers as they were at capture: any later re-allocation (a forward of another
        # shape in between) makes it unusable -- engine.buf_generation tells
        if g is not None and g["key"] == key and g["gen"] == eng.buf_generation:
            g["static"].copy_(images)
            g["graph"].replay()
            return g["out"]
        if g is not None and g["gen"] != eng.buf_generation:
            self._eval_graph = g = None
        if self._eval_seen != (key, eng.buf_generation):
            # first sighting of this shape with these buffers: eager (allocates the buffers, refreshes derived weight copies)
            out = eng.forward(images, training=False)
            self._eval_seen = ((tuple(images.shape), eng._weights_version, eng.compute), eng.buf_generation)
            return out
        static = g["static"] if g is not None and g["static"].shape == images.shape else torch.empty_like(images)
        static.copy_(images)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):       # (see training._Segments.begin)
            out = eng.forward(static, training=False)
        self._eval_graph = dict(key=key, static=static, graph=graph, out=out, gen=eng.buf_generation)
        graph.replay()
        return out

    def __call__(self, images, training=False):
        if isinstance(images, np.ndarray):
            images = torch.from_numpy(images)
        images = images.to(device=self.device, dtype=torch.float32)
        logits, boxes = self.engine.forward(images, training=True) if training else self._eval_forward(images)
        if self.headless:
            hs = self.engine._bufs["dec:hs"].view(self.engine.num_dec, images.shape[0], self.engine.Q, 256)
            return hs if training else hs.clone()
        if not training:
            # inference outputs are often kept across calls (predictions accumulated for mAP, train vs val outputs): hand out
            # fresh tensors like Keras does.  Training outputs stay views of the engine's buffers (consumed by get_losses
            # before the next forward; no allocation in the step).
            logits, boxes = logits.clone(), boxes.clone()
        out = DetrOutputs()
        out["pred_logits"], out["pred_boxes"] = logits[-1], boxes[-1]
        out["aux"] = [{"pred_logits": logits[i], "pred_boxes": boxes[i]} for i in range(logits.shape[0] - 1)]
        out.levels_logits, out.levels_boxes, out.model = logits, boxes, self
        if self.dp is not None:
            out.reduce_sums = self.dp.reduce_sums
        return out


def get_detr_model(config, include_top=False, nb_class=None, weights=None, tf_backbone=False, num_decoder_layers=6,
                   num_encoder_layers=6, num_queries=100, backbone="resnet50", device=None, seed=0, dropout=0.1,
                   precision="fp32"):
    """Same arguments and three output modes as the reference (detr.py:116-204); `num_queries`,
    `backbone` ("resnet50" | "resnet101", resnet_backbone.py:35-66), `device` and `seed` are
    extensions (the reference never exposes num_queries / ResNet101); `dropout` is the
    transformer dropout rate of training mode (the reference hard-codes 0.1, transformer.py:9)."""
    if tf_backbone:
        # detr.py:146-148: the backbone becomes tf.keras.applications.ResNet50 (ResNet v1, conv biases, BatchNormalization in
        # inference mode) and the data pipeline switches to the caffe-style BGR mean subtraction.  (The reference also
        # downloads the ImageNet weights; here the backbone starts from the seeded random init unless `weights` is given.)
        if backbone != "resnet50":
            raise ValueError("tf_backbone=True is tf.keras.applications.ResNet50")
        config.normalized_method = "tf_resnet"
    model = DetrModel(include_top=include_top, nb_class=nb_class, num_decoder_layers=num_decoder_layers,
                      num_encoder_layers=num_encoder_layers, num_queries=num_queries, backbone=backbone, device=device,
                      seed=seed, dropout=dropout, precision=precision, tf_backbone=tf_backbone)
    if weights is not None:
        if isinstance(weights, str) and weights == "detr":
            # the reference downloads three files from GCS into weights/detr/ (weights.py:5-11,24-32) and loads
            # weights/detr/detr.ckpt; there is no network here, but files placed there are read (networks/tf_checkpoint.py)
            weights = os.path.join("weights", "detr", "detr.ckpt")
            if not os.path.exists(weights + ".index"):
                raise FileNotFoundError(
                    f'weights="detr": {weights}.index not found.  The reference fetches checkpoint / detr.ckpt.index / '
                    "detr.ckpt.data-00000-of-00001 from https://storage.googleapis.com/visualbehavior-publicweights/detr/ "
                    "(weights.py:5-11); put them under weights/detr/, or convert the original PyTorch detr-r50 state-dict with "
                    "`python -m detr_tf.networks.weights in.pth out.npz` and pass the .npz path")
            model.load_weights(weights)          # (`.expect_partial()`, weights.py:35: fine-tuning heads are not in the file)
            weights = None
    if weights is not None:
        model.load_weights(weights)
    if include_top is False and nb_class is not None:
        config.add_nlayers([_Layer("cls_layer", None), _Layer("pos_layer", None)])      # detr.py:103
    return model
