"""Throw-away torch-backed stand-in for the TensorFlow / Keras symbols the reference's hot-path modules touch.

BUILD-CONTAINER ONLY (never imported by the package, the tests or the GPU box).  TensorFlow cannot be installed in this
image, so the reference's Python cannot run as published.  This module lets `scripts/crosscheck_reference.py` execute
the reference's OWN source files from /root/reference -- detr_tf/networks/{detr,resnet_backbone,transformer,
position_embeddings,custom_layers}.py, detr_tf/loss/{loss,hungarian_matching}.py, detr_tf/bbox.py, detr_tf/inference.py,
detr_tf/optimizers.py, detr_tf/training.py -- with every `tf.*` call mapped onto the torch-CPU op of the same documented
semantics.  It is OUR code: agreement between the reference-under-shim and oracle/*.py shows that the restatement follows
the reference's control flow (padding, strides, scale-after-bias, the double name swap, offsets, weights, head wiring,
aux ordering, dropout placement, group partition, accumulate cadence) -- it is NOT an independent TensorFlow oracle, and
DESIGN.md keeps "parity unpinned at the TF boundary".

Tensors are plain torch.Tensors; tf.Variable / add_weight return leaf tensors looked up BY NAME in `PARAMS`
(name = '/'.join(layer-name scope) + '/' + weight name, the Keras naming of SURVEY.md A.6), so the reference model runs
on exactly the weights the oracle and the HIP path use.  Functional-API construction (tf.keras.Input -> layers ->
tf.keras.Model(inputs, outputs)) is executed eagerly on a CONCRETE input (`STATE.input`): the resulting Model object holds
the outputs for that input; calling it again re-runs its builder (see crosscheck_reference.py).
"""
import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F


class _State:
    params = None            # name -> torch leaf tensor
    input = None             # concrete tensor handed out by tf.keras.Input
    training = False         # default `training` of a top-level call
    drop_hook = None         # callable(site_scope, call_index, x, kind) -> x   (training-mode dropout masks)
    scope = []               # layer-name stack
    train_ctx = []           # training-flag stack (Keras call-context propagation)
    top_calls = []           # layers called at scope depth 0 (functional-model bookkeeping)
    marker = 0               # index into top_calls where the current functional model starts
    drop_counters = {}
    uid = {}


STATE = _State()


def reset(params=None, input=None, training=False, drop_hook=None):
    STATE.params, STATE.input, STATE.training, STATE.drop_hook = params, input, training, drop_hook
    STATE.scope, STATE.train_ctx, STATE.top_calls, STATE.marker = [], [], [], 0
    STATE.drop_counters, STATE.uid = {}, {}


class TFTensor(torch.Tensor):
    """torch.Tensor whose augmented assignments REBIND like TensorFlow's immutable tensors (`source += ...`,
    `WQ *= ...` in transformer.py would otherwise modify a tensor another layer still needs)."""

    def __iadd__(self, other):
        return torch.add(self, other)

    def __isub__(self, other):
        return torch.sub(self, other)

    def __imul__(self, other):
        return torch.mul(self, other)

    def __itruediv__(self, other):
        return torch.div(self, other)

    def __format__(self, spec):              # f"{log['label_cost']:.2f}" in training.py:60 (eager tensors format like floats)
        if self.dim() == 0:
            return format(self.item(), spec)
        return object.__format__(self, spec)


def tft(x, requires_grad=False):
    t = torch.as_tensor(x).detach().clone().as_subclass(TFTensor)
    if requires_grad:
        t.requires_grad_(True)
    return t


def _t(x, dtype=None):
    if isinstance(x, torch.Tensor):
        return x if dtype is None else x.to(dtype)
    return torch.as_tensor(np.asarray(x) if not isinstance(x, (int, float, bool)) else x, dtype=dtype)


def _int(x):
    return int(x.item()) if isinstance(x, torch.Tensor) else int(x)


# ----------------------------------------------------------------------------------------------------
# tf.* ops
# ----------------------------------------------------------------------------------------------------
tf = types.ModuleType("tensorflow")
tf.__version__ = "2.3.0"
tf.float32, tf.float64, tf.int32, tf.int64, tf.bool = torch.float32, torch.float64, torch.int32, torch.int64, torch.bool
tf.newaxis = None
tf.Tensor = torch.Tensor


def _cast(x, dtype):
    return _t(x).to(dtype)


def _shape(x):
    return list(_t(x).shape)


def _slice(x, begin, size):
    idx = []
    for b, s in zip(begin, size):
        b, s = _int(b), _int(s)
        idx.append(slice(b, None if s == -1 else b + s))
    return x[tuple(idx)]


def _tile(x, multiples):
    return x.repeat(*[_int(m) for m in multiples])


def _gather(params, indices, axis=0):
    return torch.index_select(params, axis, _t(indices).long().reshape(-1)).reshape(
        tuple(params.shape[:axis]) + tuple(_t(indices).shape) + tuple(params.shape[axis + 1:]))


def _where(cond, x=None, y=None):
    if x is None:
        return torch.nonzero(cond)
    return torch.where(cond, x, y)


def _numpy_function(fn, inp, Tout):
    outs = fn(*[i.detach().cpu().numpy() for i in inp])
    return [torch.as_tensor(np.asarray(o)).to(dt) for o, dt in zip(outs, Tout)]


def _zeros(shape, dtype=torch.float32):
    return torch.zeros([_int(s) for s in shape], dtype=dtype)


def _reshape(x, shape):
    return x.reshape([_int(s) for s in shape])


def _matmul(a, b, transpose_a=False, transpose_b=False):
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return a @ b


def _range(*a, dtype=None):
    return torch.arange(*a, dtype=dtype)


def _norm(x, ord=2, axis=None):
    assert ord == 1
    return x.abs().sum(axis)


def _sparse_ce(labels, logits, name=None):
    return F.cross_entropy(logits, labels.long(), reduction="none")


tf.cast, tf.shape, tf.slice, tf.tile, tf.gather, tf.where = _cast, _shape, _slice, _tile, _gather, _where
tf.squeeze = lambda x, axis=None: x.squeeze() if axis is None else x.squeeze(axis)
tf.expand_dims = lambda x, axis: _t(x).unsqueeze(axis)
tf.concat = lambda xs, axis: torch.cat(list(xs), dim=axis)
tf.stack = lambda xs, axis=0: torch.stack(list(xs), dim=axis)
tf.zeros, tf.zeros_like, tf.reshape = _zeros, torch.zeros_like, _reshape
tf.transpose = lambda x, perm: x.permute(*perm)
tf.matmul, tf.range, tf.norm = _matmul, _range, _norm
tf.argmax = lambda x, axis=None: torch.argmax(x, dim=axis)
tf.reduce_mean = lambda x, axis=None: x.mean() if axis is None else x.mean(axis)
tf.reduce_sum = lambda x, axis=None: x.sum() if axis is None else x.sum(axis)
tf.reduce_max = lambda x, axis=None: x.max() if axis is None else x.max(axis).values
tf.abs, tf.sigmoid = torch.abs, torch.sigmoid
tf.clip_by_value = lambda x, lo, hi: torch.clamp(x, lo, hi)
tf.numpy_function = _numpy_function
tf.pad = lambda x, paddings, mode="CONSTANT", constant_values=0: F.pad(
    _t(x), [int(v) for pr in reversed([[_int(a), _int(b)] for a, b in paddings]) for v in pr], value=constant_values)
tf.constant = lambda v, dtype=None: _t(np.asarray(v), dtype)
tf.function = lambda f=None, **kw: f if f is not None else (lambda g: g)

tf.math = types.ModuleType("tensorflow.math")
tf.math.minimum, tf.math.maximum, tf.math.rsqrt, tf.math.log = torch.minimum, torch.maximum, torch.rsqrt, torch.log
tf.math.sin, tf.math.cos = torch.sin, torch.cos
tf.math.cumsum = lambda x, axis=0: torch.cumsum(x, dim=axis)
tf.nn = types.ModuleType("tensorflow.nn")
tf.nn.relu = torch.relu
tf.nn.softmax = lambda x, axis=-1: torch.softmax(x, dim=axis)
tf.nn.sparse_softmax_cross_entropy_with_logits = _sparse_ce
tf.linalg = types.ModuleType("tensorflow.linalg")
tf.linalg.diag_part = torch.diagonal


def _resize_nn(images, size, align_corners=False, half_pixel_centers=False):
    """tf.compat.v1.image.resize_nearest_neighbor (legacy: src = floor(dst * in / out))."""
    assert not align_corners and not half_pixel_centers
    H, W = images.shape[1], images.shape[2]
    oh, ow = _int(size[0]), _int(size[1])
    iy = torch.floor(torch.arange(oh, dtype=torch.float32) * (H / oh)).long().clamp(max=H - 1)
    ix = torch.floor(torch.arange(ow, dtype=torch.float32) * (W / ow)).long().clamp(max=W - 1)
    return images[:, iy][:, :, ix]


tf.autograph = types.ModuleType("tensorflow.autograph")            # decorators of the (unused) wandb logger, evaluated at import
tf.autograph.experimental = types.ModuleType("tensorflow.autograph.experimental")
tf.autograph.experimental.do_not_convert = lambda f=None: (f if f is not None else (lambda g: g))
tf.compat = types.ModuleType("tensorflow.compat")
tf.compat.v1 = types.ModuleType("tensorflow.compat.v1")
tf.compat.v1.image = types.ModuleType("tensorflow.compat.v1.image")
tf.compat.v1.image.resize_nearest_neighbor = _resize_nn


class ScalarVariable:
    """tf.Variable(<python scalar>): the learning-rate cells of TrainingConfig (training_config.py:63-65)."""

    def __init__(self, v):
        self.v = float(v)

    def assign(self, v):
        self.v = float(v)
        return self

    def numpy(self):
        return self.v

    def __float__(self):
        return self.v


def _variable(v, **kw):
    if isinstance(v, (int, float)):
        return ScalarVariable(v)
    return tft(v, requires_grad=True)


tf.Variable = _variable


class GradientTape:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def gradient(self, target, sources):
        grads = torch.autograd.grad(target, sources, allow_unused=True, retain_graph=True)
        return list(grads)


tf.GradientTape = GradientTape

# ----------------------------------------------------------------------------------------------------
# Keras
# ----------------------------------------------------------------------------------------------------
keras = types.ModuleType("tensorflow.keras")
layers = types.ModuleType("tensorflow.keras.layers")
models = types.ModuleType("tensorflow.keras.models")
initializers = types.ModuleType("tensorflow.keras.initializers")
optimizers = types.ModuleType("tensorflow.keras.optimizers")
applications = types.ModuleType("tensorflow.keras.applications")
initializers.GlorotUniform = lambda *a, **k: "glorot_uniform"


def _snake(name):
    out = ""
    for i, ch in enumerate(name):
        if ch.isupper() and i and (name[i - 1].islower() or (i + 1 < len(name) and name[i + 1].islower())):
            out += "_"
        out += ch.lower()
    return out


class Layer:
    def __init__(self, name=None, trainable=True, **kwargs):
        if name is None:
            base = _snake(type(self).__name__)
            n = STATE.uid.get(base, 0)
            STATE.uid[base] = n + 1
            name = base if n == 0 else f"{base}_{n}"
        self.name = name
        self.trainable = trainable
        self.built = False
        self._own_weights = []
        import inspect
        self._call_has_training = "training" in inspect.signature(self.call).parameters

    # -- weights ------------------------------------------------------------------------------------
    def add_weight(self, name=None, shape=None, initializer=None, trainable=True, dtype=None, **kw):
        full = "/".join(STATE.scope + [name])
        if STATE.params is None or full not in STATE.params:
            raise KeyError(f"tf_shim: no value for weight '{full}' (known: {len(STATE.params or {})} tensors)")
        w = STATE.params[full]
        assert tuple(w.shape) == tuple(int(s) for s in shape), (full, tuple(w.shape), tuple(shape))
        w._shim_name, w._shim_trainable = full, bool(trainable)
        self._own_weights.append(w)
        return w

    def build(self, input_shape):
        pass

    def _sublayers(self):
        out = []
        for v in self.__dict__.values():
            if isinstance(v, Layer):
                out.append(v)
            elif isinstance(v, (list, tuple)):
                out.extend(x for x in v if isinstance(x, Layer))
        return out

    @property
    def layers(self):
        return self._sublayers()

    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError(f"No such layer: {name}")

    @property
    def trainable_variables(self):
        out = [w for w in self._own_weights if w._shim_trainable and self.trainable]
        for l in self._sublayers():
            out += l.trainable_variables
        return out

    # -- call ---------------------------------------------------------------------------------------
    def __call__(self, *args, **kwargs):
        top = len(STATE.scope) == 0
        if top:
            STATE.top_calls.append(self)
        STATE.scope.append(self.name)
        # Keras call-context rule: an explicit `training` wins, otherwise the value of the enclosing call, otherwise False
        if "training" in kwargs and kwargs["training"] is not None:
            tr = bool(kwargs["training"])
        elif STATE.train_ctx:
            tr = STATE.train_ctx[-1]
        else:
            tr = bool(STATE.training)
        if self._call_has_training:
            kwargs["training"] = tr
        else:
            kwargs.pop("training", None)
        STATE.train_ctx.append(tr)
        try:
            if not self.built:
                first = args[0] if args else None
                if isinstance(first, (list, tuple)):
                    shp = [tuple(t.shape) for t in first]
                elif isinstance(first, torch.Tensor):
                    shp = tuple(first.shape)
                else:
                    shp = None
                self.build(shp)
                self.built = True
            return self.call(*args, **kwargs)
        finally:
            STATE.scope.pop()
            STATE.train_ctx.pop()

    def call(self, *a, **k):
        raise NotImplementedError


class InputLayer(Layer):
    def call(self, x):
        return x


def Input(shape=None, **kw):
    assert STATE.input is not None, "tf_shim: set STATE.input (the concrete batch) before building a functional model"
    STATE.marker = len(STATE.top_calls)
    return STATE.input


class Model(Layer):
    """Subclassed models behave like layers.  Model(inputs, outputs, name=...) is the functional form: it holds the
    outputs computed (eagerly) for the concrete `inputs` and the layers called since the last Input() / Model()."""

    def __init__(self, *args, **kwargs):
        if args or "inputs" in kwargs:
            inputs = args[0] if args else kwargs.pop("inputs")
            outputs = args[1] if len(args) > 1 else kwargs.pop("outputs")
            super().__init__(name=kwargs.get("name"))
            self._functional = True
            self._inputs, self._outputs = inputs, outputs
            seen, used = set(), []
            for l in STATE.top_calls[STATE.marker:]:
                if id(l) not in seen:
                    seen.add(id(l))
                    used.append(l)
            self._layers = [InputLayer(name="input")] + used
            STATE.marker = len(STATE.top_calls)
            self.built = True
        else:
            super().__init__(**kwargs)
            self._functional = False

    @property
    def layers(self):
        return self._layers if getattr(self, "_functional", False) else self._sublayers()

    @property
    def trainable_variables(self):
        if getattr(self, "_functional", False):
            out = []
            for l in self._layers:
                out += l.trainable_variables
            return out
        return Layer.trainable_variables.fget(self)

    def call(self, x, training=None):
        if getattr(self, "_functional", False):
            assert x is self._inputs, "tf_shim: a functional model only holds the outputs of the batch it was built on"
            return self._outputs
        raise NotImplementedError

    def summary(self):
        pass


class Sequential(Model):
    def __init__(self, layer_list, name=None):
        Layer.__init__(self, name=name)
        self._functional = False
        self.seq = list(layer_list)
        for i, l in enumerate(self.seq):          # Keras would name them dense, dense_1, ...: index them inside the model
            l.name = f"{_snake(type(l).__name__)}_{i}"

    def call(self, x):
        for l in self.seq:
            x = l(x)
        return x


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding="valid", use_bias=True, dilation_rate=1, **kw):
        super().__init__(**kw)
        assert padding == "valid"
        self.filters, self.k, self.s, self.d, self.use_bias = filters, kernel_size, strides, dilation_rate, use_bias

    def build(self, shape):
        self.kernel = self.add_weight(name="kernel", shape=[self.k, self.k, shape[-1], self.filters])
        self.bias = self.add_weight(name="bias", shape=[self.filters]) if self.use_bias else None

    def call(self, x):
        y = F.conv2d(x.permute(0, 3, 1, 2), self.kernel.permute(3, 2, 0, 1), self.bias, stride=self.s, padding=0, dilation=self.d)
        return y.permute(0, 2, 3, 1)


class ZeroPadding2D(Layer):
    def __init__(self, padding=1, **kw):
        super().__init__(**kw)
        self.p = padding

    def call(self, x):
        return F.pad(x, (0, 0, self.p, self.p, self.p, self.p))


class ReLU(Layer):
    def call(self, x):
        return torch.relu(x)


class Activation(Layer):
    def __init__(self, activation, **kw):
        super().__init__(**kw)
        assert activation == "relu"

    def call(self, x):
        return torch.relu(x)


class MaxPool2D(Layer):
    def __init__(self, pool_size=2, strides=None, padding="valid", **kw):
        super().__init__(**kw)
        assert padding == "valid"
        self.k, self.s = pool_size, strides or pool_size

    def call(self, x):
        return F.max_pool2d(x.permute(0, 3, 1, 2), self.k, self.s).permute(0, 2, 3, 1)


class Dropout(Layer):
    """Keras Dropout: identity unless training.  TensorFlow's RNG stream cannot be reproduced, so in training mode the mask
    comes from `STATE.drop_hook(scope, k, x)` where `scope` is the layer-name path of the call and k counts the Dropout
    calls inside the enclosing encoder / decoder layer in CALL ORDER -- i.e. the reference's own control flow decides
    WHICH tensors are dropped and in which order; only the bits come from the build's counter hash."""

    def __init__(self, rate=0.0, **kw):
        super().__init__(**kw)
        self.rate = rate

    def call(self, x, training=None):
        if not training or self.rate <= 0.0 or STATE.drop_hook is None:
            return x
        scope = list(STATE.scope[:-1])
        owner = None
        for i, s in enumerate(scope):
            if s.startswith("layer_") and i > 0 and scope[i - 1] in ("encoder", "decoder"):
                owner = "/".join(scope[:i + 1])
        assert owner is not None, scope
        k = STATE.drop_counters.get(owner, 0)
        STATE.drop_counters[owner] = k + 1
        return STATE.drop_hook(owner, k, x, scope, self.rate)


class LayerNormalization(Layer):
    def __init__(self, epsilon=1e-3, **kw):
        super().__init__(**kw)
        self.eps = epsilon

    def build(self, shape):
        self.gamma = self.add_weight(name="gamma", shape=[shape[-1]])
        self.beta = self.add_weight(name="beta", shape=[shape[-1]])

    def call(self, x):
        return F.layer_norm(x, (x.shape[-1],), self.gamma, self.beta, self.eps)


class Dense(Layer):
    def __init__(self, units, activation=None, **kw):
        super().__init__(**kw)
        self.units, self.act = units, activation

    def build(self, shape):
        self.kernel = self.add_weight(name="kernel", shape=[shape[-1], self.units])
        self.bias = self.add_weight(name="bias", shape=[self.units])

    def call(self, x):
        y = x @ self.kernel + self.bias
        if self.act == "relu":
            y = torch.relu(y)
        elif self.act == "sigmoid":
            y = torch.sigmoid(y)
        else:
            assert self.act is None
        return y


class BatchNormalization(Layer):
    pass


for _n, _c in dict(Layer=Layer, Conv2D=Conv2D, ZeroPadding2D=ZeroPadding2D, ReLU=ReLU, Activation=Activation, MaxPool2D=MaxPool2D,
                   Dropout=Dropout, LayerNormalization=LayerNormalization, Dense=Dense, BatchNormalization=BatchNormalization).items():
    setattr(layers, _n, _c)
models.Sequential, models.Model = Sequential, Model
keras.Model, keras.Input, keras.layers, keras.models, keras.initializers = Model, Input, layers, models, initializers
keras.optimizers, keras.applications = optimizers, applications


class Adam:
    """tf.keras.optimizers.Adam(learning_rate=<callable>, clipnorm=c) as documented for TF 2.3: every gradient is clipped
    with tf.clip_by_norm(g, c) (per tensor), then m, v, and var -= lr_t * m / (sqrt(v) + eps) with
    lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), eps = 1e-7.  (Our restatement of third-party Keras code.)"""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=None, **kw):
        self.lr, self.b1, self.b2, self.eps, self.clipnorm = learning_rate, beta_1, beta_2, epsilon, clipnorm
        self.iterations = 0
        self.slots = {}

    def _lr(self):
        v = self.lr() if callable(self.lr) else self.lr
        return float(v)

    def _serialize_hyperparameter(self, name):
        assert name == "learning_rate"
        return self._lr()

    def apply_gradients(self, grads_and_vars):
        self.iterations += 1
        t = self.iterations
        lr_t = self._lr() * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        with torch.no_grad():
            for g, var in grads_and_vars:
                if g is None:
                    continue
                g = g.to(torch.float32)
                if self.clipnorm is not None:
                    n = torch.sqrt((g * g).sum())
                    g = g * self.clipnorm / torch.maximum(n, torch.tensor(self.clipnorm))
                m, v = self.slots.setdefault(id(var), (torch.zeros_like(var), torch.zeros_like(var)))
                m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
                var.sub_(lr_t * m / (torch.sqrt(v) + self.eps))


optimizers.Adam = Adam
tf.keras = keras


class _Permissive(types.ModuleType):
    """cv2 / wandb / matplotlib / requests ...: imported at module level by the reference, never used on the hot path."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Permissive(f"{self.__name__}.{name}")

    def __call__(self, *a, **k):
        return None


class _PermissiveFinder:
    """Any submodule of a permissive root (skimage.color, imgaug.augmentables.segmaps, pycocotools.coco ...) resolves to
    another permissive module."""
    ROOTS = ("cv2", "wandb", "imgaug", "imageio", "pycocotools", "skimage", "matplotlib", "requests")

    def find_spec(self, name, path=None, target=None):
        import importlib.machinery
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Permissive(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install():
    """Put the stand-ins into sys.modules and /root/reference on sys.path (build container only)."""
    mods = {"tensorflow": tf, "tensorflow.keras": keras, "tensorflow.keras.layers": layers, "tensorflow.keras.models": models,
            "tensorflow.keras.initializers": initializers, "tensorflow.keras.optimizers": optimizers,
            "tensorflow.keras.applications": applications, "tensorflow.math": tf.math, "tensorflow.nn": tf.nn,
            "tensorflow.linalg": tf.linalg, "tensorflow.compat": tf.compat, "tensorflow.compat.v1": tf.compat.v1}
    sys.modules.update(mods)
    for n in _PermissiveFinder.ROOTS:            # (matplotlib is really installed here: the stand-in keeps the import cheap)
        sys.modules.pop(n, None)
    sys.meta_path.insert(0, _PermissiveFinder())
    if not hasattr(np, "bool"):
        np.bool = bool                       # hungarian_matching.py:37,41 predates numpy 1.24
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
