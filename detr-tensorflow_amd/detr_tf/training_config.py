"""Training configuration of the drop-in API.

Same attribute names, defaults and command-line flags as the reference's
detr_tf/training_config.py (:6-38 flags, :41-77 attributes, :79-103 helpers, :106-112
DataConfig), but plain Python: the learning rates are `LRVar` cells (the reference uses
tf.Variable so that scripts can `.assign()` new values between epochs, finetune_voc.py:76-96).
"""
import argparse
import os


class LRVar:
    """Mutable scalar read at every optimiser step (stand-in for tf.Variable(lr))."""

    def __init__(self, value):
        self.value = float(value)

    def assign(self, value):
        self.value = float(value)
        return self

    def numpy(self):
        return self.value

    def __float__(self):
        return self.value

    def __repr__(self):
        return f"LRVar({self.value})"


# flag, type, default, is_switch   (the reference types the three lr flags as bool -- a defect;
# they are floats here)
_FLAGS = (
    ("data_dir", str, None, False), ("img_dir", str, None, False), ("ann_file", str, None, False),
    ("ann_dir", str, None, False), ("background_class", int, 0, False),
    ("train_backbone", None, False, True), ("train_transformers", None, False, True),
    ("train_nlayers", None, False, True), ("finetuning", None, False, True),
    ("batch_size", int, 1, False), ("gradient_norm_clipping", float, 0.1, False),
    ("target_batch", int, None, False),
    ("backbone_lr", float, 1e-5, False), ("transformers_lr", float, 1e-4, False), ("nlayers_lr", float, 1e-4, False),
    ("log", None, False, True),
)


def training_config_parser():
    parser = argparse.ArgumentParser(description="DETR training options (override TrainingConfig attributes)")
    for name, typ, default, switch in _FLAGS:
        if switch:
            parser.add_argument(f"--{name}", action="store_true", default=default, required=False)
        else:
            parser.add_argument(f"--{name}", type=typ, default=default, required=False)
    return parser


_DEFAULTS = dict(
    data_dir=None, img_dir=None, ann_dir=None, ann_file=None,
    background_class=0, image_size=(376, 672),
    train_backbone=False, train_transformers=False, train_nlayers=False,
    finetuning=False, batch_size=1, gradient_norm_clipping=0.1, target_batch=1,
    global_step=0, log=False, normalized_method="torch_resnet",
)


class TrainingConfig:
    def __init__(self):
        for k, v in _DEFAULTS.items():
            setattr(self, k, v)
        self.data = DataConfig()
        self.backbone_lr = LRVar(1e-5)
        self.transformers_lr = LRVar(1e-4)
        self.nlayers_lr = LRVar(1e-4)
        self.nlayers = []

    def add_nlayers(self, layers):
        """Register the names of the freshly added head layers (they form the 'nlayers' group)."""
        self.nlayers = [layer.name for layer in layers]

    def update_from_args(self, args):
        for key, value in vars(args).items():
            cur = getattr(self, key, None)
            if isinstance(cur, LRVar):
                cur.assign(value)
            else:
                setattr(self, key, value)
        self.data = DataConfig(self.data_dir, self.img_dir, self.ann_file, self.ann_dir)


class DataConfig:
    """Joins the dataset paths (reference :106-112)."""

    def __init__(self, data_dir=None, img_dir=None, ann_file=None, ann_dir=None):
        def join(leaf):
            return os.path.join(data_dir, leaf) if (data_dir is not None and leaf is not None) else None
        self.data_dir = data_dir
        self.img_dir = join(img_dir)
        self.ann_file = join(ann_file)
        self.ann_dir = join(ann_dir)
