"""the device-side input stage.  normalisation and the padded target layout are pinned by the outputs
of the reference's own detr_tf/data/processing.py (fixture refpy_input.npz, scripts/crosscheck_reference.py); the resize
restates third-party cv2 / imgaug behaviour (absent here: "parity unpinned") and is compared with oracle/input_ref.py."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with np.load(os.path.join(GOLD, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


class _Cfg:
    def __init__(self, method="torch_resnet", size=(376, 672)):
        self.normalized_method, self.image_size = method, size


def test_host_functions_and_lookup_table_equal_reference_code_outputs():
    from detr_tf.data.processing import normalization_table, normalized_images, pad_labels
    from oracle import input_ref as I
    fx = _load("refpy_input.npz")
    for method in ("torch_resnet", "tf_resnet"):
        cfg = _Cfg(method)
        for name in ("img", "ramp"):
            assert np.array_equal(normalized_images(fx[name], cfg), fx[f"norm_{method}_{name}"])
            assert np.array_equal(I.normalized_images(fx[name], method), fx[f"norm_{method}_{name}"])
        lut, perm = normalization_table(method)          # the device path: lut[c][value of source channel perm[c]]
        ramp = fx["ramp"]                                # [256, 1, 3], every value in every channel
        got = np.stack([lut[c][ramp[:, 0, perm[c]]] for c in range(3)], -1)[:, None, :]
        assert np.array_equal(got, fx[f"norm_{method}_ramp"])
    for ci in range(int(fx["n_pad"])):
        _, b, c = pad_labels(None, fx[f"pad{ci}_in_bbox"], fx[f"pad{ci}_in_class"])
        assert b.dtype == np.float32 and c.dtype == np.int64
        assert np.array_equal(b, fx[f"pad{ci}_bbox"]) and np.array_equal(c, fx[f"pad{ci}_class"])
    with pytest.raises(ValueError):
        pad_labels(None, np.zeros((100, 4)), np.zeros((100, 1)))
    with pytest.raises(Exception):
        normalized_images(fx["img"], _Cfg("caffe"))


def test_oracle_resize_properties():
    from oracle import input_ref as I
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (24, 31, 3)).astype(np.uint8)
    assert np.array_equal(I.resize_uint8(img, 24, 31), img)                          # identity at equal sizes
    flat = np.full((10, 12, 3), 77, np.uint8)
    for interp in ("nearest", "linear", "cubic"):
        assert np.array_equal(I.resize_uint8(flat, 23, 17, interp), np.full((23, 17, 3), 77, np.uint8))      # partition of unity
    up = I.resize_uint8(img, 48, 62, "nearest")
    assert np.array_equal(up[::2, ::2], img)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["torch_resnet", "tf_resnet"])
def test_device_normalisation_equals_reference_code_outputs(hip, method):
    from detr_tf.data.processing import DeviceInputStage
    fx = _load("refpy_input.npz")
    img = fx["img"]                                                                   # [2, 37, 53, 3] uint8
    stage = DeviceInputStage(_Cfg(method, size=img.shape[1:3]))
    assert torch.equal(stage.images(img).cpu(), torch.from_numpy(fx[f"norm_{method}_img"]))       # bit-exact
    ramp = np.ascontiguousarray(np.broadcast_to(fx["ramp"][None], (1, 256, 1, 3)))
    stage2 = DeviceInputStage(_Cfg(method, size=(256, 1)))
    assert torch.equal(stage2.images(ramp).cpu()[0], torch.from_numpy(fx[f"norm_{method}_ramp"]))


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst,interp", [((37, 53), (64, 96), "cubic"), ((480, 640), (376, 672), "cubic"), ((50, 40), (20, 30), "linear"),
                                            ((33, 47), (66, 94), "nearest")])
def test_device_resize_and_normalise_vs_oracle(hip, src, dst, interp):
    from detr_tf.data.processing import DeviceInputStage
    from oracle import input_ref as I
    rng = np.random.default_rng(src[0] + dst[1])
    batch = rng.integers(0, 256, (3,) + src + (3,)).astype(np.uint8)
    stage = DeviceInputStage(_Cfg("torch_resnet", size=dst), interpolation=interp)
    got = stage.images(batch).cpu().numpy()
    ref = np.stack([I.normalized_images(I.resize_uint8(batch[b], dst[0], dst[1], interp), "torch_resnet") for b in range(3)])
    assert got.shape == ref.shape == (3,) + dst + (3,)
    # same fp32 formula in the same order: identical grey levels except where the sum lands within rounding of x.5
    lvl = 1.0 / (255.0 * 0.224)
    diff = np.abs(got - ref)
    assert float((diff > 1e-6).mean()) < 1e-3 and float(diff.max()) < 1.6 * lvl, (float((diff > 1e-6).mean()), float(diff.max()))


@pytest.mark.gpu
def test_device_pad_labels_equals_reference_code_outputs(hip):
    from detr_tf.data.processing import DeviceInputStage
    fx = _load("refpy_input.npz")
    stage = DeviceInputStage(_Cfg())
    n = int(fx["n_pad"])
    tb, tc = stage.targets([fx[f"pad{i}_in_bbox"] for i in range(n)], [fx[f"pad{i}_in_class"] for i in range(n)])
    assert tuple(tb.shape) == (n, 100, 4) and tuple(tc.shape) == (n, 100, 1) and tc.dtype == torch.int64
    for i in range(n):
        assert np.array_equal(tb[i].cpu().numpy(), fx[f"pad{i}_bbox"]) and np.array_equal(tc[i].cpu().numpy(), fx[f"pad{i}_class"])
    with pytest.raises(ValueError):
        stage.targets([np.zeros((100, 4))], [np.zeros(100)])


@pytest.mark.gpu
def test_input_stage_feeds_the_training_step(hip):
    """uint8 batch + ragged targets -> device stage -> run_train_step == the same step fed with the host-side reference
    pipeline (normalized_images + pad_labels)."""
    from detr_tf import training
    from detr_tf.data.processing import DeviceInputStage, normalized_images, pad_labels
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.optimizers import setup_optimizers
    from detr_tf.training_config import TrainingConfig
    from oracle import detr_ref as R
    rng = np.random.default_rng(5)
    cfg = TrainingConfig()
    cfg.background_class, cfg.image_size, cfg.target_batch = 91, (96, 128), None
    cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
    batch = rng.integers(0, 256, (2, 96, 128, 3)).astype(np.uint8)
    boxes = [rng.uniform(0.2, 0.6, (n, 4)).astype(np.float32) for n in (3, 6)]
    classes = [rng.integers(1, 91, n) for n in (3, 6)]
    totals = []
    for device_stage in (True, False):
        model = get_detr_model(cfg, include_top=True, num_encoder_layers=1, num_decoder_layers=1, dropout=0.0)
        model.load_weights(R.make_params(2, num_enc=1, num_dec=1))
        opt = setup_optimizers(model, cfg)
        if device_stage:
            im, tb, tc = DeviceInputStage(cfg)(batch, boxes, classes)
        else:
            im = normalized_images(batch, cfg)
            padded = [pad_labels(None, b, c) for b, c in zip(boxes, classes)]
            tb, tc = np.stack([p[1] for p in padded]), np.stack([p[2] for p in padded])
        _, total, _, _ = training.run_train_step(model, im, tb, tc, opt, cfg)
        totals.append(float(total))
    assert abs(totals[0] - totals[1]) <= 1e-6 * abs(totals[1]), totals
