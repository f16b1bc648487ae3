"""ORACLE (test infrastructure only) -- CPU restatement of the reference's input stage.

normalized_images / pad_labels follow detr_tf/data/processing.py:6-21,35-55 line by line (NumPy; pinned by the outputs of
the reference's own functions, tests/golden/refpy_input.npz).  resize_uint8 restates what the reference delegates to
third-party code -- imgaug `Resize` (data/transformation.py:82-91, default interpolation "cubic") -> cv2.resize -- from
cv2's documentation: half-pixel centres, replicated border, cubic kernel with a = -0.75, fp32 arithmetic, result rounded
(half to even) and saturated to uint8.  cv2 / imgaug are not installable here: that part is PARITY UNPINNED (OpenCV's
uint8 path uses 11-bit fixed-point coefficients and may differ by one grey level)."""
import numpy as np


def normalized_images(image, method):
    if method == "torch_resnet":
        channel_avg = np.array([0.485, 0.456, 0.406])
        channel_std = np.array([0.229, 0.224, 0.225])
        image = (image / 255.0 - channel_avg) / channel_std
        return image.astype(np.float32)
    if method == "tf_resnet":
        mean = [103.939, 116.779, 123.68]
        image = image[..., ::-1]
        image = image - mean
        return image.astype(np.float32)
    raise Exception("Can't handler thid normalized method")


def pad_labels(t_bbox, t_class, rows=100):
    t_bbox = np.asarray(t_bbox, np.float32).reshape(-1, 4)
    t_class = np.asarray(t_class, np.int64).reshape(-1, 1)
    n = t_bbox.shape[0]
    header = np.zeros((1, 4), np.float32)
    header[0, 0] = n
    b = np.concatenate([header, t_bbox, np.zeros((rows - 1 - n, 4), np.float32)], 0)
    c = np.concatenate([np.zeros((1, 1), np.int64), t_class, np.zeros((rows - 1 - n, 1), np.int64)], 0)
    return b, c


def _cubic(f):
    A = np.float32(-0.75)
    f = f.astype(np.float32)
    one = np.float32(1.0)
    c0 = ((A * (f + one) - np.float32(5.0) * A) * (f + one) + np.float32(8.0) * A) * (f + one) - np.float32(4.0) * A
    c1 = ((A + np.float32(2.0)) * f - (A + np.float32(3.0))) * f * f + one
    c2 = ((A + np.float32(2.0)) * (one - f) - (A + np.float32(3.0))) * (one - f) * (one - f) + one
    c3 = one - c0 - c1 - c2
    return [c0, c1, c2, c3]


def resize_uint8(img, Hd, Wd, interpolation="cubic"):
    """[Hs, Ws, 3] uint8 -> [Hd, Wd, 3] uint8 (same operation order as csrc/input_stage.hip, fp32)."""
    Hs, Ws, _ = img.shape
    if (Hs, Ws) == (Hd, Wd):
        return img.copy()
    sy, sx = np.float32(Hs) / np.float32(Hd), np.float32(Ws) / np.float32(Wd)
    ys, xs = np.arange(Hd, dtype=np.float32), np.arange(Wd, dtype=np.float32)
    if interpolation == "nearest":
        yy = np.clip(np.floor(ys * sy).astype(np.int64), 0, Hs - 1)
        xx = np.clip(np.floor(xs * sx).astype(np.int64), 0, Ws - 1)
        return img[yy][:, xx]
    fy, fx = (ys + np.float32(0.5)) * sy - np.float32(0.5), (xs + np.float32(0.5)) * sx - np.float32(0.5)
    y0, x0 = np.floor(fy).astype(np.int64), np.floor(fx).astype(np.int64)
    if interpolation == "linear":
        cy = [np.float32(1.0) - (fy - y0.astype(np.float32)), fy - y0.astype(np.float32)]
        cx = [np.float32(1.0) - (fx - x0.astype(np.float32)), fx - x0.astype(np.float32)]
        off = 0
    else:
        cy, cx = _cubic(fy - y0.astype(np.float32)), _cubic(fx - x0.astype(np.float32))
        off = -1
    src = img.astype(np.float32)
    acc = np.zeros((Hd, Wd, 3), np.float32)
    for j, cyj in enumerate(cy):
        yy = np.clip(y0 + off + j, 0, Hs - 1)
        row = np.zeros((Hd, Wd, 3), np.float32)
        for k, cxk in enumerate(cx):
            xx = np.clip(x0 + off + k, 0, Ws - 1)
            row = row + cxk[None, :, None] * src[yy][:, xx]
        acc = acc + cyj[:, None, None] * row
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)
