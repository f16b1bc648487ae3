"""mAP evaluation loop -- what the reference's eval.py:30-61 `eval_model` does, for any batch size: forward (eval mode),
batched device post-processing (one launch per batch instead of one get_model_inference per image), header-stripped
targets converted to [y1, x1, y2, x2] (bbox.py:186-196), the vectorised COCO-style accumulator, the reference's table."""
import numpy as np
import torch

from . import bbox
from .inference import get_model_inference_batched
from .loss.compute_map import APAccumulator


def eval_model(model, config, class_names, valid_dt, max_batches=None, print_result=True):
    acc = APAccumulator(len(class_names))
    for it, (images, target_bbox, target_class) in enumerate(valid_dt):
        m_outputs = model(images, training=False)
        dets = get_model_inference_batched(m_outputs, config.background_class, bbox_format="yxyx")
        tb = torch.as_tensor(np.asarray(target_bbox) if not torch.is_tensor(target_bbox) else target_bbox).float().cpu()
        tc = torch.as_tensor(np.asarray(target_class) if not torch.is_tensor(target_class) else target_class).cpu()
        for b, (p_bbox, p_labels, p_scores) in enumerate(dets):
            n = int(tb[b, 0, 0])                                          # header row (data/processing.py:35-55)
            t_bbox = bbox.xcycwh_to_yx_min_yx_max(tb[b, 1:1 + n])
            t_class = tc[b, 1:1 + n].reshape(-1)
            acc.add_image(p_bbox.cpu().numpy(), p_labels.cpu().numpy(), p_scores.cpu().numpy(), t_bbox.numpy(), t_class.numpy())
        if max_batches is not None and it + 1 >= max_batches:
            break
    return acc.result(class_names, print_result=print_result)
