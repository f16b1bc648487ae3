"""Data-parallel equivalence on the GPU box (1 GPU): two ranks (gloo transport, both on cuda:0 -- RCCL refuses
two ranks on one device) with one image each must produce the gradients of a single-process batch of 2:
checks the all-reduce of the loss normalisers (whole-batch semantics of loss.py:66-67,82,94) and the bucketed
all-reduce(sum) of the flat gradient buffer launched from inside the backward."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(cfg_only=False):
    import sys
    for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.training_config import TrainingConfig
    from oracle import detr_ref as R, set_loss_ref as L
    cfg = TrainingConfig()
    cfg.background_class = 91
    cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
    cfg.target_batch = None
    params = R.make_params(12, num_enc=1, num_dec=2)
    model = get_detr_model(cfg, include_top=True, num_encoder_layers=1, num_decoder_layers=2, device="cuda:0", dropout=0.0)
    model.load_weights(params)
    images = np.random.default_rng(4).normal(size=(2, 96, 128, 3)).astype(np.float32)
    t_bbox, t_class = L.make_targets(2, seed=40, force_full=False)
    return cfg, model, images, t_bbox, t_class


def _rank_job(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    cfg, model, images, t_bbox, t_class = _make()
    import torch.distributed as dist
    from detr_tf import parallel, training
    from detr_tf.optimizers import setup_optimizers
    parallel.init_distributed(backend="gloo")
    model.dp = parallel.DataParallel(model.engine.P.grad, model.engine.P.bucket_bounds())
    opt = setup_optimizers(model, cfg)
    lo, hi = parallel.shard_batch(2, rank, world)
    out, total, log, steps = training.run_train_step(model, images[lo:hi], t_bbox[lo:hi], t_class[lo:hi], opt, cfg)
    torch.cuda.synchronize()
    ret[rank] = (model.engine.P.grad.cpu().numpy(), float(total), {k: float(v) for k, v in log.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_equals_single_process_batch(hip):
    from detr_tf import training
    from detr_tf.optimizers import setup_optimizers
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_job, args=(2, _free_port(), ret), nprocs=2, join=True)
    cfg, model, images, t_bbox, t_class = _make()
    opt = setup_optimizers(model, cfg)
    out, total, log, steps = training.run_train_step(model, images, t_bbox, t_class, opt, cfg)
    torch.cuda.synchronize()
    ref = model.engine.P.grad.cpu().numpy()
    g0, t0, l0 = ret[0]
    g1, t1, l1 = ret[1]
    assert np.array_equal(g0, g1), "ranks disagree after the all-reduce"
    assert abs(t0 - float(total)) < 1e-4 * abs(float(total)) and abs(t1 - t0) < 1e-6 * abs(t0)
    for k in ("label_cost", "giou_loss", "l1_loss", "l1_loss_0"):
        assert abs(l0[k] - float(log[k])) < 1e-4 * max(1.0, abs(float(log[k]))), k
    num = np.linalg.norm(g0 - ref)
    assert num < 2e-3 * np.linalg.norm(ref), num / np.linalg.norm(ref)
