"""CPU tests of the multi-process data-parallel path (gloo, world_size 2 / 4 / 8) and of the host logic
that does not need a GPU (parameter store layout, buckets, config, positional encoding)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "detr-tensorflow_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from detr_tf import parallel
    r, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return dict(ret)


def _bucket_job(rank, world):
    from detr_tf import parallel
    n = 1000
    grad = torch.arange(n, dtype=torch.float32) * (rank + 1)
    bounds = [(0, 300), (300, 300), (300, 900), (900, 1000)]      # one empty bucket
    dp = parallel.DataParallel(grad, bounds)
    order = []
    for i in range(4):
        dp.on_bucket(i)
        order.append(i)
    dp.finish()
    sums = torch.tensor([1.0 + rank, 2.0, 3.0 * rank])
    dp.reduce_sums(sums)
    return grad.numpy().copy(), sums.numpy().copy(), parallel.shard_batch(64, rank, world)


def test_bucketed_allreduce_and_shards_gloo():
    out = _run(_bucket_job, 2)
    expect = np.arange(1000, dtype=np.float32) * 3            # (1 + 2) x
    for r in (0, 1):
        g, s, shard = out[r]
        assert np.array_equal(g, expect)
        assert np.allclose(s, [3.0, 4.0, 3.0])
        assert shard == (r * 32, (r + 1) * 32)


@pytest.mark.parametrize("world", [4, 8])
def test_bucketed_allreduce_and_shards_at_4_and_8_ranks(world):
    """SURVEY 4 planned 2-8 ranks: the same four-bucket exchange (one bucket empty), the normaliser all-reduce and the batch
    shards at world sizes 4 and 8; every replica ends with the same bits."""
    out = _run(_bucket_job, world)
    k = world * (world + 1) // 2                                   # sum of (rank + 1)
    expect = np.arange(1000, dtype=np.float32) * k
    for r in range(world):
        g, s, shard = out[r]
        assert np.array_equal(g, expect)
        assert np.allclose(s, [k, 2.0 * world, 3.0 * world * (world - 1) / 2])
        assert shard == (r * (64 // world), (r + 1) * (64 // world))
        assert np.array_equal(g, out[0][0]) and np.array_equal(s, out[0][1])


def _replica_job(rank, world):
    """Three steps of bucketed exchange + the oracle's Adam on every rank: replicas must stay bit-identical (fp32 and bf16 buckets)."""
    from detr_tf import parallel
    from oracle import optim_ref as O
    n = 2048
    res = {}
    for dt in ("fp32", "bf16"):
        w = {"w": np.linspace(-1.0, 1.0, n).astype(np.float32)}
        opt = O.Adam(1e-3, clipnorm=0.1)
        for step in range(3):
            gen = torch.Generator().manual_seed(1000 * step + rank)
            grad = torch.randn(n, generator=gen)
            dp = parallel.DataParallel(grad, [(0, 700), (700, 1500), (1500, 1500), (1500, n)], bucket_dtype=dt)
            for i in range(4):
                dp.on_bucket(i)
            dp.finish()
            opt.apply({"w": grad.numpy()}, w)
        res[dt] = w["w"].copy()
    return res["fp32"], res["bf16"]


@pytest.mark.parametrize("world", [4, 8])
def test_replicas_stay_bit_identical_over_steps_at_4_and_8_ranks(world):
    out = _run(_replica_job, world)
    for r in range(1, world):
        assert np.array_equal(out[r][0], out[0][0]), f"fp32 buckets: rank {r} drifted"
        assert np.array_equal(out[r][1], out[0][1]), f"bf16 buckets: rank {r} drifted"
    assert not np.array_equal(out[0][0], out[0][1])                # (the bf16 exchange is a different, coarser sum)


def _stall_worker(rank, world, port, timeout_s):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "detr-tensorflow_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from detr_tf import parallel
    wd = parallel.Watchdog(timeout_s)
    wd.feed("rendezvous")
    parallel.init_distributed(backend="gloo", timeout_s=timeout_s)
    parallel.checked_barrier(wd, "first barrier")
    if rank == 1:
        time.sleep(60)                    # the stalled rank
    parallel.checked_barrier(wd, "second barrier")
    wd.stop()


def test_a_stalled_rank_is_detected_within_the_timeout():
    """parallel.Watchdog + the process-group timeout: rank 1 of 2 sleeps in front of a barrier; rank 0 must give up within the
    timeout (monitored_barrier names the missing rank, or the watchdog dumps the stacks and exits 1) instead of waiting forever."""
    import subprocess
    import sys
    import time
    code = ("import sys; sys.path.insert(0, %r); import tests.test_parallel_cpu as t; "
            "t._stall_worker(int(sys.argv[1]), 2, int(sys.argv[2]), 5.0)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port)], stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    try:
        _, err0 = procs[0].communicate(timeout=45)
    finally:
        for p in procs:
            p.kill()
    took = time.time() - t0
    assert procs[0].returncode not in (0, None), "the healthy rank must fail, not hang or succeed"
    assert took < 45.0
    assert ("most recent call first" in err0) or ("Rank 1" in err0) or ("rank 1" in err0), err0[-2000:]


def test_free_port_and_single_rank_group_without_master_port():
    """parallel.free_port() returns a bindable port; a forced single-rank group picks one by itself (no fixed 29500)."""
    import socket
    import subprocess
    import sys
    from_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['DETR_DP_FORCE'] = '1'; os.environ.pop('MASTER_PORT', None); "
            "from detr_tf import parallel; p = parallel.free_port(); import socket; s = socket.socket(); s.bind(('127.0.0.1', p)); s.close(); "
            "r, w = parallel.init_distributed(backend='gloo'); assert (r, w) == (0, 1); assert os.environ['MASTER_PORT'] != '29500'; "
            "import torch.distributed as d; d.destroy_process_group(); print('ok')") % os.path.join(from_root, "detr-tensorflow_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]


def _bf16_bucket_job(rank, world):
    """The same four-bucket exchange with fp32 buckets and with bf16 buckets (DataParallel(bucket_dtype="bf16")), followed by the
    SAME fp32 Adam step on both results (oracle/optim_ref.py): what a bf16 exchange costs the master update."""
    from detr_tf import parallel
    from oracle import optim_ref as O
    n = 4096
    gen = torch.Generator().manual_seed(100 + rank)
    grad0 = torch.randn(n, generator=gen) * torch.logspace(-4, 1, n)          # six decades of magnitudes
    bounds = [(0, 1000), (1000, 1000), (1000, 3000), (3000, n)]
    out = {}
    for dt in ("fp32", "bf16"):
        grad = grad0.clone()
        dp = parallel.DataParallel(grad, bounds, bucket_dtype=dt)
        assert dp.bucket_dtype == dt and (dp.staging is None) == (dt == "fp32")
        for i in range(4):
            dp.on_bucket(i)
        dp.finish()
        out[dt] = grad.numpy().copy()
    w = np.linspace(-1.0, 1.0, n).astype(np.float32)
    upd = {}
    for dt in out:
        params = {"w": w.copy()}
        O.Adam(1e-4, clipnorm=0.1).apply({"w": out[dt]}, params)
        upd[dt] = params["w"] - w
    return out["fp32"], out["bf16"], upd["fp32"], upd["bf16"], grad0.numpy()


def test_bf16_gradient_buckets_against_fp32_buckets_gloo():
    """gradient exchange in bf16 buckets as an OPTION (half the bytes over xGMI), with the fp32 master update intact.
    2-rank gloo: every element of the bf16-exchanged sum is within 2^-7 * (|g_0| + |g_1|) of the fp32-exchanged sum (one RNE
    rounding per addend, half an ulp = 2^-8 relative each, + one of the sum), and the first Adam update (clipnorm 0.1) computed from it differs by less than 5 % of the update's norm (measured 1.8 %: Adam's
    g / (|g| + eps) cancels the rounding of the large entries; the entries that the clip scales below eps carry it through)."""
    out = _run(_bf16_bucket_job, 2)
    g0, g1 = out[0][4], out[1][4]
    for r in (0, 1):
        s32, s16, u32, u16, _ = out[r]
        assert np.array_equal(s32, g0 + g1)
        err = np.abs(s16 - s32)
        assert (err <= 2.0 ** -7 * (np.abs(g0) + np.abs(g1)) + 1e-30).all(), float((err / (np.abs(g0) + np.abs(g1) + 1e-30)).max())
        assert float(err.max()) > 0                                  # it IS a different exchange
        assert np.linalg.norm(u16 - u32) <= 5e-2 * np.linalg.norm(u32)
    assert np.array_equal(out[0][1], out[1][1])                      # replicas stay identical


def _loss_normaliser_job(rank, world):
    """The whole-batch set loss (loss.py:66-67,82,94) from per-rank sums + ONE all-reduce of the
    normalisers equals the oracle on the concatenated batch."""
    from detr_tf import parallel
    from oracle import set_loss_ref as L
    B, Q, C = 4, 100, 92
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(B, Q, C, generator=g)
    boxes = torch.rand(B, Q, 4, generator=g) * 0.5 + 0.1
    tb, tc = L.make_targets(B, seed=4, force_full=False)
    tb, tc = torch.tensor(tb), torch.tensor(tc)
    lo, hi = parallel.shard_batch(B, rank, world)
    sums = torch.zeros(4)                                        # sum w*CE, sum w, sum L1, sum (1-giou)  (+ n_pos = col 1 pos part)
    npos = torch.zeros(1)
    for b in range(lo, hi):
        ti, pi, sel, tbs, tcs = L.hungarian_matching(tb[b], tc[b], boxes[b], logits[b])
        ce = torch.logsumexp(logits[b], -1)
        tgt = torch.full((Q,), 91, dtype=torch.int64)
        tgt[pi] = tcs[ti].long()
        w = torch.full((Q,), 0.1)
        w[pi] = 1.0
        ce = ce - logits[b].gather(1, tgt[:, None])[:, 0]
        sums[0] += (ce * w).sum()
        sums[1] += w.sum()
        sums[2] += (boxes[b][pi] - tbs[ti]).abs().sum()
        sums[3] += (1 - torch.diagonal(L.giou_matrix(L.xcycwh_to_xy_min_xy_max(boxes[b][pi]), L.xcycwh_to_xy_min_xy_max(tbs[ti])))).sum()
        npos += len(pi)
    dp = parallel.DataParallel(torch.zeros(4), [(0, 4)])
    dp.reduce_sums(sums)
    dp.reduce_sums(npos)
    total = sums[0] / sums[1] + 2 * sums[3] / npos[0] + 5 * sums[2] / npos[0]
    ref_total, _ = L.get_losses({"pred_logits": logits, "pred_boxes": boxes}, tb, tc, 91)
    return float(total), float(ref_total)


def test_dp_loss_normalisers_match_whole_batch_oracle():
    out = _run(_loss_normaliser_job, 2)
    for r in (0, 1):
        got, ref = out[r]
        assert abs(got - ref) < 1e-4 * abs(ref), (got, ref)


def test_param_store_layout_and_buckets_cpu():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "detr-tensorflow_amd"))
    from detr_tf.params import ParamStore, variable_group
    from oracle import detr_ref as R
    st = ParamStore("cpu", num_enc=1, num_dec=1)
    assert all(o % 4 == 0 for o, _ in st.offsets.values())              # 16-byte aligned tensors
    b = st.bucket_bounds()
    assert b[0][0] == 0 and b[-1][1] == st.total and all(b[i][1] == b[i + 1][0] for i in range(3))
    names = list(st.shapes)
    assert names[0].startswith("class_embed") and names[-1] == "backbone/conv1/kernel"    # reverse-forward order
    params = R.make_params(1, num_enc=1, num_dec=1)
    assert st.load_dict(params) == []
    back = st.state_dict()
    assert all(np.array_equal(back[k], params[k]) for k in params)
    t = st.build_tables(["cls_layer", "pos_layer"], chunk=4096)
    assert t["n_chunks"] == sum(-(-n // 4096) for _, n in st.offsets.values())
    assert variable_group("input_proj/bias", []) == 0 and variable_group("query_embed/kernel", []) == 0
    assert variable_group("bbox_embed_1/kernel", []) == 1 and variable_group("pos_layer/dense_0/bias", ["pos_layer"]) == 2


def test_config_and_parser_surface():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "detr-tensorflow_amd"))
    from detr_tf.training_config import TrainingConfig, training_config_parser
    cfg = TrainingConfig()
    assert cfg.image_size == (376, 672) and cfg.batch_size == 1 and cfg.target_batch == 1
    assert cfg.gradient_norm_clipping == 0.1 and float(cfg.backbone_lr) == 1e-5 and float(cfg.nlayers_lr) == 1e-4
    args = training_config_parser().parse_args(["--train_backbone", "--batch_size", "8", "--backbone_lr", "3e-5",
                                                "--data_dir", "/d", "--img_dir", "im"])
    cfg.update_from_args(args)
    assert cfg.train_backbone and not cfg.train_transformers and cfg.batch_size == 8
    assert float(cfg.backbone_lr) == 3e-5 and cfg.target_batch is None and cfg.data.img_dir == "/d/im"
    cfg.transformers_lr.assign(5e-5)
    assert float(cfg.transformers_lr) == 5e-5


def test_position_embedding_host_matches_oracle():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "detr-tensorflow_amd"))
    from detr_tf.engine import position_embedding_sine_host
    from oracle import detr_ref as R
    for H, W in ((25, 42), (15, 20), (32, 42), (2, 3)):
        got = position_embedding_sine_host(H, W)
        ref = R.position_embedding_sine(1, H, W)[0].reshape(H * W, 256).numpy()
        assert np.abs(got - ref).max() < 1e-6


def test_param_store_layout_and_split_heuristics():
    """Host logic without a GPU: every tensor of the flat parameter buffer starts on a 16-byte boundary in BOTH the fp32
    buffer and its bf16 shadow (offsets % 8 == 0), buckets are contiguous prefixes, and pick_split_k stays in range."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "detr-tensorflow_amd"))
    from detr_tf import _hip
    from detr_tf.params import ParamStore
    P = ParamStore("cpu")
    assert all(o % 8 == 0 for o, _ in P.offsets.values())
    assert P.total % 8 == 0 and P.total >= 41_500_000
    bounds = P.bucket_bounds()
    assert bounds[0][0] == 0 and bounds[-1][1] == P.total and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    P.shadow16()
    assert P.views16["transformer/encoder/layer_0/linear1/kernel"].shape == P.views["transformer/encoder/layer_0/linear1/kernel"].shape
    for bf in (0, 1):
        _hip.COMPUTE_BF16 = bf
        try:
            for (M, N, K) in [(256, 256, 8400), (64, 256, 534400), (147, 64, 2134400), (256, 256, 800), (2048, 256, 8400), (4, 256, 4800)]:
                sk = _hip.pick_split_k(M, N, K)
                assert 1 <= sk <= 1024 and sk <= max(1, K // 16)
        finally:
            _hip.COMPUTE_BF16 = 0
