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


def _rank_job_steps(rank, world, port, ret, use_graph):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    cfg, model, images, t_bbox, t_class = _make()
    import torch.distributed as dist
    from detr_tf import parallel, training
    from detr_tf.optimizers import setup_optimizers
    parallel.init_distributed(backend="gloo")
    model.dp = parallel.DataParallel(model.engine.P.grad, model.engine.P.bucket_bounds(), engine=model.engine)
    assert model.engine.dp_rank == rank
    opt = setup_optimizers(model, cfg)
    lo, hi = parallel.shard_batch(2, rank, world)
    stepper = training.GraphedTrainStep(model, opt, cfg) if use_graph else None
    totals, grads = [], []
    for i in range(3):
        if use_graph:
            _, total, _ = stepper(images[lo:hi], t_bbox[lo:hi], t_class[lo:hi], i)
        else:
            _, total, _ = training.train_step(model, images[lo:hi], t_bbox[lo:hi], t_class[lo:hi], opt, cfg, i)
        totals.append(float(total))
        torch.cuda.synchronize()
        grads.append(model.engine.P.grad.cpu().numpy().copy())
    ret[rank] = (model.engine.P.flat.cpu().numpy(), totals, len(stepper.step_graph.graphs) if use_graph else 0, grads)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_rank_dp_steps_keep_replicas_identical(hip, use_graph):
    """3 optimiser steps on 2 ranks x 1 image, eager and as the recorded step (the graph is cut at the exchange points --
    loss normalisers, 4 gradient buckets -- and the collectives run eagerly between the segments: 6 segments): after
    every step both ranks hold bit-identical gradients and parameters, and the first step equals the single-process step
    on the batch of 2 (later steps are compared through the loss only: Hungarian matching of a near-degenerate random-init
    cost matrix amplifies rounding noise into different -- equally optimal to rounding -- assignments)."""
    from detr_tf import training
    from detr_tf.optimizers import setup_optimizers
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_job_steps, args=(2, _free_port(), ret, use_graph), nprocs=2, join=True)
    f0, t0, nseg0, g0 = ret[0]
    f1, t1, nseg1, g1 = ret[1]
    if use_graph:
        assert nseg0 == nseg1 == 6, (nseg0, nseg1)             # 1 normaliser cut + 4 bucket cuts -> 6 segments
    for s in range(3):
        nd = int((g0[s] != g1[s]).sum())
        assert nd == 0, f"step {s}: {nd} gradient entries differ between the ranks after the all-reduce (max {np.abs(g0[s] - g1[s]).max():.3e})"
    assert np.array_equal(f0, f1), f"replicas diverged: {int((f0 != f1).sum())} parameters differ, max {np.abs(f0 - f1).max():.3e}"
    assert np.allclose(t0, t1, rtol=1e-6)
    cfg, model, images, t_bbox, t_class = _make()
    opt = setup_optimizers(model, cfg)
    _, total, _ = training.train_step(model, images, t_bbox, t_class, opt, cfg, 0)
    torch.cuda.synchronize()
    assert abs(t0[0] - float(total)) < 1e-4 * abs(float(total))
    ref = model.engine.P.grad.cpu().numpy()
    assert np.linalg.norm(g0[0] - ref) < 2e-3 * np.linalg.norm(ref)
    assert t0[2] < t0[0]                                       # and it trains


def _run_bench(args, env_extra=None, timeout=600):
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_forced_single_rank_rccl_path(hip):
    """bench.py through torch.distributed.run with ONE rank and DETR_DP_FORCE=1: the RCCL (backend "nccl") process group, the
    parameter broadcast, the normaliser all-reduce and the bucketed gradient all-reduce on the side stream all execute for
    real (world size 1), inside the segmented hipGraph replay."""
    import json
    import sys
    def run():
        return _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
                           str(_free_port()), "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "2", "--height", "128",
                           "--width", "160", "--no-cpu-baseline", "--no-fp32-leg", "--no-configs", "--no-kernel-events", "--launch", "graph"],
                          env_extra={"DETR_DP_FORCE": "1"})
    r = run()
    if r.returncode != 0 and any(l.startswith("{") for l in r.stdout.splitlines()):
        # a rank that has printed its result line and then aborts during process teardown (exit -6 from a library helper
        # thread was seen once on the 1-GPU box, stderr truncated) has completed the measurement: run again, and report both
        # runs if it repeats; an abort BEFORE the result line fails right away with the head and the tail of stderr
        first = r
        r = run()
        assert r.returncode == 0, "twice:\n" + first.stderr[:1500] + "\n...\n" + r.stderr[:1500] + "\n...\n" + r.stderr[-1500:]
    assert r.returncode == 0, r.stderr[:1500] + "\n...\n" + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["steps"] == 3 and np.isfinite(j["loss"]) and j["value"] > 0
    assert j["config"]["launch"].startswith("hipGraph replay") and j["config"]["parallelism"] == "dp1"
    # the gradient exchange is instrumented: HIP events at every bucket hand-over / completion
    assert j["ranks_seen"] == 1 and j["comm_ms"] > 0 and 0 <= j["exposed_comm_ms"] < 1e3
    assert [b["bucket"] for b in j["comm"]["buckets"]] == [0, 1, 2, 3] and j["comm"]["steps"] == 3


def test_bench_two_ranks_gloo_prints_the_comm_fields(hip):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per process), here with both ranks on the
    one GPU over gloo: the JSON line carries comm_ms / exposed_comm_ms / ranks_seen and the per-bucket all-reduce timings, so the
    first real multi-GPU line can be read (how much of the exchange the backward hides)."""
    import json
    r = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                    str(_free_port()), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--height", "128",
                    "--width", "160", "--dist-backend", "gloo", "--no-cpu-baseline", "--no-kernel-events", "--launch", "eager"])
    assert r.returncode == 0, r.stderr[:1500] + "\n...\n" + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly rank 0 prints"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 4 and j["config"]["parallelism"] == "dp2" and j["scaling"] == "weak"
    assert j["ranks_seen"] == 2 and j["comm_ms"] > 0 and j["exposed_comm_ms"] >= 0
    c = j["comm"]
    assert c["steps"] == 3 and len(c["buckets"]) == 4 and abs(sum(b["mbytes"] for b in c["buckets"]) - 166.0) < 8.0
    assert all(b["allreduce_ms"] > 0 and b["handover_to_done_ms"] >= b["allreduce_ms"] * 0.5 for b in c["buckets"])


def test_bench_with_a_stalled_rank_exits_nonzero_with_a_stack_dump(hip):
    """a rank that never reaches the timed steps (DETR_BENCH_STALL_RANK: it sleeps behind the barrier that opens the
    timed region) must cost the watchdog timeout, not the lease: the healthy rank's watchdog (parallel.Watchdog, --dp-timeout 20)
    dumps every thread's stack and exits 1, torchrun tears the job down and returns non-zero -- well inside 120 s."""
    import time
    t0 = time.time()
    r = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                    str(_free_port()), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "1", "--height", "64",
                    "--width", "96", "--dist-backend", "gloo", "--no-cpu-baseline", "--no-kernel-events", "--launch", "eager",
                    "--dp-timeout", "20"], env_extra={"DETR_BENCH_STALL_RANK": "1"}, timeout=300)
    took = time.time() - t0
    assert r.returncode != 0, "a stalled rank must not look like a finished benchmark"
    assert not any(l.startswith("{") for l in r.stdout.splitlines()), "no result line may be printed"
    assert "most recent call first" in r.stderr, r.stderr[-3000:]          # faulthandler's stack dump
    assert took < 120.0, f"took {took:.0f} s"


def test_bench_refuses_rank_count_mismatch(hip):
    """`python bench.py --gpus 2` spawns two ranks itself; on a 1-GPU box that must fail loudly, never print a 1-GPU number
    labelled n_gpus = 2."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a 1-GPU box")
    r = _run_bench(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "1", "--height", "64", "--width", "96",
                    "--no-cpu-baseline", "--no-fp32-leg", "--no-configs"], timeout=300)
    assert r.returncode != 0
    assert not any(l.startswith("{") for l in r.stdout.splitlines())
    assert "only 1 GPU(s) visible" in (r.stderr + r.stdout)
    # and a launcher world size that contradicts --gpus is refused as well
    r = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
                    str(_free_port()), "bench.py", "--gpus", "4", "--steps", "1", "--warmup", "1"], timeout=300)
    assert r.returncode != 0 and "--gpus 4 but the process group has 1 rank" in (r.stderr + r.stdout)
