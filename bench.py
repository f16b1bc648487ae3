#!/usr/bin/env python
"""bench.py -- images/sec of one DETR-R50 training step (forward with training=True, Hungarian
set loss with 5 aux levels, backward, per-tensor clipnorm, 3x Adam) at 800x1333, batch 8 per GPU,
synthetic data, random-init weights (BASELINE.json metric / SURVEY.md 8d).  Default --precision bf16
(BASELINE.json config C3: bf16 MFMA, bf16 weight shadow and backbone activation storage, fp32 master
weights / accumulation / loss); --precision fp32 is the exact-f32 parity mode, whose rate the bf16 line
also reports (`images_per_sec_fp32_parity_mode`).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0.  `roofline` = the dominant kernel (template instantiation of the GEMM
tile engine) timed live with HIP events on the launch stream; `cpu_baseline` = the CPU oracle
(restatement of the TF reference; TF is not installable here) timed on a bounded sample on this box's
host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

FWD_GFLOP_PER_IMAGE = 203.3          # SURVEY.md 8d (R50, 800x1333, Q=100)
STEP_GFLOP_PER_IMAGE = 610.0         # fwd + dgrad + wgrad
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def make_targets(B, rng, rows=100):
    """SURVEY.md 8d synthetic targets in the header layout of detr_tf/data/processing.py:35-55."""
    t_bbox = np.zeros((B, rows, 4), np.float32)
    t_class = np.zeros((B, rows, 1), np.int64)
    for b in range(B):
        n = int(np.clip(rng.poisson(7), 1, rows - 1))
        if b == B - 1:
            n = rows - 1
        t_bbox[b, 0, 0] = n
        t_bbox[b, 1:1 + n, 0:2] = rng.uniform(0.2, 0.8, (n, 2))
        t_bbox[b, 1:1 + n, 2:4] = rng.uniform(0.05, 0.4, (n, 2))
        t_class[b, 1:1 + n, 0] = rng.integers(1, 91, n)
    return t_bbox, t_class


def _pick_threads(requested):
    if requested > 0:
        return requested
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 32))        # torch-CPU convs stop scaling (and thrash) far below 256 threads


def cpu_baseline(height, width, budget_s=25.0, threads=0):
    """The oracle (kind "port": CPU restatement of the reference, torch-CPU fp32 + SciPy matcher)
    timed on the host cores on a bounded sample: train steps at batch 1 of the same shape."""
    from oracle import detr_ref as R, optim_ref as O, set_loss_ref as L
    cores = _pick_threads(threads)
    torch.set_num_threads(cores)
    params = R.make_params(0)
    rng = np.random.default_rng(1234)
    img = torch.from_numpy(rng.normal(size=(1, height, width, 3)).astype(np.float32))
    tb, tc = L.make_targets(1, seed=1235, force_full=False)
    opts = {g: O.Adam(lr, clipnorm=0.1) for g, lr in (("backbone", 1e-5), ("transformers", 1e-4), ("nlayers", 1e-4))}
    n, t0 = 0, time.perf_counter()
    times = []
    while True:
        t1 = time.perf_counter()
        P = R.to_torch(params, requires_grad=True)
        out = R.detr_forward(img, P)
        total, _ = L.get_losses(out, torch.from_numpy(tb), torch.from_numpy(tc), 91)
        total.backward()
        grads = {k: P[k].grad.numpy() for k in P if R.trainable(k)}
        for g in opts:
            opts[g].apply({k: v for k, v in grads.items() if O.variable_group(k) == g}, params)
        times.append(time.perf_counter() - t1)
        n += 1
        if n >= 4 or time.perf_counter() - t0 > budget_s:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": round(1.0 / best, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} train steps (fwd+set loss+bwd+clip+Adam) at batch 1, {height}x{width}, torch-CPU fp32 "
                      f"restatement of the TF reference (TF not installable) + SciPy matcher; best step {best:.2f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--mode", choices=["train", "fwdloss"], default="train")
    ap.add_argument("--dropout", type=float, default=0.1, help="transformer dropout of the training step (reference: 0.1)")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="bf16",
                    help="fp32 = exact-f32 MFMA (parity mode); bf16 = bf16 MFMA with fp32 storage/accumulation (config C3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the 3 extra fp32-parity-mode steps of a bf16 run")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run every step eagerly (default: the step is recorded once as a "
                                                           "hipGraph and replayed, exactly as training.fit does)")
    ap.add_argument("--dist-backend", type=str, default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--event-steps", type=int, default=1, help="timed steps (the last ones) whose launches carry HIP events")
    ap.add_argument("--dump-shapes", type=str, default=None, help="write the per-shape GEMM timing table (JSON) here")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0 = auto)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) ourselves and relay rank 0's line
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        import subprocess
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    from detr_tf import _hip, parallel, training
    from detr_tf.loss.loss import get_losses
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.optimizers import setup_optimizers
    from detr_tf.training_config import TrainingConfig

    rank, world = parallel.init_distributed(backend=args.dist_backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world != max(1, args.gpus) and not (world == 1 and os.environ.get("DETR_DP_FORCE") == "1"):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s); launch with "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` "
                         "(or plain `python bench.py --gpus N`, which spawns the ranks itself)")
    if world > torch.cuda.device_count() and (args.dist_backend or "nccl") == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")

    cfg = TrainingConfig()
    cfg.background_class = 91
    cfg.batch_size = args.batch
    cfg.target_batch = None
    cfg.train_backbone = cfg.train_transformers = cfg.train_nlayers = True
    model = get_detr_model(cfg, include_top=True, device=str(dev), seed=0, dropout=args.dropout, precision=args.precision)
    opt = setup_optimizers(model, cfg)
    if world > 1 or dist.is_initialized():
        # identical replicas: broadcast rank 0's parameters, then all-reduce gradients every step
        dist.broadcast(model.engine.P.flat, src=0)
        for raw in model.engine.P.bn_raw.values():
            dist.broadcast(raw, src=0)
        model.engine.fold_bn()
        model.dp = parallel.DataParallel(model.engine.P.grad, model.engine.P.bucket_bounds(), engine=model.engine)

    rng = np.random.default_rng(1234 + rank)
    images = torch.from_numpy(rng.normal(size=(args.batch, args.height, args.width, 3)).astype(np.float32)).to(dev)
    tb, tc = make_targets(args.batch, np.random.default_rng(1235 + rank))
    tb, tc = torch.from_numpy(tb).to(dev), torch.from_numpy(tc).to(dev)

    use_graph = not args.no_graph
    stepper = {}

    def step(i):
        if args.mode == "train":
            if use_graph and _hip.PROFILER is None:      # (HIP events around single launches need the eager step)
                if id(model) not in stepper:
                    stepper.clear()
                    stepper[id(model)] = training.GraphedTrainStep(model, opt, cfg)
                out, total, log = stepper[id(model)](images, tb, tc, i)
            else:
                out, total, log = training.train_step(model, images, tb, tc, opt, cfg, i)
            return total
        out = model(images, training=False)
        total, log = get_losses(out, tb, tc, cfg)
        return total

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss_first = None
    for i in range(args.warmup):
        last = step(i)
        if i == 0:
            loss_first = last          # device scalar: read after the timed region
    # HIP events around every GEMM / conv launch (roofline leg) cost ~4 us of host time per event (~3.5 ms per step),
    # so they are recorded in the LAST `event_steps` timed steps only; the other timed steps run uninstrumented.
    prof = None
    ev_steps = 0 if (args.no_kernel_events or rank != 0) else max(1, min(args.event_steps, args.steps))
    if ev_steps:
        prof = _hip.KernelProfiler(prealloc=1200 * ev_steps)     # event objects exist before the timed region
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if prof is not None and i == args.steps - ev_steps:
            _hip.PROFILER = prof
        last = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    _hip.PROFILER = None
    loss_val = float(last)
    # the same step with dropout disabled (SURVEY.md 8d asks for both numbers)
    value_nodrop = None
    if args.mode == "train" and args.dropout > 0.0:
        model.engine.dropout_p = 0.0
        step(0)
        barrier()
        t1 = time.perf_counter()
        for i in range(3):
            step(i)
        barrier()
        value_nodrop = args.batch * world * 3 / (time.perf_counter() - t1)
        model.engine.dropout_p = args.dropout
    # fp32 parity mode (exact-f32 MFMA everywhere: the mode the oracle parity tests run in), reported beside the headline
    value_fp32 = None
    bf16_loss_dev = None
    if args.mode == "train" and args.precision == "bf16" and world == 1 and not args.no_fp32_leg:
        model = opt = None
        torch.cuda.empty_cache()
        model = get_detr_model(cfg, include_top=True, device=str(dev), seed=0, dropout=args.dropout, precision="fp32")
        opt = setup_optimizers(model, cfg)
        loss_first_fp32 = float(step(0))          # same weights, batch and dropout masks as the first bf16 step
        barrier()
        t1 = time.perf_counter()
        for i in range(3):
            step(1 + i)
        barrier()
        value_fp32 = args.batch * 3 / (time.perf_counter() - t1)
        if loss_first is not None:
            bf16_loss_dev = abs(float(loss_first) - loss_first_fp32) / abs(loss_first_fp32)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt

    if rank == 0:
        gflop = STEP_GFLOP_PER_IMAGE if args.mode == "train" else FWD_GFLOP_PER_IMAGE
        scale = (args.height * args.width) / (800.0 * 1333.0)
        roofline = None
        if prof is not None:
            fam = prof.summary()
            if args.dump_shapes:
                with open(args.dump_shapes, "w") as f:
                    json.dump({"steps": ev_steps, "rows": prof.by_shape(60)}, f, indent=1)
            if fam:
                # dominant kernel = the template instantiation (as rocprofv3 names it, tile sizes pooled) with the largest
                # total time; its average launch duration is directly comparable with profiles/r01_rocprofv3_kernel_stats*.txt
                dom = max(fam, key=lambda k: fam[k]["ms"])
                d = fam[dom]
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                traffic, traffic_src = None, None
                tpath = os.path.join(ROOT, "profiles", "r01_traffic_bf16.json" if args.precision == "bf16" else "r01_traffic.json")
                if os.path.exists(tpath):
                    with open(tpath) as f:          # measured offline by separate rocprofv3 --pmc passes
                        tj = json.load(f)
                    if dom in tj.get("per_symbol", {}):
                        traffic = round(tj["per_symbol"][dom]["traffic_bytes_per_launch"])
                        traffic_src = f"profiles/{os.path.basename(tpath)} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
                gemm_all = {"ms": sum(v["ms"] for k, v in fam.items() if k.startswith("gemm_")),
                            "launches": sum(v["launches"] for k, v in fam.items() if k.startswith("gemm_")),
                            "flops": sum(v["flops"] for k, v in fam.items() if k.startswith("gemm_")),
                            "bytes": sum(v["bytes"] for k, v in fam.items() if k.startswith("gemm_"))}
                gemm_all_out = {"ms_per_step": round(gemm_all["ms"] / ev_steps, 3), "launches_per_step": gemm_all["launches"] // ev_steps,
                                "tflops": round(gemm_all["flops"] / (gemm_all["ms"] * 1e-3) / 1e12, 2),
                                "gbs": round(gemm_all["bytes"] / (gemm_all["ms"] * 1e-3) / 1e9, 1)}
                if args.precision == "bf16":
                    # fp32 storage + bf16 MFMA: the GEMM-class kernels are HBM bound -> algorithmic bytes / time
                    gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                    roofline = {"bound": "hbm", "kernel": "detr::" + dom + (" (all K / layout / epilogue instantiations pooled)" if dom.startswith("gemm_stream")
                                                                  else " (64x64 / 128x128 tiles pooled)"), "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                                "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                                "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                                "tflops": round(ach, 2), "launches_per_step": d["launches"] // ev_steps,
                                "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                                "events": f"HIP events on the launch stream around every launch of the last {ev_steps} timed step(s)",
                                "all_gemm_kernels": gemm_all_out,
                                "families": {k: {"ms_per_step": round(v["ms"] / ev_steps, 3),
                                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                                 "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                                 "launches_per_step": v["launches"] // ev_steps} for k, v in fam.items()}}
                else:
                  roofline = {"bound": "mfma", "kernel": "detr::" + dom + " (tile sizes pooled)", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                            "traffic_source": traffic_src,
                            "algorithmic_flops_per_launch": round(d["flops"] / d["launches"]),
                            "launches_per_step": d["launches"] // ev_steps,
                            "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                            "events": f"HIP events on the launch stream around every launch of the last {ev_steps} timed step(s)",
                            "all_gemm_kernels": gemm_all_out,
                            "families": {k: {"ms_per_step": round(v["ms"] / ev_steps, 3),
                                             "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                             "launches_per_step": v["launches"] // ev_steps} for k, v in fam.items()}}
        res = {
            "metric": "images/sec training step, DETR-R50 800x1333 bs=8/GPU" if args.mode == "train"
                      else "images/sec forward+set-loss, DETR-R50 800x1333 bs=8/GPU",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"DETR-R50 {'train step (fwd+set loss 6 levels+bwd+clipnorm+3xAdam)' if args.mode == 'train' else 'forward+set loss'}, "
                                   f"{args.height}x{args.width}, batch {args.batch}/GPU, 100 queries, 92 logits, 6+6 layers, dropout {args.dropout}",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "weights": "random init (seeded)"},
            "loss": round(loss_val, 5),
            "images_per_sec_dropout_off": round(value_nodrop, 3) if value_nodrop else None,
            "images_per_sec_fp32_parity_mode": round(value_fp32, 3) if value_fp32 else None,
            "bf16_vs_fp32_loss_rel_dev_first_step": (float(f"{bf16_loss_dev:.3e}") if bf16_loss_dev is not None else None),
            "whole_step_fraction_of_f32_mfma_peak": round(value / world * gflop * scale / 1e3 / PEAK_F32_MFMA_TFLOPS, 4),
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(args.height, args.width, threads=args.cpu_threads)
            except Exception as e:          # the baseline is a report, never the product path
                res["cpu_baseline"] = {"error": repr(e)}
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
