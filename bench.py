#!/usr/bin/env python
"""bench.py -- images/sec of one DETR-R50 training step (forward with training=True, Hungarian set loss with 5 aux
levels, backward, per-tensor clipnorm, 3x Adam) at 800x1333, batch 8 per GPU, synthetic data, random-init weights
(BASELINE.json metric / SURVEY.md 8d).  Default --precision bf16 = BASELINE.json config C3 (bf16 MFMA, bf16 weight shadow
and backbone activation storage, fp32 master weights / accumulation / LayerNorm / softmax statistics / heads / loss);
--precision fp32 is the exact-f32 parity mode -- the reference's own precision -- whose numbers every bf16 line carries
as first-class fields (`fp32`).

  python bench.py --gpus N --steps K --warmup W
      N > 1: the driver's `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU,
      RCCL); plain `python bench.py --gpus N` spawns the N ranks itself.  A rank count that differs from --gpus is an error.

Prints ONE JSON line on rank 0:
  value / ms_per_step   K timed steps (barrier + synchronize on both sides, max over ranks).  The step is what training.fit
                        runs, launched either as a recorded hipGraph (replayed) or eagerly on two HIP streams: --launch
                        auto (default) times both inside the warm-up and keeps the faster (config.launch says which); the last
                        --event-steps timed steps run eagerly with HIP events around every GEMM / conv / attention launch.
  roofline              the dominant kernel family by total time, from those events (on the launch stream).
  step_roofline         SURVEY 8d's mixed roofline: sum over the instrumented launches of max(FLOPs / MFMA peak, bytes / HBM peak)
                        against the measured step.
  fp32                  the parity mode (3 steps) with its own roofline.
  configs               the other single-GPU BASELINE.json configs measured in the same process: C1 (one 480x640 image,
                        eval forward + get_model_inference), C2 (fp32 forward + set loss, batch 8, 800x1333), and C2's bf16 twin.
  cpu_baseline          the CPU restatement of the TF reference (oracle/, kind "port": TF is not installable) on this box's
                        host cores, bounded samples: a batch-1 train step (same unit as `value`), the C1 forward, the C2 forward+loss.
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
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32, exact f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0       # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0                # spec; ~6300 GB/s achievable by a copy kernel


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


def cpu_baseline(height, width, threads=0, budget_s=14.0):
    """The oracle (kind "port": CPU restatement of the reference, torch-CPU fp32 + SciPy matcher) timed on the host cores on
    bounded samples.  `value` = batch-1 train steps of the metric's shape (images/s); `c1_forward_480x640` and
    `c2_forward_loss` are the two CPU-runnable BASELINE.json configs."""
    from oracle import detr_ref as R, optim_ref as O, set_loss_ref as L
    cores = _pick_threads(threads)
    torch.set_num_threads(cores)
    params = R.make_params(0)
    rng = np.random.default_rng(1234)
    t_start = time.perf_counter()

    def best_of(fn, n):
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
            if time.perf_counter() - t_start > budget_s and len(ts) >= 2:
                break
        return min(ts[1:]) if len(ts) > 1 else ts[0], len(ts)

    img = torch.from_numpy(rng.normal(size=(1, height, width, 3)).astype(np.float32))
    tb, tc = L.make_targets(1, seed=1235, force_full=False)
    tb_t, tc_t = torch.from_numpy(tb), torch.from_numpy(tc)
    Pn = R.to_torch(params)
    # C1: single 480x640 image, eval forward + post-processing (eval.py:41-45)
    img1 = torch.from_numpy(rng.normal(size=(1, 480, 640, 3)).astype(np.float32))

    def c1():
        with torch.no_grad():
            L.get_model_inference(R.detr_forward(img1, Pn), 91)
    t_c1, n_c1 = best_of(c1, 4)

    # C2: forward + set loss at the metric's shape, batch 1 (batch 8 = 8 x this: images are independent)
    def c2():
        with torch.no_grad():
            L.get_losses(R.detr_forward(img, Pn), tb_t, tc_t, 91)
    t_c2, n_c2 = best_of(c2, 3)
    opts = {g: O.Adam(lr, clipnorm=0.1) for g, lr in (("backbone", 1e-5), ("transformers", 1e-4), ("nlayers", 1e-4))}

    def train():
        P = R.to_torch(params, requires_grad=True)
        total, _ = L.get_losses(R.detr_forward(img, P), tb_t, tc_t, 91)
        total.backward()
        grads = {k: P[k].grad.numpy() for k in P if R.trainable(k)}
        for g in opts:
            opts[g].apply({k: v for k, v in grads.items() if O.variable_group(k) == g}, params)
    t_tr, n_tr = best_of(train, 4)
    return {"value": round(1.0 / t_tr, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n_tr} train steps (fwd+set loss+bwd+clip+Adam) at batch 1, {height}x{width}, torch-CPU fp32 restatement of the "
                      f"TF reference (TF not installable) + SciPy matcher; best step {t_tr:.2f}s",
            "c1_forward_480x640": {"value": round(1.0 / t_c1, 3), "unit": "images/sec", "latency_ms": round(t_c1 * 1e3, 1),
                                   "sample": f"best of {n_c1} eval forwards + get_model_inference of one 480x640 image"},
            "c2_forward_loss": {"value": round(1.0 / t_c2, 4), "unit": "images/sec",
                                "sample": f"best of {n_c2} forward + set-loss passes at batch 1, {height}x{width} (no backward)"}}


def family_tables(prof, ev_steps, precision):
    """Per-family totals of the HIP-event records, the dominant family and SURVEY 8d's mixed roofline over the instrumented launches.
    ONE grouping rule: a family is one kernel BODY with all its template instantiations pooled -- the tile GEMM engine
    (`gemm_bf16c_kernel` / `gemm_f32_kernel`: every layout, storage type, tile size, plain and grouped launches), the streaming
    GEMM, the 3x3 convolution per direction, the stem, attention forward / backward.  The dominant family is the one with the
    largest pooled time."""
    fam = prof.summary()
    if not fam:
        return None, None
    rows = {}
    mixed_ms = cov_ms = 0.0
    for k, v in fam.items():
        f32 = precision != "bf16" or k.startswith("gemm_f32")
        peak = PEAK_F32_MFMA_TFLOPS if f32 else PEAK_BF16_MFMA_TFLOPS
        t_mfma = v["flops"] / (peak * 1e12) * 1e3
        t_hbm = v["bytes"] / (PEAK_HBM_GBS * 1e9) * 1e3
        # the family's own mixed roofline: every launch priced at max(its FLOPs / MFMA peak, its bytes / HBM peak)
        t_mix = v.get("mixed_s", 0.0) * 1e3 if v.get("mixed_s") else max(t_mfma, t_hbm)
        mixed_ms += t_mix
        cov_ms += v["ms"]
        rows[k] = {"ms_per_step": round(v["ms"] / ev_steps, 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                   "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "launches_per_step": v["launches"] // ev_steps,
                   "roofline_ms": round(t_mix / ev_steps, 3), "bound": "mfma" if t_mfma >= t_hbm else "hbm",
                   "frac_of_mfma_peak": round(t_mfma / v["ms"], 4), "frac_of_hbm_peak": round(t_hbm / v["ms"], 4),
                   "frac_of_mixed_roofline": round(t_mix / v["ms"], 4)}
    dom = max(fam, key=lambda k: fam[k]["ms"])
    return (fam, dom, rows), {"mixed_ms": mixed_ms / ev_steps, "covered_ms": cov_ms / ev_steps}


def roofline_entry(fam, dom, rows, ev_steps, precision, traffic_file):
    d = fam[dom]
    tflops = d["flops"] / (d["ms"] * 1e-3) / 1e12
    gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
    f32 = precision != "bf16" or dom.startswith("gemm_f32")
    peak_tf = PEAK_F32_MFMA_TFLOPS if f32 else PEAK_BF16_MFMA_TFLOPS
    bound = "mfma" if tflops / peak_tf >= gbs / PEAK_HBM_GBS else "hbm"
    traffic = traffic_src = None
    # (the newest committed PMC summary of this precision: profiles/rNN_traffic[_bf16].json)
    for rnd in ("r06", "r05", "r04", "r03"):
        cand = traffic_file.replace("r03", rnd)
        if os.path.exists(os.path.join(ROOT, "profiles", cand)):
            traffic_file = cand
            break
    tpath = os.path.join(ROOT, "profiles", traffic_file)
    if os.path.exists(tpath):
        with open(tpath) as f:          # measured offline by separate rocprofv3 --pmc passes
            tj = json.load(f)
        if dom in tj.get("per_symbol", {}):
            traffic = round(tj["per_symbol"][dom]["traffic_bytes_per_launch"])
            traffic_src = f"profiles/{traffic_file} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction)"
    r = {"bound": bound, "kernel": "detr::" + dom + (" (the bf16 tile-GEMM family: gemm_bf16c{,_group,_k64,_ln}_kernel and the round-5 8-wave LDS-DMA ring kernels "
                                          "gemm_ring{,_wgrad}_kernel, every template instantiation pooled)" if dom == "gemm_bf16c_kernel" else
                                          " (every template instantiation of the kernel body pooled; plain + grouped launches)"),
         "achieved": round(gbs, 1) if bound == "hbm" else round(tflops, 2), "peak": PEAK_HBM_GBS if bound == "hbm" else peak_tf,
         "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
         "frac": round(gbs / PEAK_HBM_GBS if bound == "hbm" else tflops / peak_tf, 4),
         "frac_mfma": round(tflops / peak_tf, 4), "frac_hbm": round(gbs / PEAK_HBM_GBS, 4),
         "frac_mixed": rows[dom]["frac_of_mixed_roofline"],
         "traffic": traffic, "traffic_source": traffic_src,
         "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]), "algorithmic_flops_per_launch": round(d["flops"] / d["launches"]),
         "tflops": round(tflops, 2), "gbs": round(gbs, 1), "launches_per_step": d["launches"] // ev_steps,
         "avg_launch_ms": round(d["ms"] / d["launches"], 4),
         "events": f"HIP events on the launch stream around every launch of {ev_steps} eager step(s)", "families": rows}
    return r


def metric_name(args):
    """BASELINE.json's metric string for the default workload; another --backbone / --height / --width / --batch / --queries names
    itself (the C4 / C5 lines printed the R50 800x1333 bs=8 string)."""
    what = "training step" if args.mode == "train" else "forward+set-loss"
    net = "R101" if args.backbone == "resnet101" else "R50"
    q = "" if args.queries == 100 else f" {args.queries} queries"
    return f"images/sec {what}, DETR-{net} {args.height}x{args.width} bs={args.batch}/GPU{q}"


def attention_valu_bound(B, L, heads=8, p_drop=0.1):
    """VALU-issue floor of the encoder self-attention FORWARD at d_head = 32 (north_star's 80 % MFMA target): per score the softmax +
    dropout costs a fixed number of VALU instructions whatever the MFMA does.  Round 6, re-published from the NEW kernel's count
    (csrc/attention_dma.hip; profiles/r06_attn2_counters.txt: SQ_INSTS_VALU 8.27 M over 4352 waves = 1900 VALU instructions per wave,
    a wave = 64 queries x a quarter of the keys = 16.4 score tiles of 32 x 32): `valu_per_tile` = 116 VALU instructions per 32 x 32 tile
    and wave, prologue / merge included (rounds 3-5: 295 in the key loop alone), 2 issue cycles each at the guide's throughput
    (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cyc; the kernel measures 4.6 cycles per VALU instruction, dependent chains included),
    1024 SIMDs at 2.4 GHz; the MFMA floor of the same tile is 4 x v_mfma_f32_32x32x16_bf16 = 128 cycles.  With 232 VALU cycles per tile
    the MFMA pipe cannot be busier than 128 / 232 = 55 % under perfect overlap, and not busier than 128 / (116 x 4.6) = 24 % at the
    measured VALU issue rate (measured: 0.146) -- the north_star's 80 % stays out of reach at d_head = 32 with softmax + dropout in the loop."""
    valu_per_tile, cyc_per_valu, mfma_cyc_per_tile = 116, 2, 4 * 32
    tiles = B * heads * ((L + 31) // 32) ** 2            # 32-query x 32-key score tiles
    simds, clk = 1024, 2.4e9
    valu_ms = tiles * valu_per_tile * cyc_per_valu / simds / clk * 1e3
    mfma_ms = tiles * mfma_cyc_per_tile / simds / clk * 1e3
    return {"valu_bound_ms_per_launch": round(valu_ms, 4), "mfma_bound_ms_per_launch": round(mfma_ms, 4),
            "max_mfma_busy_if_perfect_overlap": round(mfma_ms / max(valu_ms, mfma_ms), 3),
            "assumptions": f"{valu_per_tile} VALU instr / (32x32 tile, wave) x {cyc_per_valu} cyc, {simds} SIMDs @ 2.4 GHz, B={B} H={heads} L={L}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--mode", choices=["train", "fwdloss"], default="train")
    ap.add_argument("--backbone", choices=["resnet50", "resnet101"], default="resnet50", help="BASELINE config C4 uses resnet101 at 1000x1333")
    ap.add_argument("--queries", type=int, default=100, help="object queries (BASELINE config C5: 300 with --batch 16)")
    ap.add_argument("--dropout", type=float, default=0.1, help="transformer dropout of the training step (reference: 0.1)")
    ap.add_argument("--precision", choices=["fp32", "bf16", "fp32x3"], default="bf16",
                    help="fp32 = exact-f32 MFMA (parity mode); bf16 = bf16 MFMA with fp32 accumulation (config C3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32-parity-mode steps of a bf16 run")
    ap.add_argument("--no-configs", action="store_true", help="skip the C1 / C2 / C4 / C5 measurements")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--launch", choices=("auto", "graph", "eager"), default="auto",
                    help="how a step reaches the GPU: graph = recorded once as hipGraph(s) and replayed (training.GraphedTrainStep), "
                         "eager = one launch call per kernel on two HIP streams (training.train_step); auto (default) times "
                         "3 steps of each inside the warm-up and keeps the faster one -- which one wins depends on the host")
    ap.add_argument("--no-graph", action="store_true", help="same as --launch eager")
    ap.add_argument("--dist-backend", type=str, default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--event-steps", type=int, default=1, help="timed steps (the last ones) whose launches carry HIP events")
    ap.add_argument("--dump-shapes", type=str, default=None, help="write the per-shape GEMM timing table (JSON) here")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0 = auto)")
    ap.add_argument("--grad-buckets", choices=("fp32", "bf16"), default="fp32",
                    help="N > 1: dtype of the gradient buckets on the wire (bf16 = half the bytes over xGMI, fp32 master update; parallel.DataParallel)")
    ap.add_argument("--dp-timeout", type=float, default=float(os.environ.get("DETR_DP_TIMEOUT_S", "120")),
                    help="N > 1: seconds a phase (rendezvous, a warm-up or timed step, a barrier) may take before the rank dumps its stacks and exits 1")
    ap.add_argument("--phase-events", action="store_true", help="after the timed region: 5 eager steps with HIP events at the phase "
                    "boundaries of the launch sequence (backbone / encoder / decoder / heads / set loss, forward and backward, optimiser)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) ourselves and relay rank 0's line
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        sock = socket.socket()                      # a port that is free right now (a fixed one collides with a concurrent run)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    from detr_tf import _hip, parallel, training
    from detr_tf import engine as engine_mod
    from detr_tf.inference import get_model_inference
    from detr_tf.loss.loss import get_losses
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.optimizers import setup_optimizers
    from detr_tf.training_config import TrainingConfig

    # N > 1: every phase is bounded (parallel.Watchdog: stack dump + exit 1 when a phase exceeds --dp-timeout seconds; the process
    # group carries the same timeout), so a stalled or dead rank costs that long -- not the lease
    wd = parallel.Watchdog(args.dp_timeout if (args.gpus > 1 or os.environ.get("DETR_DP_FORCE") == "1") else 0.0)
    wd.feed("rendezvous")
    rank, world = parallel.init_distributed(backend=args.dist_backend, timeout_s=args.dp_timeout)
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
    if args.no_graph:
        args.launch = "eager"
    # the launch path is the product's own: training.GraphedTrainStep(launch=...) is what training.fit() uses, with "auto" its default
    wgrad_stream_default = engine_mod.WGRAD_STREAM

    def build(precision):
        m = get_detr_model(cfg, include_top=True, device=str(dev), seed=0, dropout=args.dropout, precision=precision,
                           backbone=args.backbone, num_queries=args.queries)
        o = setup_optimizers(m, cfg)
        if world > 1 or dist.is_initialized():
            # identical replicas: broadcast rank 0's parameters, then all-reduce gradients every step
            dist.broadcast(m.engine.P.flat, src=0)
            for raw in m.engine.P.bn_raw.values():
                dist.broadcast(raw, src=0)
            m.engine.fold_bn()
            m.dp = parallel.DataParallel(m.engine.P.grad, m.engine.P.bucket_bounds(), engine=m.engine, bucket_dtype=args.grad_buckets)
        return m, o, training.GraphedTrainStep(m, o, cfg, launch=args.launch)

    rng = np.random.default_rng(1234 + rank)
    images = torch.from_numpy(rng.normal(size=(args.batch, args.height, args.width, 3)).astype(np.float32)).to(dev)
    tb, tc = make_targets(args.batch, np.random.default_rng(1235 + rank))
    tb, tc = torch.from_numpy(tb).to(dev), torch.from_numpy(tc).to(dev)
    model, opt, stepper = build(args.precision)

    def step(i, single=False):
        # HIP events around single launches: that step runs on ONE stream, so that a kernel's duration is its own (on two
        # streams the events of co-running kernels overlap and every duration is stretched by its neighbour)
        engine_mod.WGRAD_STREAM = wgrad_stream_default and _hip.PROFILER is None and not single
        if args.mode == "train":
            if _hip.PROFILER is None and not single:   # (HIP events around single launches need the single-stream eager step)
                return stepper(images, tb, tc, i)[1]
            return training.train_step(model, images, tb, tc, opt, cfg, i)[1]
        out = model(images, training=False)
        return get_losses(out, tb, tc, cfg)[0]

    def barrier(phase="barrier"):
        if world > 1:
            parallel.checked_barrier(wd, phase)
        torch.cuda.synchronize()
        wd.feed(phase + " passed")
        if os.environ.get("DETR_BENCH_STALL_RANK") == str(rank) and phase == "timed region start":
            time.sleep(10 * max(args.dp_timeout, 1.0))      # test hook: this rank never reaches the timed steps (tests/test_parallel_cpu.py)

    def timed(n, first=0):
        barrier()
        t = time.perf_counter()
        for i in range(n):
            last = step(first + i)
        barrier()
        return time.perf_counter() - t, last

    loss_first = None
    if args.mode == "train" and not args.no_kernel_events:
        loss_first = step(0, single=True).clone()   # (also allocates the single-stream scratch tensors of the event-instrumented step)
    # warm-up: at least until the stepper's launch path is final ("auto": eager step, recording pass, 3 timed replays, 3 timed
    # eager steps -- all ordinary training steps; which path wins depends on the host, and the ranks agree through a MAX all-reduce)
    for i in range(max(args.warmup, stepper.settle_calls if args.mode == "train" else 1)):
        wd.feed(f"warm-up step {i}")
        last = step(i)
        if i == 0 and loss_first is None:
            loss_first = last.clone()      # device scalar: read after the timed region
    main_choice, main_probe = stepper.choice, stepper.probe
    if model.dp is not None:
        model.dp.enable_timing()               # HIP events around the bucket exchange: comm_ms / exposed_comm_ms of the timed steps
    # HIP events around every GEMM / conv / attention launch (roofline leg) cost ~4 us of host time per event, so they are
    # recorded in the LAST `event_steps` timed steps only (which run eagerly); the other timed steps run uninstrumented.
    prof = None
    ev_steps = 0 if (args.no_kernel_events or rank != 0 or args.mode != "train") else max(1, min(args.event_steps, args.steps))
    if ev_steps:
        prof = _hip.KernelProfiler(prealloc=1400 * ev_steps, f32=(args.precision != "bf16"))     # event objects exist before the timed region
    barrier("timed region start")
    t0 = time.perf_counter()
    for i in range(args.steps):
        if prof is not None and i == args.steps - ev_steps:
            _hip.PROFILER = prof
        last = step(args.warmup + i)
        wd.feed(f"timed step {i}")
    barrier("timed region end")
    dt = time.perf_counter() - t0
    _hip.PROFILER = None
    loss_val = float(last)
    dp_timing = model.dp.timing_summary() if model.dp is not None else None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    solo = world == 1 and not dist.is_initialized()

    # ---- the same step with dropout disabled (SURVEY.md 8d asks for both numbers)
    value_nodrop = None
    if args.mode == "train" and args.dropout > 0.0 and solo:
        model.engine.dropout_p = 0.0
        for i in range(stepper.settle_calls):      # (a new step signature: the stepper records and probes again)
            step(i)
        d, _ = timed(3)
        value_nodrop = args.batch * 3 / d
        model.engine.dropout_p = args.dropout
    phases = None
    if args.phase_events and args.mode == "train":
        acc = {}
        for i in range(5):
            model.engine.phase_events = []
            engine_mod.WGRAD_STREAM = wgrad_stream_default
            training.train_step(model, images, tb, tc, opt, cfg, 1000 + i)
            torch.cuda.synchronize()
            evs = model.engine.phase_events
            model.engine.phase_events = None
            for (n0, e0), (n1, e1) in zip(evs[:-1], evs[1:]):
                acc.setdefault(n0, []).append(e0.elapsed_time(e1))
        phases = {k: round(sorted(v)[len(v) // 2], 3) for k, v in acc.items()}          # median ms; main-stream spans
        phases["sum"] = round(sum(phases.values()), 3)
    main_tables = family_tables(prof, ev_steps, args.precision) if prof is not None else (None, None)
    attn_report = None
    if prof is not None and args.mode == "train":
        Lf = ((args.height + 31) // 32) * ((args.width + 31) // 32)
        attn_report = attention_valu_bound(args.batch, Lf, p_drop=args.dropout)
        for r in prof.by_shape(400):
            if r["family"] == "attention_fwd" and r["shape"] == f"B{args.batch} H8 T{Lf} S{Lf}":
                per = r["ms"] / max(r["launches"], 1)
                attn_report["measured_fwd_ms_per_launch"] = round(per, 4)
                attn_report["frac_of_valu_bound"] = round(attn_report["valu_bound_ms_per_launch"] / per, 3)
    if prof is not None and args.dump_shapes and rank == 0:
        with open(args.dump_shapes, "w") as f:
            json.dump({"steps": ev_steps, "rows": prof.by_shape(80)}, f, indent=1)

    # ---- fp32 parity mode (exact-f32 MFMA everywhere: the reference's precision, the mode of the oracle parity tests)
    fp32 = None
    bf16_loss_dev = None
    if args.mode == "train" and args.precision == "bf16" and solo and not args.no_fp32_leg:
        model = opt = stepper = None
        torch.cuda.empty_cache()
        model, opt, stepper = build("fp32")
        loss_first_fp32 = float(step(0))          # same weights, batch and dropout masks as the first bf16 step
        for i in range(1, stepper.settle_calls):
            step(i)
        n32 = max(10, min(args.steps, 20))
        d, _ = timed(n32, first=stepper.settle_calls)
        fp32 = {"value": round(args.batch * n32 / d, 3), "unit": "images/sec", "ms_per_step": round(d / n32 * 1e3, 3), "steps": n32,
                "dtype": "f32 (v_mfma_f32_32x32x2_f32, exact)",
                "launch": ("hipGraph replay" if stepper.choice == "graph" else "eager, 2 HIP streams"), "launch_probe": stepper.probe}
        if not args.no_kernel_events:
            p32 = _hip.KernelProfiler(prealloc=1400, f32=True)
            _hip.PROFILER = p32
            step(5)
            torch.cuda.synchronize()
            _hip.PROFILER = None
            (fam32, dom32, rows32), mix32 = family_tables(p32, 1, "fp32")
            fp32["roofline"] = roofline_entry(fam32, dom32, rows32, 1, "fp32", "r03_traffic.json")
            fp32["step_roofline"] = {"mixed_ms": round(mix32["mixed_ms"], 3), "frac": round(mix32["mixed_ms"] / (d / n32 * 1e3), 4)}
        if loss_first is not None:
            bf16_loss_dev = abs(float(loss_first) - loss_first_fp32) / abs(loss_first_fp32)
        # ---- the same fp32 step with its GEMMs / convolutions on the bf16 matrix pipe at fp32 accuracy (precision="fp32x3", round 6:
        #      detr_gemm_desc.compute = 2, csrc/gemm_core.h mma_ktile_split3): fp32 storage, same parity tests and bounds as the exact mode
        model = opt = stepper = None
        torch.cuda.empty_cache()
        model, opt, stepper = build("fp32x3")
        loss_first_x3 = float(step(0))
        for i in range(1, stepper.settle_calls):
            step(i)
        d3, _ = timed(n32, first=stepper.settle_calls)
        fp32["f32x3"] = {"value": round(args.batch * n32 / d3, 3), "unit": "images/sec", "ms_per_step": round(d3 / n32 * 1e3, 3), "steps": n32,
                         "dtype": "f32 storage, products = 6 bf16 partial products of the exact 3-way bf16 split of both operands "
                                  "(v_mfma_f32_32x32x16_bf16), f32 accumulate; attention / LayerNorm / heads / loss as in the exact mode",
                         "launch": ("hipGraph replay" if stepper.choice == "graph" else "eager, 2 HIP streams"),
                         "first_step_loss_rel_dev_from_exact_fp32": abs(loss_first_x3 - loss_first_fp32) / abs(loss_first_fp32)}

    # ---- the other single-GPU configs of BASELINE.json
    configs = None
    if solo and not args.no_configs and args.mode == "train":
        configs = {}
        model = opt = stepper = None
        torch.cuda.empty_cache()
        for prec in ("fp32", "fp32x3", "bf16"):
            m = get_detr_model(cfg, include_top=True, device=str(dev), seed=0, dropout=args.dropout, precision=prec)
            for _ in range(2):
                get_losses(m(images, training=False), tb, tc, cfg)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                tot = get_losses(m(images, training=False), tb, tc, cfg)[0]
            torch.cuda.synchronize()
            d = (time.perf_counter() - t) / 3
            key = f"c2_forward_loss_{prec}"
            configs[key] = {"value": round(args.batch / d, 2), "unit": "images/sec", "ms": round(d * 1e3, 3), "loss": round(float(tot), 5),
                            "workload": f"DETR-R50 {prec} forward (eval) + set loss 6 levels, batch {args.batch}, {args.height}x{args.width}",
                            "frac_of_mfma_peak": round(args.batch / d * FWD_GFLOP_PER_IMAGE / 1e3 /
                                                       (PEAK_F32_MFMA_TFLOPS if prec != "bf16" else PEAK_BF16_MFMA_TFLOPS), 4)}
            if True:                 # C1: one 480x640 image, eval forward + post-processing (eval.py:41-45), both precisions
                img1 = torch.from_numpy(np.random.default_rng(7).normal(size=(1, 480, 640, 3)).astype(np.float32)).to(dev)
                for _ in range(2):
                    get_model_inference(m(img1, training=False), 91)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(5):
                    get_model_inference(m(img1, training=False), 91)
                torch.cuda.synchronize()
                d1 = (time.perf_counter() - t) / 5
                configs[f"c1_forward_480x640_{prec}"] = {"value": round(1.0 / d1, 1), "unit": "images/sec", "latency_ms": round(d1 * 1e3, 3),
                                                         "workload": f"DETR-R50 {prec} eval forward + get_model_inference, ONE 480x640 image (58.3 GFLOP)"}
            m = None
            torch.cuda.empty_cache()
        # C4 (R101, 1000x1333, batch 8) and C5 (R50, 300 queries, batch 16) train steps in the driver-run line (they only
        # existed as builder-run files): bf16, eager two-stream launch, 6 warm-up + 8 timed steps each
        default_workload = (args.backbone == "resnet50" and args.queries == 100 and args.batch == 8 and (args.height, args.width) == (800, 1333))
        if default_workload and args.precision == "bf16":
            for key, kw in (("c4_r101_1000x1333_b8_bf16_train", dict(backbone="resnet101", H=1000, W=1333, B=8, Q=100)),
                            ("c5_r50_300q_b16_bf16_train", dict(backbone="resnet50", H=800, W=1333, B=16, Q=300))):
                try:
                    m = get_detr_model(cfg, include_top=True, device=str(dev), seed=0, dropout=args.dropout, precision="bf16",
                                       backbone=kw["backbone"], num_queries=kw["Q"])
                    o = setup_optimizers(m, cfg)
                    st = training.GraphedTrainStep(m, o, cfg, launch="eager")
                    r2 = np.random.default_rng(99)
                    im = torch.from_numpy(r2.normal(size=(kw["B"], kw["H"], kw["W"], 3)).astype(np.float32)).to(dev)
                    b2, c2 = make_targets(kw["B"], np.random.default_rng(98))
                    b2, c2 = torch.from_numpy(b2).to(dev), torch.from_numpy(c2).to(dev)
                    engine_mod.WGRAD_STREAM = wgrad_stream_default
                    for i in range(6):
                        st(im, b2, c2, i)
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for i in range(8):
                        tot = st(im, b2, c2, 6 + i)[1]
                    torch.cuda.synchronize()
                    d = (time.perf_counter() - t) / 8
                    net = "R101" if kw["backbone"] == "resnet101" else "R50"
                    configs[key] = {"value": round(kw["B"] / d, 2), "unit": "images/sec", "ms_per_step": round(d * 1e3, 3), "steps": 8,
                                    "loss": round(float(tot), 5), "launch": "eager, 2 HIP streams",
                                    "metric": f"images/sec training step, DETR-{net} {kw['H']}x{kw['W']} bs={kw['B']}/GPU" + ("" if kw["Q"] == 100 else f" {kw['Q']} queries"),
                                    "workload": f"DETR-{net} bf16 train step (fwd+set loss 6 levels+bwd+clipnorm+3xAdam), {kw['H']}x{kw['W']}, batch {kw['B']}, {kw['Q']} queries, dropout {args.dropout}"}
                except Exception as e:           # (a report: never fails the headline)
                    configs[key] = {"error": repr(e)}
                m = o = st = im = None
                torch.cuda.empty_cache()

    if rank == 0:
        roofline = step_roofline = None
        if main_tables[0] is not None:
            (fam, dom, rows), mix = main_tables
            roofline = roofline_entry(fam, dom, rows, ev_steps, args.precision,
                                      "r03_traffic_bf16.json" if args.precision == "bf16" else "r03_traffic.json")
            step_roofline = {"what": "SURVEY 8d mixed roofline: sum over the instrumented GEMM / conv / attention launches of "
                                     "max(FLOPs / MFMA peak of the launch's dtype, algorithmic bytes / 8 TB/s)",
                             "mixed_ms": round(mix["mixed_ms"], 3), "instrumented_kernel_ms": round(mix["covered_ms"], 3),
                             "ms_per_step": round(ms_per_step, 3), "frac": round(mix["mixed_ms"] / ms_per_step, 4),
                             "frac_of_bf16_mfma_peak": round(value / world * STEP_GFLOP_PER_IMAGE / 1e3 / PEAK_BF16_MFMA_TFLOPS, 4)
                             if args.precision == "bf16" else None,
                             "frac_of_f32_mfma_peak": round(value / world * STEP_GFLOP_PER_IMAGE / 1e3 / PEAK_F32_MFMA_TFLOPS, 4)
                             if args.precision == "fp32" else None}
        res = {
            "metric": metric_name(args),
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "fp32x3": "f32 (bf16x3 split MFMA, fp32-accurate)", "bf16": "bf16"}[args.precision], "data": "synthetic",
            "config": {"workload": f"DETR-{'R101' if args.backbone == 'resnet101' else 'R50'} "
                                   f"{'train step (fwd+set loss 6 levels+bwd+clipnorm+3xAdam)' if args.mode == 'train' else 'forward+set loss'}, "
                                   f"{args.height}x{args.width}, batch {args.batch}/GPU, {args.queries} queries, 92 logits, 6+6 layers, dropout {args.dropout}",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "weights": "random init (seeded)",
                       "launch": ("hipGraph replay" if (main_choice == "graph" and args.mode == "train") else ("eager, 2 HIP streams" if wgrad_stream_default else "eager, 1 HIP stream"))
                                 + (f" (training.fit default 'auto' probe: graph {main_probe['graph_ms']} ms vs eager {main_probe['eager_ms']} ms per step)"
                                    if main_probe else f" (--launch {args.launch})"),
                       "launch_probe": main_probe},
            "loss": round(loss_val, 5),
            "images_per_sec_dropout_off": round(value_nodrop, 3) if value_nodrop else None,
            "fp32": fp32,
            "images_per_sec_fp32_parity_mode": fp32["value"] if fp32 else None,
            "bf16_vs_fp32_loss_rel_dev_first_step": (float(f"{bf16_loss_dev:.3e}") if bf16_loss_dev is not None else None),
            "roofline": roofline, "step_roofline": step_roofline, "attention": attn_report, "phases_ms": phases, "configs": configs,
        }
        if dp_timing is not None:      # N > 1: the gradient exchange, from HIP events at each bucket hand-over / completion (rank 0)
            res["comm_ms"] = dp_timing["comm_ms"]
            res["exposed_comm_ms"] = dp_timing["exposed_comm_ms"]
            res["ranks_seen"] = dp_timing["ranks_seen"]
            res["comm"] = dp_timing
            res["grad_bucket_dtype"] = args.grad_buckets
        if not args.no_cpu_baseline and solo:
            try:
                res["cpu_baseline"] = cpu_baseline(args.height, args.width, threads=args.cpu_threads)
            except Exception as e:          # the baseline is a report, never the product path
                res["cpu_baseline"] = {"error": repr(e)}
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        parallel.checked_barrier(wd, "shutdown")
        dist.destroy_process_group()
    wd.stop()


if __name__ == "__main__":
    main()
