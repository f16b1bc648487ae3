"""Data parallelism: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).  The reference has no multi-GPU path (README.md:19,135); this is the
only exchange step of the hot path:

  1. all-reduce(sum) of the set-loss normalisers (levels*10 floats) between the "sums" and
     "finalize" kernels, so that a B x world step equals the reference's whole-batch semantics;
  2. all-reduce(sum) of the flat gradient buffer in 4 contiguous buckets, each launched on a side
     stream as soon as the backward has finalised it (heads+transformer first, stem last), so the
     collective overlaps the remaining backward.  xGMI is point-to-point, so few large buckets
     (~70 / 60 / 28 / 6 MB) are preferred over many small ones.
"""
import datetime
import faulthandler
import os
import socket
import sys

import torch
import torch.distributed as dist

# Every blocking step of the multi-process path is bounded (round 6): the process group carries a timeout (DETR_DP_TIMEOUT_S,
# default 120 s: a rendezvous that never completes or a collective that a dead / stalled rank never joins raises instead of
# waiting forever), and Watchdog below covers what a timeout inside the transport cannot -- a rank stuck in a device
# synchronisation behind a collective kernel that spins on a peer.
DEFAULT_TIMEOUT_S = float(os.environ.get("DETR_DP_TIMEOUT_S", "120"))


def free_port():
    """A TCP port that is free right now on 127.0.0.1 (for launchers that pick MASTER_PORT themselves)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class Watchdog:
    """Bounds the time between two `feed()` calls: when a phase takes longer than `timeout_s`, every thread's Python stack is
    written to stderr and the process exits with status 1 (faulthandler.dump_traceback_later: a C-level timer thread that needs
    neither the GIL nor a responsive main thread, so it also fires while the main thread sits in hipStreamSynchronize or inside a
    collective).  Under torchrun the launcher then tears the other ranks down and returns non-zero: a stalled rank costs
    `timeout_s`, not the lease.  timeout_s <= 0 disables it."""

    def __init__(self, timeout_s=None, what="data-parallel step"):
        self.timeout_s = DEFAULT_TIMEOUT_S if timeout_s is None else float(timeout_s)
        self.what = what
        self.armed = False

    def feed(self, phase=""):
        if self.timeout_s <= 0:
            return
        self.phase = phase
        faulthandler.dump_traceback_later(self.timeout_s, repeat=False, file=sys.stderr, exit=True)
        self.armed = True

    def stop(self):
        if self.armed:
            faulthandler.cancel_dump_traceback_later()
            self.armed = False

    def __enter__(self):
        self.feed("start")
        return self

    def __exit__(self, *exc):
        self.stop()
        return False


def init_distributed(backend=None, timeout_s=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).  timeout_s (default DETR_DP_TIMEOUT_S = 120):
    the process group's timeout -- rendezvous and every collective."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("DETR_DP_FORCE") == "1"     # single-rank group: exercises the RCCL path on a 1-GPU box
    if world <= 1 and not force:
        return 0, 1
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        # only a group of ONE may choose its own port (every rank of a larger group must be told the same one by its launcher)
        if int(os.environ["WORLD_SIZE"]) > 1:
            raise RuntimeError("init_distributed: WORLD_SIZE > 1 without MASTER_PORT (launch with torch.distributed.run, or set it)")
        os.environ["MASTER_PORT"] = str(free_port())
    if torch.cuda.is_available():
        # one process per GPU; ranks wrap around when a test box has fewer devices than ranks (gloo only)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    t = DEFAULT_TIMEOUT_S if timeout_s is None else float(timeout_s)
    dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=t))
    return dist.get_rank(), dist.get_world_size()


def checked_barrier(watchdog=None, phase="barrier", group=None):
    """dist.barrier() under the watchdog: gloo groups use monitored_barrier (names the ranks that did not arrive), RCCL groups rely
    on the process-group timeout plus the watchdog."""
    if not dist.is_initialized():
        return
    if watchdog is not None:
        watchdog.feed(phase)
    if dist.get_backend(group) == "gloo":
        dist.monitored_barrier(group=group, timeout=datetime.timedelta(seconds=(watchdog.timeout_s if watchdog and watchdog.timeout_s > 0
                                                                                  else DEFAULT_TIMEOUT_S)))
    else:
        dist.barrier(group=group)


def shard_batch(global_batch, rank, world):
    """Contiguous shard of the global batch owned by `rank` (images are independent: pure DP)."""
    per = global_batch // world
    assert per * world == global_batch, "global batch must divide by the world size"
    return rank * per, (rank + 1) * per


class DataParallel:
    """bucket_dtype: "fp32" (default: the exchange is exact up to the summation order of the collective) or "bf16" (env
    DETR_HIP_DP_BF16=1): every bucket is rounded to bf16 (RNE) into a staging buffer, all-reduced there -- half the bytes over the
    per-link-bound xGMI rings, 83 instead of 166 MB per step -- and converted back into the fp32 gradient buffer; the Adam moments
    and the master weights stay fp32.  What it costs: one bf16 rounding per addend plus the collective's bf16 partial sums, i.e.
    a relative error of up to 2^-7 * sum|g_r| / |sum g_r| per gradient element (tests/test_parallel_cpu.py states the bound on a 2-rank exchange)."""

    def __init__(self, grad_flat, bucket_bounds, group=None, engine=None, bucket_dtype=None):
        if engine is not None and dist.is_initialized():
            engine.dp_rank = dist.get_rank(group)      # every rank draws its own dropout masks (whole-batch semantics)
        self.grad = grad_flat
        self.bounds = list(bucket_bounds)
        self.group = group
        if bucket_dtype is None:
            bucket_dtype = "bf16" if os.environ.get("DETR_HIP_DP_BF16") == "1" else "fp32"
        assert bucket_dtype in ("fp32", "bf16")
        self.bucket_dtype = bucket_dtype
        self.staging = torch.empty(sum(max(0, hi - lo) for lo, hi in self.bounds), dtype=torch.bfloat16,
                                   device=grad_flat.device) if bucket_dtype == "bf16" else None
        self._stage_off = {}
        off = 0
        for i, (lo, hi) in enumerate(self.bounds):
            self._stage_off[i] = off
            off += max(0, hi - lo)
        self._pending16 = []             # (fp32 view, bf16 view) of the buckets in flight
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("DETR_DP_FORCE") == "1")
        self.works = []
        self.cuda = grad_flat.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda else None
        self.timing = None               # enable_timing(): HIP events around every bucket exchange (bench.py --gpus N)

    # ---- instrumentation: how much of the exchange is hidden behind the backward ------------------------------------------
    def enable_timing(self, on=True):
        """Record, per step, one HIP event pair around every bucket all-reduce (on the communication stream) and one around
        the compute stream's wait in finish().  timing_summary() then reports, averaged over the recorded steps:
        comm_ms (sum of the collectives' durations), exposed_comm_ms (what the compute stream actually waited at the end of
        the backward = communication NOT hidden by overlap) and the per-bucket hand-over -> done latencies."""
        self.timing = {"steps": [], "cur": []} if (on and self.cuda and self.active) else None

    def _ev(self, stream):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def timing_summary(self):
        t = self.timing
        if not t or not t["steps"]:
            return None
        torch.cuda.synchronize()
        comm = exposed = 0.0
        per_bucket = {}
        for buckets, w0, w1 in t["steps"]:
            exposed += w0.elapsed_time(w1)
            for i, handed, b0, b1 in buckets:
                d = b0.elapsed_time(b1)
                comm += d
                pb = per_bucket.setdefault(i, [0.0, 0.0, 0])
                pb[0] += d
                pb[1] += handed.elapsed_time(b1)
                pb[2] += 1
        n = len(t["steps"])
        return {"steps": n, "comm_ms": round(comm / n, 3), "exposed_comm_ms": round(exposed / n, 3),
                "hidden_frac": round(1.0 - exposed / comm, 4) if comm > 0 else None,
                "buckets": [{"bucket": i, "mbytes": round((self.bounds[i][1] - self.bounds[i][0]) * 4 / 1e6, 1),
                             "allreduce_ms": round(v[0] / v[2], 3), "handover_to_done_ms": round(v[1] / v[2], 3)}
                            for i, v in sorted(per_bucket.items())],
                "ranks_seen": self.world}

    def reduce_sums(self, sums):
        """All-reduce of the loss normalisers (tiny, on the compute stream)."""
        if self.active:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)

    def on_bucket(self, i):
        """Called by the engine when gradient bucket i is final: launch its all-reduce."""
        if not self.active:
            return
        lo, hi = self.bounds[i]
        if hi <= lo:
            return
        view = self.grad[lo:hi]
        v16 = None
        if self.staging is not None:     # bf16 exchange: the collective runs on the staging slice of this bucket
            so = self._stage_off[i]
            v16 = self.staging[so:so + (hi - lo)]
        if self.cuda:
            handed = self._ev(torch.cuda.current_stream()) if self.timing is not None else None
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                b0 = self._ev(self.comm_stream) if handed is not None else None
                if v16 is not None:
                    v16.copy_(view)                    # RNE rounding, on the communication stream
                    self.works.append(dist.all_reduce(v16, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    self._pending16.append((view, v16))
                else:
                    self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                if handed is not None:
                    # RCCL enqueues the collective on this stream: the event behind it marks its completion (a host-side
                    # transport such as gloo completes in finish(): the end event is re-recorded there)
                    self.timing["cur"].append([i, handed, b0, self._ev(self.comm_stream)])
        elif v16 is not None:
            v16.copy_(view)
            self.works.append(dist.all_reduce(v16, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._pending16.append((view, v16))
        else:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the compute stream wait for every outstanding bucket."""
        timing = self.timing is not None and self.cuda and bool(self.works)
        w0 = self._ev(torch.cuda.current_stream()) if timing else None
        host_side = dist.is_initialized() and dist.get_backend(self.group) != "nccl"
        # With async_op=True the RCCL process group runs a collective on its OWN internal stream; Work.wait() orders the
        # stream that is current at the call behind it.  The waits are therefore issued with the COMMUNICATION stream current:
        # the bf16 copy-back below is enqueued there and must not start before its bucket is reduced, and the
        # compute stream then waits for the communication stream once, which covers collectives and copies alike.
        for k, w in enumerate(self.works):
            if self.cuda:
                with torch.cuda.stream(self.comm_stream):
                    w.wait()
            else:
                w.wait()
            if timing and host_side and k < len(self.timing["cur"]):
                self.timing["cur"][k][3] = self._ev(self.comm_stream)
        self.works = []
        if self._pending16:              # bf16 exchange: the reduced buckets back into the fp32 gradient buffer (behind the collectives)
            if self.cuda:
                with torch.cuda.stream(self.comm_stream):
                    for view, v16 in self._pending16:
                        view.copy_(v16)
            else:
                for view, v16 in self._pending16:
                    view.copy_(v16)
            self._pending16 = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if timing:
            self.timing["steps"].append((self.timing["cur"], w0, self._ev(torch.cuda.current_stream())))
            self.timing["cur"] = []
