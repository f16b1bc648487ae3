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
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
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
    os.environ.setdefault("MASTER_PORT", "29500")
    if torch.cuda.is_available():
        # one process per GPU; ranks wrap around when a test box has fewer devices than ranks (gloo only)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def shard_batch(global_batch, rank, world):
    """Contiguous shard of the global batch owned by `rank` (images are independent: pure DP)."""
    per = global_batch // world
    assert per * world == global_batch, "global batch must divide by the world size"
    return rank * per, (rank + 1) * per


class DataParallel:
    def __init__(self, grad_flat, bucket_bounds, group=None, engine=None):
        if engine is not None and dist.is_initialized():
            engine.dp_rank = dist.get_rank(group)      # every rank draws its own dropout masks (whole-batch semantics)
        self.grad = grad_flat
        self.bounds = list(bucket_bounds)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("DETR_DP_FORCE") == "1")
        self.works = []
        self.cuda = grad_flat.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda else None

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
        if self.cuda:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the compute stream wait for every outstanding bucket."""
        for w in self.works:
            w.wait()
        self.works = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
