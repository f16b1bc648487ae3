"""ctypes binding of libdetr_hip.so (the C ABI declared in include/detr_hip.h).

PyTorch is used only as the owner of device memory and of the HIP stream: every wrapper
passes raw device pointers + the current stream to the library.  There is NO fallback: if the
shared library is missing or a call is rejected, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch  # imported first so that libamdhip64 (same SONAME) is the one torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DETR_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libdetr_hip.so")   # DETR_HIP_LIB: A/B builds (scripts/experiments)

f32p = c_void_p


class ReduceDesc(Structure):
    _fields_ = [("ws", c_void_p), ("splits", c_int32), ("part_stride", c_int64), ("rows", c_int32), ("cols", c_int32),
                ("C", c_void_p), ("ldc", c_int64), ("alpha", c_float), ("scale", c_void_p),
                ("rs_ws", c_void_p), ("rs_out", c_void_p), ("rs_alpha", c_float),
                ("ts_bm", c_int32), ("ts_bn", c_int32), ("ts_tiles_n", c_int32)]       # ABI 5: tile-ordered slabs


class GemmDesc(Structure):
    _fields_ = [("M", c_int32), ("N", c_int32), ("K", c_int32),
                ("A", c_void_p), ("lda", c_int64), ("a_kcontig", c_int32),
                ("B", c_void_p), ("ldb", c_int64), ("b_kcontig", c_int32),
                ("C", c_void_p), ("ldc", c_int64),
                ("batch", c_int32), ("batch_inner", c_int32),
                ("sA0", c_int64), ("sA1", c_int64), ("sB0", c_int64), ("sB1", c_int64),
                ("sC0", c_int64), ("sC1", c_int64),
                ("alpha", c_float),
                ("scale", c_void_p), ("bias", c_void_p),
                ("residual", c_void_p), ("ldr", c_int64),
                ("mask", c_void_p), ("ldmask", c_int64),
                ("act", c_int32), ("split_k", c_int32),
                ("workspace", c_void_p), ("workspace_bytes", c_int64),
                ("dropout_p", c_float), ("dropout_seed", ctypes.c_uint32), ("compute", c_int32),
                ("rowsum_a", c_void_p), ("rowsum_alpha", c_float), ("b_dtype", c_int32),
                ("a_dtype", c_int32), ("c_dtype", c_int32), ("r_dtype", c_int32), ("m_dtype", c_int32),
                ("dropout_step", c_void_p), ("defer_out", POINTER(ReduceDesc)),
                ("maskbits_out", c_void_p), ("ld_maskbits_out", c_int64),          # ABI 5: bit-packed ReLU masks
                ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_y", c_void_p), ("ln_mean", c_void_p), ("ln_rstd", c_void_p),
                ("ln_add", c_void_p), ("ln_add_rows", c_int32), ("ln_y2", c_void_p), ("ln_y16", c_void_p), ("ln_eps", c_float)]


class AttnDesc(Structure):
    _fields_ = [("B", c_int32), ("H", c_int32), ("T", c_int32), ("S", c_int32),
                ("q", c_void_p), ("ldq", c_int64), ("k", c_void_p), ("ldk", c_int64), ("v", c_void_p), ("ldv", c_int64),
                ("o", c_void_p), ("ldo", c_int64), ("lse", c_void_p),
                ("d_o", c_void_p), ("ldd_o", c_int64), ("dq", c_void_p), ("lddq", c_int64), ("dk", c_void_p), ("lddk", c_int64),
                ("dv", c_void_p), ("lddv", c_int64), ("delta", c_void_p),
                ("scale", c_float), ("dropout_p", c_float), ("dropout_site", ctypes.c_uint32), ("dropout_step", c_void_p),
                ("compute", c_int32), ("io_dtype", c_int32), ("dropmask", c_void_p)]       # ABI 8: all-bf16 operands + keep bits


class LayerNormDesc(Structure):
    _fields_ = [("rows", c_int32), ("C", c_int32), ("eps", c_float),
                ("x", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("y", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
                ("add", c_void_p), ("add_rows", c_int32), ("y2", c_void_p), ("y16", c_void_p),
                ("dy", c_void_p), ("dx", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_int64),
                ("dx_add", c_void_p),
                ("dx_drop", c_void_p), ("dropout_p", c_float), ("dropout_site", ctypes.c_uint32), ("dropout_step", c_void_p),
                ("dx_drop16", c_void_p), ("defer_blocks_out", POINTER(c_int32)), ("dy_add", c_void_p)]


class StemDesc(Structure):
    _fields_ = [("N", c_int32), ("H", c_int32), ("W", c_int32), ("Ho", c_int32), ("Wo", c_int32),
                ("img", c_void_p), ("w", c_void_p), ("y", c_void_p),
                ("alpha", c_float), ("scale", c_void_p), ("bias", c_void_p),
                ("act", c_int32), ("split", c_int32),
                ("workspace", c_void_p), ("workspace_bytes", c_int64), ("compute", c_int32),
                ("w_dtype", c_int32), ("y_dtype", c_int32)]


class Conv3x3Desc(Structure):
    _fields_ = [("N", c_int32), ("Hi", c_int32), ("Wi", c_int32), ("Ci", c_int32),
                ("Ho", c_int32), ("Wo", c_int32), ("Co", c_int32), ("stride", c_int32), ("pad", c_int32),
                ("x", c_void_p), ("w", c_void_p), ("y", c_void_p),
                ("alpha", c_float),
                ("scale", c_void_p), ("bias", c_void_p), ("residual", c_void_p), ("mask", c_void_p),
                ("act", c_int32), ("split", c_int32),
                ("workspace", c_void_p), ("workspace_bytes", c_int64), ("compute", c_int32), ("w_dtype", c_int32),
                ("x_dtype", c_int32), ("y_dtype", c_int32), ("r_dtype", c_int32), ("m_dtype", c_int32),
                ("maskbits_out", c_void_p)]                                   # ABI 5: bit-packed ReLU masks


class InputDesc(Structure):
    _fields_ = [("B", c_int32), ("Hs", c_int32), ("Ws", c_int32), ("Hd", c_int32), ("Wd", c_int32),
                ("src", c_void_p), ("src_batch_stride", c_int64), ("dst", c_void_p), ("lut", c_void_p),
                ("perm", c_int32 * 3), ("interpolation", c_int32)]


class PostprocessDesc(Structure):
    _fields_ = [("B", c_int32), ("Q", c_int32), ("C", c_int32),
                ("logits", c_void_p), ("sL_b", c_int64), ("sL_q", c_int64),
                ("boxes", c_void_p), ("sB_b", c_int64), ("sB_q", c_int64),
                ("background_class", c_int32), ("bbox_format", c_int32),
                ("out_boxes", c_void_p), ("out_labels", c_void_p), ("out_scores", c_void_p), ("counts", c_void_p)]


class SetLossDesc(Structure):
    _fields_ = [("levels", c_int32), ("B", c_int32), ("Q", c_int32), ("C", c_int32), ("R", c_int32),
                ("logits", c_void_p), ("sL_l", c_int64), ("sL_b", c_int64), ("sL_q", c_int64),
                ("boxes", c_void_p), ("sB_l", c_int64), ("sB_b", c_int64), ("sB_q", c_int64),
                ("t_bbox", c_void_p), ("t_class", c_void_p),
                ("background_class", c_int32)]


# name -> argtypes (every function returns int)
_SIGNATURES = {
    "detr_hip_abi_version": [],
    "detr_hip_reload_tuning": [],
    "detr_hip_struct_layout": [c_int32, POINTER(c_int32), c_int32],
    "detr_hip_memset_zero": [c_void_p, c_size_t, c_void_p],
    "detr_hip_gemm_f32": [POINTER(GemmDesc), c_void_p],
    "detr_hip_gemm_group_f32": [POINTER(GemmDesc), c_int32, c_void_p],
    "detr_hip_gemm_family": [POINTER(GemmDesc)],
    "detr_hip_gemm_ring_plan": [c_int32, c_int32, c_int32, POINTER(c_int32)],
    "detr_hip_splitk_reduce_many": [POINTER(ReduceDesc), c_int32, c_void_p],
    "detr_hip_conv3x3_f32": [POINTER(Conv3x3Desc), c_int32, c_void_p],
    "detr_hip_maxpool3x3s2_fwd_bf16": [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    "detr_hip_maxpool3x3s2_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                       c_void_p],
    "detr_hip_cvt_bf16": [f32p, c_void_p, c_int64, c_void_p],
    "detr_hip_scale_cols_bf16": [f32p, f32p, c_void_p, c_int64, c_int32, c_void_p],
    "detr_hip_scale_cols_bf16_group": [c_void_p, c_int32, c_void_p],
    "detr_hip_stem_conv7x7_f32": [POINTER(StemDesc), c_int32, c_void_p],
    "detr_hip_maxpool3x3s2_fwd_f32": [f32p, f32p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    "detr_hip_maxpool3x3s2_bwd_f32": [f32p, c_void_p, f32p, f32p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    "detr_hip_subsample2_fwd_f32": [f32p, f32p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    "detr_hip_subsample2_bwd_f32": [f32p, f32p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    "detr_hip_layernorm_fwd": [POINTER(LayerNormDesc), c_void_p],
    "detr_hip_layernorm_bwd": [POINTER(LayerNormDesc), c_void_p],
    "detr_hip_attention_fwd": [POINTER(AttnDesc), c_void_p],
    "detr_hip_attention_bwd": [POINTER(AttnDesc), c_void_p],
    "detr_hip_attention_dropmask": [POINTER(AttnDesc), c_void_p],
    "detr_hip_attention_dropmask_many": [POINTER(AttnDesc), c_int32, c_void_p],
    "detr_hip_dropout_f32": [f32p, f32p, c_int64, c_float, ctypes.c_uint32, c_void_p, c_void_p],
    "detr_hip_multi_copy": [c_void_p, c_int32, c_int32, c_void_p],
    "detr_hip_colsum_scaled": [c_void_p, c_int32, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_void_p],
    "detr_hip_fma_vec_group": [c_void_p, c_int32, c_void_p],
    "detr_hip_set_u32x8": [c_void_p] + [ctypes.c_uint32] * 8 + [c_void_p],
    "detr_hip_colsum_f32": [f32p, f32p, c_int64, c_int32, c_int64, c_float, c_void_p],
    "detr_hip_colsum_det_f32": [f32p, f32p, c_int64, c_int32, c_int64, c_float, f32p, c_int64, c_void_p],
    "detr_hip_conv1x1_bwd_fused_bf16": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32, f32p, c_int64, f32p, c_float,
                                        c_int64, c_int32, c_int32, f32p, c_int64, c_void_p],
    "detr_hip_add_bcast_f32": [f32p, f32p, f32p, c_int64, c_int64, c_void_p],
    "detr_hip_add_f32": [f32p, f32p, f32p, c_int64, c_void_p],
    "detr_hip_sigmoid_bwd_f32": [f32p, f32p, f32p, c_int64, c_void_p],
    "detr_hip_relu_mask_f32": [f32p, f32p, f32p, c_int64, c_void_p],
    "detr_hip_scale_cols_f32": [f32p, f32p, f32p, c_int64, c_int32, c_void_p],
    "detr_hip_scale_cols_t_f32": [f32p, f32p, f32p, c_int32, c_int32, c_void_p],
    "detr_hip_bn_fold_f32": [f32p, f32p, f32p, f32p, f32p, f32p, c_int32, c_float, c_void_p],
    "detr_hip_postprocess": [POINTER(PostprocessDesc), c_void_p],
    "detr_hip_input_stage": [POINTER(InputDesc), c_void_p],
    "detr_hip_pad_labels": [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p],
    "detr_hip_match_cost_f32": [POINTER(SetLossDesc), f32p, c_void_p],
    "detr_hip_assign_f32": [f32p, c_int32, c_int32, c_int32, f32p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p],
    "detr_hip_set_loss_sums_f32": [POINTER(SetLossDesc), c_void_p, f32p, c_void_p],
    "detr_hip_set_loss_finalize_f32": [f32p, c_int32, f32p, f32p, c_void_p],
    "detr_hip_set_loss_grad_f32": [POINTER(SetLossDesc), c_void_p, f32p, c_float, f32p, f32p, c_void_p],
    "detr_hip_sumsq_segments_f32": [f32p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, f32p, c_void_p],
    "detr_hip_clip_adam_f32": [f32p, f32p, f32p, f32p, c_void_p, c_void_p, c_void_p, c_void_p, f32p, f32p, c_int32, c_int32, c_void_p],
    "detr_hip_axpy_f32": [f32p, f32p, c_float, c_int64, c_void_p],
    "detr_hip_set_floats8_f32": [f32p] + [c_float] * 8 + [c_void_p],
}
# scratch sizing queries (return int64 bytes)
_SIGNATURES_I64 = {
    "detr_hip_attention_dropmask_words": [c_int32, c_int32, c_int32, c_int32],
    "detr_hip_colsum_det_scratch_floats": [c_int64, c_int32],
    "detr_hip_conv1x1_bwd_fused_workspace_floats": [c_int64],
    "detr_hip_workspace_bytes_gemm": [POINTER(GemmDesc)],
    "detr_hip_workspace_bytes_conv3x3": [POINTER(Conv3x3Desc), c_int32],
    "detr_hip_workspace_bytes_stem": [POINTER(StemDesc), c_int32],
    "detr_hip_workspace_bytes_layernorm": [POINTER(LayerNormDesc)],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + list(_SIGNATURES_I64) + ["detr_hip_last_error"])
ABI_VERSION = 9
# order of detr_hip_struct_layout's `which`
LAYOUT_STRUCTS = (ReduceDesc, GemmDesc, Conv3x3Desc, StemDesc, LayerNormDesc, AttnDesc, SetLossDesc, InputDesc, PostprocessDesc)

_lib = None


class KernelProfiler:
    """Optional HIP-event timing of the GEMM-class launches (bench.py roofline leg): events are
    recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self, prealloc=0, f32=False):
        self.f32 = f32             # the whole pass runs the exact-f32 MFMA kernels (fp32 parity mode)
        self.records = []          # (family, flops, start_event, end_event, signature, bytes)
        # event objects are created (and recorded once, which instantiates the HIP event) BEFORE the timed region:
        # creating them per launch costs more host time than the record itself
        self._pool = []
        for _ in range(prealloc):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._pool.append(ev)

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def begin(self):
        ev = self._event()
        ev.record()
        return ev

    def end(self, family, flops, ev0, sig=None, nbytes=0.0):
        ev1 = self._event()
        ev1.record()
        self.records.append((family, flops, ev0, ev1, sig, nbytes))

    def by_shape(self, top=40):
        """Per (family, shape signature) totals, sorted by time: where the GEMM time goes."""
        torch.cuda.synchronize()
        agg = {}
        for fam, flops, e0, e1, sig, _nb in self.records:
            d = agg.setdefault((fam, sig), [0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
            d[2] += flops
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]
        return [{"family": k[0], "shape": k[1], "launches": v[0], "ms": round(v[1], 3),
                 "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2) if v[1] > 0 else 0.0} for k, v in rows]

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, flops, e0, e1, _sig, nb in self.records:
            d = out.setdefault(fam, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "mixed_s": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nb
            # this launch priced on the mixed roofline: max(FLOPs / MFMA peak of its arithmetic type, bytes / 8 TB/s)
            d["mixed_s"] += max(flops / (157.3e12 if fam.startswith("gemm_f32") or self.f32 else 2.5e15), nb / 8.0e12)
        return out


PROFILER = None
# Deferred split-K reductions: while DEFER is a list (the engine's backward pass), every split-K GEMM gets a private slab from
# the bump allocator over DEFER_WS and only DESCRIBES its reduction; flush_reduces() then reduces up to 16 slab sets per launch.
DEFER = None
DEFER_WS = None
_defer_top = 0
_defer_outs = set()
COMPUTE_BF16 = 0      # default compute mode of gemm / conv3x3: 0 = exact fp32 MFMA, 1 = bf16 MFMA (fp32 storage), 2 = f32x3 (fp32 accuracy on the bf16 pipe)
WORKSPACE = None      # fp32 scratch tensor for the deterministic split-K reductions (set by the engine)


WS_GENERATION = 0     # bumped whenever the shared scratch is replaced: recorded hipGraphs that address the old one are stale
_WS_RETIRED = []      # outgrown scratch tensors: kept until the next backward pass begins (launches on either stream may still read them)
_DEVICE = None


def ensure_workspace(device, floats=2 * 1024 * 1024):
    """Shared scratch of the immediate (non-queued) deterministic split reductions: starts small and grows to what the
    launches ask for (need_workspace; sizes come from detr_hip_workspace_bytes_*), instead of a fixed 256 MB."""
    global WORKSPACE, _DEVICE, WS_GENERATION
    _DEVICE = torch.device(device)
    if WORKSPACE is None or WORKSPACE.device != _DEVICE or WORKSPACE.numel() < floats:
        if WORKSPACE is not None:
            _WS_RETIRED.append(WORKSPACE)
            WS_GENERATION += 1
        WORKSPACE = torch.empty(floats, dtype=torch.float32, device=device)
    return WORKSPACE


def need_workspace(nbytes):
    """Grow the shared scratch to at least nbytes (geometric steps; only ever happens in the first passes over a shape --
    a recorded hipGraph replays shapes an eager pass has already sized)."""
    if WORKSPACE is not None and nbytes > WORKSPACE.numel() * 4:
        ensure_workspace(WORKSPACE.device, max((int(nbytes) + 3) // 4, WORKSPACE.numel() * 3 // 2))
    return WORKSPACE


def workspace_bytes_held():
    """Scratch currently held by this module (shared workspace + slab pool of the queued reductions)."""
    n = WORKSPACE.numel() * 4 if WORKSPACE is not None else 0
    return n + sum(t.numel() * 4 for t in _DEFER_CHUNKS) + sum(t.numel() * 4 for t in _WS_RETIRED)


_WS_NEED = {}


def _ws_query(fn, key, *args):
    """detr_hip_workspace_bytes_* through a per-shape cache (one ctypes call per distinct shape, none per launch)."""
    v = _WS_NEED.get(key)
    if v is None:
        v = int(getattr(load(), fn)(*args))
        if v < 0:
            raise RuntimeError(f"{fn} rejected the descriptor: {load().detr_hip_last_error().decode()}")
        _WS_NEED[key] = v
    return v


DEFER_LIMIT = int(os.environ.get("DETR_HIP_DEFER_MB", "512")) * 1024 * 1024     # pending slab bytes that trigger a flush
DEFER_CHUNK = 64 * 1024 * 1024     # the slab pool grows in chunks of this size (a larger slab gets a chunk of its own)
_DEFER_CHUNKS = []                 # device tensors; a flush rewinds the cursor to chunk 0, nothing is freed
_defer_chunk_i = 0


def ensure_defer_workspace(device):
    """Slab pool of the queued split-K reductions.  It used to be 2 x DEFER_LIMIT = 1 GB up front; it now grows by 64 MB chunks
    to the high-water mark of what one backward pass queues between two flushes (sized by the shapes actually run)."""
    global _DEVICE
    _DEVICE = torch.device(device)
    if _DEFER_CHUNKS and _DEFER_CHUNKS[0].device != _DEVICE:
        _DEFER_CHUNKS.clear()
    return _DEFER_CHUNKS


def begin_deferred_reduces(device):
    global DEFER, _defer_top, _defer_outs, _defer_chunk_i, _defer_pending
    ensure_defer_workspace(device)
    del _WS_RETIRED[:]          # everything the previous pass launched has been ordered before this point on the main stream
    DEFER, _defer_top, _defer_outs, _defer_chunk_i, _defer_pending = [], 0, set(), 0, 0


AFTER_FLUSH = None     # hook of the engine: orders its two launch streams after a flush


def flush_reduces(end=False):
    """Reduce everything pending (bucket boundary / a consumer of the gradients / end of the backward pass): up to 16 slab
    sets per launch; the slabs become reusable."""
    global DEFER, _defer_top, _defer_outs, _defer_chunk_i, _defer_pending
    if DEFER is None:
        return
    if DEFER:
        arr = (ReduceDesc * len(DEFER))(*DEFER)
        _check(load().detr_hip_splitk_reduce_many(arr, len(DEFER), _stream()), "detr_hip_splitk_reduce_many")
        if AFTER_FLUSH is not None:
            AFTER_FLUSH()              # (two launch streams: the recycled slabs must not be handed to the other one early)
    _defer_top, _defer_outs, _defer_chunk_i, _defer_pending = 0, set(), 0, 0
    DEFER = None if end else []


def _defer_push(rds):
    """Queue the reductions a launch described.  Two reductions into the same output would race inside one grouped launch
    (read-modify-write of C): then the queue is flushed first and the new ones run right away, one launch each (their slabs
    must not outlive the flush, which recycles the slab pool).  A full slab pool flushes too."""
    rds = [r for r in rds if r is not None and r.splits > 0]
    outs = [r.C for r in rds] + [r.rs_out for r in rds if r.rs_out]
    if len(set(outs)) != len(outs) or any(o in _defer_outs for o in outs):
        flush_reduces()
        for r in rds:
            arr = (ReduceDesc * 1)(r)
            _check(load().detr_hip_splitk_reduce_many(arr, 1, _stream()), "detr_hip_splitk_reduce_many")
        if AFTER_FLUSH is not None:
            # the flush above reset the slab cursor, but these reductions still READ slabs above it: order the two launch
            # streams once more, so that the other stream is not handed an overlapping slab before they have run
            AFTER_FLUSH()
        return
    DEFER.extend(rds)
    _defer_outs.update(outs)
    if _defer_pending >= DEFER_LIMIT:
        flush_reduces()


_defer_pending = 0


def _defer_slab(nbytes):
    """256-byte aligned slab of the pool for one queued reduction: bump allocation over the chunk list (a flush rewinds it)."""
    global _defer_top, _defer_chunk_i, _defer_pending
    nbytes = (int(nbytes) + 255) & ~255
    while True:
        if _defer_chunk_i >= len(_DEFER_CHUNKS):
            _DEFER_CHUNKS.append(torch.empty(max(DEFER_CHUNK, nbytes) // 4, dtype=torch.float32, device=_DEVICE))
        chunk = _DEFER_CHUNKS[_defer_chunk_i]
        if _defer_top + nbytes <= chunk.numel() * 4:
            break
        _defer_chunk_i += 1
        _defer_top = 0
    ptr_ = chunk.data_ptr() + _defer_top
    _defer_top += nbytes
    _defer_pending += nbytes
    return ptr_, nbytes


def load():
    """Load libdetr_hip.so; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C detr-tensorflow_amd`) first; there is no CPU/eager fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int32
    for name, argtypes in _SIGNATURES_I64.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int64
    lib.detr_hip_last_error.argtypes = []
    lib.detr_hip_last_error.restype = ctypes.c_char_p
    if lib.detr_hip_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {lib.detr_hip_abi_version()}, this binding speaks {ABI_VERSION}: rebuild the library")
    check_struct_layouts(lib)
    _lib = lib
    return lib


def struct_layout_mirror(S):
    """[sizeof, offsetof(field 0), ...] of a ctypes mirror, in declaration order."""
    return [ctypes.sizeof(S)] + [getattr(S, f[0]).offset for f in S._fields_]


def check_struct_layouts(lib):
    """The descriptor structs are mirrored by hand above: compare every size / field offset with what the library was
    compiled with (detr_hip_struct_layout) -- a drifted mirror would silently scramble arguments."""
    for which, S in enumerate(LAYOUT_STRUCTS):
        want = struct_layout_mirror(S)
        buf = (c_int32 * 96)()
        n = lib.detr_hip_struct_layout(which, buf, 96)
        got = list(buf[:max(n, 0)])
        if n != len(want) or got != want:
            raise RuntimeError(f"ctypes mirror of struct {which} ({S.__name__}) does not match {LIB_PATH}: library {got}, mirror {want}")


def set_tuning(name, value):
    """Set (value=None: unset) a DETR_HIP_<NAME> tuning variable of the LIBRARY and have it re-read: the library reads its
    tuning variables once at load time, never on the launch path."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    _check(load().detr_hip_reload_tuning(), "detr_hip_reload_tuning")
    _WS_NEED.clear()        # the cached scratch sizes depend on the tuning variables (tile / split plans, slab layout)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().detr_hip_last_error().decode()}")


def ptr(t):
    return None if t is None else t.data_ptr()


def _f32(t, name="tensor"):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA float32 tensor, got {t.dtype} on {t.device}")
    return t


# ------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------
def _gemm_desc(M, N, K, A, lda, a_kcontig, B, ldb, b_kcontig, C, ldc, *, alpha=1.0, scale=None, bias=None,
               residual=None, ldr=0, mask=None, ldmask=0, act=0, split_k=1, batch=1, batch_inner=1,
               sA=(0, 0), sB=(0, 0), sC=(0, 0), a_off=0, b_off=0, c_off=0, workspace=None, dropout_p=0.0, dropout_seed=0,
               dropout_step=None, compute=None, rowsum_a=None, rowsum_alpha=1.0, ws_slice=None, maskbits_out=None, ln=None):
    """Fills a detr_gemm_desc; returns (desc, profiler info).  ws_slice = (index, count): this call's share of WORKSPACE.
    ln: dict(gamma, beta, y, mean, rstd, eps[, add, y2][, y16]) -- the LayerNorm of the output rows from the same launch."""
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    if ln is not None:
        d.ln_gamma, d.ln_beta, d.ln_y = ln["gamma"].data_ptr(), ln["beta"].data_ptr(), ln["y"].data_ptr()
        d.ln_mean, d.ln_rstd, d.ln_eps = ln["mean"].data_ptr(), ln["rstd"].data_ptr(), float(ln["eps"])
        if ln.get("y2") is not None:
            d.ln_add, d.ln_add_rows, d.ln_y2 = ln["add"].data_ptr(), ln["add"].shape[0], ln["y2"].data_ptr()
        d.ln_y16 = ptr(ln.get("y16"))
    d.A, d.lda, d.a_kcontig = A.data_ptr() + A.element_size() * a_off, lda, int(a_kcontig)
    is16 = lambda t: 1 if (t is not None and t.dtype == torch.bfloat16) else 0      # bf16 activation storage
    d.a_dtype, d.c_dtype, d.r_dtype, d.m_dtype = is16(A), is16(C), is16(residual), is16(mask)
    if mask is not None and mask.dtype == torch.uint8:
        d.m_dtype = 2                      # bit-packed ReLU mask: one byte per 8 columns, ldmask = row pitch in bytes
    if maskbits_out is not None:           # also emit (C > 0) as bits (the ReLU mask of this output for a later backward GEMM)
        d.maskbits_out, d.ld_maskbits_out = maskbits_out.data_ptr(), maskbits_out.stride(0)
    d.B, d.ldb, d.b_kcontig = B.data_ptr() + B.element_size() * b_off, ldb, int(b_kcontig)
    d.b_dtype = 1 if B.dtype == torch.bfloat16 else 0          # bf16 weight shadow (bf16 compute only)
    d.C, d.ldc = C.data_ptr() + C.element_size() * c_off, ldc
    d.batch, d.batch_inner = batch, batch_inner
    d.sA0, d.sA1 = sA
    d.sB0, d.sB1 = sB
    d.sC0, d.sC1 = sC
    d.alpha = alpha
    d.scale, d.bias = ptr(scale), ptr(bias)
    d.residual, d.ldr = ptr(residual), ldr
    d.mask, d.ldmask = ptr(mask), ldmask
    d.act, d.split_k = act, split_k
    d.dropout_p, d.dropout_seed = dropout_p, dropout_seed & 0xFFFFFFFF        # dropout_seed = SITE id; the step seed lives on the device
    d.dropout_step = ptr(dropout_step)
    d.compute = COMPUTE_BF16 if compute is None else int(compute)
    d.rowsum_a, d.rowsum_alpha = ptr(rowsum_a), rowsum_alpha
    # scratch of the deterministic split-K path: what the LIBRARY asks for (detr_hip_workspace_bytes_gemm -- the tile-ordered
    # slabs of round 4 are padded to whole tiles, so split_k * M * N floats is no longer the whole story), per-shape cached
    ws_need = 0
    if split_k > 1 and batch == 1 and workspace is None and (WORKSPACE is not None or DEFER is not None):
        ws_need = (_ws_query("detr_hip_workspace_bytes_gemm",
                             ("gemm", M, N, K, int(a_kcontig), int(b_kcontig), split_k, d.compute, d.a_dtype, d.b_dtype, ldc % 4,
                              (d.C or 0) % 16, rowsum_a is not None, scale is not None and (d.scale or 0) % 16), byref(d)) + 255) & ~255
    if workspace is None and WORKSPACE is not None and split_k > 1 and batch == 1 and DEFER is None:
        # immediate deterministic split-K: the shared scratch must hold every member's (256-byte aligned) share
        need_workspace(ws_need * (ws_slice[1] if ws_slice else 1))
    ws = workspace if workspace is not None else WORKSPACE
    if ws is None:
        d.workspace, d.workspace_bytes = None, 0
    elif ws_slice is None:
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    else:                      # members of a grouped launch run concurrently: disjoint, 256-byte aligned shares
        idx, cnt = ws_slice
        share = (ws.numel() * 4 // cnt) & ~255
        d.workspace, d.workspace_bytes = ws.data_ptr() + idx * share, share
    rd = None
    if DEFER is not None and split_k > 1 and batch == 1 and workspace is None:
        slab = _defer_slab(ws_need)
        if slab is not None:
            rd = ReduceDesc()
            d.workspace, d.workspace_bytes = slab
            d.defer_out = ctypes.pointer(rd)
    # family = the kernel BODY (every template instantiation -- layouts, storage types, tile sizes, grouped launches -- pooled): asked from the
    # library (detr_hip_gemm_family = the decision of the launch itself), only while a profiler is recording
    fam = "gemm_bf16c_kernel" if d.compute == 1 else "gemm_f32_kernel"
    if PROFILER is not None and d.compute == 1 and load().detr_hip_gemm_family(byref(d)) == 1:
        fam = "gemm_stream_bf16_kernel"          # one kernel body; its K / layout / epilogue instantiations are pooled
    sig = (f"M{M} N{N} K{K} b{batch} ak{int(a_kcontig)} bk{int(b_kcontig)} a16{d.a_dtype} b16{d.b_dtype} sk{split_k}"
           f"{' res' if residual is not None else ''}{' mask' if mask is not None else ''}")
    nbytes = float(batch) * (A.element_size() * M * K + B.element_size() * K * N + C.element_size() * M * N
                             + (residual.element_size() * M * N if residual is not None else 0)
                             + ((M * N // 8 if mask.dtype == torch.uint8 else mask.element_size() * M * N) if mask is not None else 0)
                             + (M * N // 8 if maskbits_out is not None else 0))     # algorithmic bytes at the stored widths
    return d, (fam, 2.0 * M * N * K * batch, sig, nbytes, rd)


def gemm(*args, **kw):
    """C = epi(A @ B) on raw layouts (see detr_gemm_desc).  *_off are element offsets."""
    d, (fam, flops, sig, nbytes, rd) = _gemm_desc(*args, **kw)
    ev0 = PROFILER.begin() if PROFILER is not None else None
    _check(load().detr_hip_gemm_f32(byref(d), _stream()), "detr_hip_gemm_f32")
    if rd is not None:
        _defer_push([rd])
    if ev0 is not None:
        PROFILER.end(fam, flops, ev0, sig, nbytes)


def gemm_group(calls):
    """Several independent GEMMs (list of (args, kwargs) of gemm()) issued through detr_hip_gemm_group_f32: members that share
    one kernel variant (e.g. the Q / K / V projections of an attention block, their three dgrads, their three weight
    gradients) run as ONE launch (+ one grouped split-K reduction); anything else falls back to sequential launches."""
    n = len(calls)
    arr = (GemmDesc * n)()
    infos = []
    for i, (a, kw) in enumerate(calls):
        d, info = _gemm_desc(*a, ws_slice=(i, n), **kw)
        arr[i] = d
        infos.append(info)
    ev0 = PROFILER.begin() if PROFILER is not None else None
    _check(load().detr_hip_gemm_group_f32(arr, n, _stream()), "detr_hip_gemm_group_f32")
    if DEFER is not None:
        _defer_push([info[4] for info in infos])
    if ev0 is not None:
        PROFILER.end(infos[0][0], sum(i[1] for i in infos), ev0, f"group{n}: " + infos[0][2], sum(i[3] for i in infos))


# workgroups a split-K launch aims for (A/B hooks).  128x128 tiles: 512 -> 256 measured -0.23 ms per step on the same box (half
# the partial slabs to write and reduce; scripts/micro_wgrad.py shows the same 3-7 % per launch with cold caches)
SPLIT_TARGET_128 = int(os.environ.get("DETR_HIP_SPLIT_TARGET", "256"))
SPLIT_TARGET_64 = int(os.environ.get("DETR_HIP_SPLIT_TARGET64", "1024"))
X3_SPLIT_TARGET = int(os.environ.get("DETR_HIP_X3_SPLIT_TARGET", "512"))      # workgroups a split-K f32x3 GEMM on 128 x 128 tiles aims at


def pick_split_k(M, N, K, max_split=1024):
    """Reduction-heavy GEMMs (weight gradients): split K so that enough workgroups exist to fill 256 CUs."""
    if COMPUTE_BF16 == 1 and ((N >= 128 and K >= 16384) or (M >= 512 and N >= 512 and K >= 4096) or
                         (K >= 4096 and ((M >= 256 and N >= 2048) or (M >= 2048 and N >= 256)))):   # 128x128 tiles (gemm_f32.hip: gemm_pick_tile)
        tiles = -(-M // 128) * -(-N // 128)
        ktiles = -(-K // 32)
        return int(max(1, min(max(1, SPLIT_TARGET_128 // tiles), max_split, ktiles // 8 if ktiles >= 16 else 1)))
    if COMPUTE_BF16 == 2 and X3_SPLIT_TARGET > 0 and N >= 128 and K >= 256 and M >= 128:
        # f32x3 (gemm_x3.h): 128 x 128 tiles, two workgroups per CU = 512 slots; the 64 x 64 plan below left e.g. the 256 x 1024 weight gradient of layer3
        # with 16 tiles x 16 splits = 256 workgroups, half a round of single workgroups per CU (round 6; DETR_HIP_X3_SPLIT_TARGET=0: the old plan)
        t128 = -(-M // 128) * -(-N // 128)
        ktiles = -(-K // 32)
        cap = ktiles // 8 if ktiles >= 64 else max(1, ktiles // 4)
        s128 = int(max(1, min(X3_SPLIT_TARGET // t128, max_split, cap)))
        if t128 * s128 >= 192:                      # (gemm_pick_tile: 128 x 128 tiles from 192 of them on)
            return s128
    # 64x64 tiles (fp32 always; bf16 for small outputs, see gemm_f32.hip): ~1024 workgroups measured best
    tiles = -(-M // 64) * -(-N // 64)
    want = max(1, SPLIT_TARGET_64 // max(tiles, 1))
    ktiles = -(-K // (32 if COMPUTE_BF16 == 1 else 16))
    cap = ktiles // 8 if ktiles >= 64 else ktiles // 4      # short reductions: 4 k-tiles per split are enough
    return int(max(1, min(want, max_split, cap)))


def linear_fwd_call(x2d, w_out_in, bias, out2d, *, alpha=1.0, residual=None, act=0, dropout_p=0.0, dropout_seed=0,
                    dropout_step=None, ln=None):
    """(args, kwargs) of the gemm() that computes out = act((x @ W^T + b) * alpha + residual); W is (out, in) like
    custom_layers.Linear.  The *_call forms exist so that independent Linear products can be issued with gemm_group()."""
    M, K = x2d.shape
    N = w_out_in.shape[0]
    return ((M, N, K, x2d, x2d.stride(0), 1, w_out_in, w_out_in.stride(0), 1, out2d, out2d.stride(0)),
            dict(alpha=alpha, bias=bias, residual=residual, ldr=(residual.stride(0) if residual is not None else 0), act=act,
                 dropout_p=dropout_p, dropout_seed=dropout_seed, dropout_step=dropout_step, **({"ln": ln} if ln is not None else {})))


def linear_dgrad_call(dy2d, w_out_in, dx2d, *, alpha=1.0, residual=None, mask=None):
    """dx = (dy @ W) * alpha (+ residual) (masked by mask > 0)."""
    M, N = dy2d.shape
    K = w_out_in.shape[1]
    return ((M, K, N, dy2d, dy2d.stride(0), 1, w_out_in, w_out_in.stride(0), 0, dx2d, dx2d.stride(0)),
            dict(alpha=alpha, residual=residual, ldr=(residual.stride(0) if residual is not None else 0),
                 mask=mask, ldmask=(mask.stride(0) if mask is not None else 0)))


def linear_wgrad_call(dy2d, x2d, dw_out_in, *, alpha=1.0, bias_grad=None):
    """dW(out,in) += alpha * dy^T @ x (deterministic split-K through the workspace; dW holds zeros / the running sum);
    bias_grad (out,) += alpha * column sums of dy, fused into the same launch (row sums of the A operand)."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    sk = pick_split_k(N, K, M)
    args = (N, K, M, dy2d, dy2d.stride(0), 0, x2d, x2d.stride(0), 0, dw_out_in, dw_out_in.stride(0))
    if sk == 1:
        return args, dict(alpha=alpha, residual=dw_out_in, ldr=dw_out_in.stride(0), rowsum_a=bias_grad, rowsum_alpha=alpha)
    return args, dict(alpha=alpha, split_k=sk, rowsum_a=bias_grad, rowsum_alpha=alpha)


def linear_fwd(*a, **kw):
    args, kwargs = linear_fwd_call(*a, **kw)
    gemm(*args, **kwargs)


def linear_dgrad(*a, **kw):
    args, kwargs = linear_dgrad_call(*a, **kw)
    gemm(*args, **kwargs)


def linear_wgrad(*a, **kw):
    args, kwargs = linear_wgrad_call(*a, **kw)
    gemm(*args, **kwargs)


# ------------------------------------------------------------------------------------------
# conv
# ------------------------------------------------------------------------------------------
def conv1x1_bwd_fused_scratch_floats(M):
    return int(load().detr_hip_conv1x1_bwd_fused_workspace_floats(int(M)))


def conv1x1_bwd_fused(dy, a, w, da, dw, scratch, *, scale=None, use_mask=True, alpha=1.0):
    """Backward of a 64 -> 256 channel 1x1 convolution in one pass over dy (csrc/bwd_fused.hip; resnet_backbone.py:116-137):
    da = (a > 0 if use_mask) * (dy @ w^T), dw += alpha * scale[n] * (a^T @ dy).  dy [M, 256], a [M, 64], w [64, 256], da [M, 64] bf16;
    dw [64, 256], scale [256], scratch (conv1x1_bwd_fused_scratch_floats(M) floats) fp32."""
    M = dy.shape[0]
    for t, shape, dt in ((dy, (M, 256), torch.bfloat16), (a, (M, 64), torch.bfloat16), (w, (64, 256), torch.bfloat16), (da, (M, 64), torch.bfloat16),
                         (dw, (64, 256), torch.float32)):
        if tuple(t.shape) != shape or t.dtype != dt or t.stride(1) != 1:
            raise ValueError(f"conv1x1_bwd_fused: operand of shape {tuple(t.shape)} / {t.dtype} / strides {t.stride()}, expected {shape} {dt}, unit column stride")
    if scratch.dtype != torch.float32 or (scale is not None and (scale.dtype != torch.float32 or scale.numel() != 256)):
        raise ValueError("conv1x1_bwd_fused: scratch / scale must be fp32 (scale: 256 entries)")
    ev0 = PROFILER.begin() if PROFILER is not None else None
    _check(load().detr_hip_conv1x1_bwd_fused_bf16(dy.data_ptr(), dy.stride(0), a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), da.data_ptr(),
                                                  da.stride(0), 1 if use_mask else 0, dw.data_ptr(), dw.stride(0), ptr(scale), c_float(alpha), M,
                                                  a.shape[1], dy.shape[1], scratch.data_ptr(), scratch.numel(), _stream()),
           "detr_hip_conv1x1_bwd_fused_bf16")
    if ev0 is not None:
        PROFILER.end("conv1x1_bwd_fused", 4.0 * M * a.shape[1] * dy.shape[1], ev0, f"M{M} {a.shape[1]}->{dy.shape[1]}",
                     float(2 * M * (dy.shape[1] + 2 * a.shape[1])))


def conv3x3(mode, x, w, y, N, Hi, Wi, Ci, Ho, Wo, Co, stride, *, pad=1, alpha=1.0, scale=None, bias=None,
            residual=None, mask=None, act=0, split=0, compute=None, maskbits_out=None):
    d = Conv3x3Desc()
    d.N, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co, d.stride, d.pad = N, Hi, Wi, Ci, Ho, Wo, Co, stride, pad
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    is16 = lambda t: 1 if (t is not None and t.dtype == torch.bfloat16) else 0      # bf16 weight shadow / activation storage
    d.w_dtype, d.x_dtype, d.y_dtype, d.r_dtype, d.m_dtype = is16(w), is16(x), is16(y), is16(residual), is16(mask)
    if mask is not None and mask.dtype == torch.uint8:
        d.m_dtype = 2                      # bit-packed ReLU mask, one byte per 8 channels of a pixel
    d.maskbits_out = ptr(maskbits_out)     # mode 0: also emit (y > 0) as bits
    d.alpha = alpha
    d.scale, d.bias, d.residual, d.mask = ptr(scale), ptr(bias), ptr(residual), ptr(mask)
    d.act, d.split = act, split
    d.compute = COMPUTE_BF16 if compute is None else int(compute)
    if mode == 2 and WORKSPACE is not None:      # weight gradient: per-split partial kernels, sized by the library
        need_workspace(_ws_query("detr_hip_workspace_bytes_conv3x3", ("conv", N, Hi, Wi, Ci, Ho, Wo, Co, stride, pad, split, d.compute, d.x_dtype),
                                 byref(d), mode))
    d.workspace, d.workspace_bytes = (WORKSPACE.data_ptr(), WORKSPACE.numel() * 4) if WORKSPACE is not None else (None, 0)
    ev0 = PROFILER.begin() if PROFILER is not None else None
    _check(load().detr_hip_conv3x3_f32(byref(d), mode, _stream()), "detr_hip_conv3x3_f32")
    if ev0 is not None:
        rows = N * (Hi * Wi if mode == 1 else Ho * Wo)
        # FLOPs: every mode performs the forward convolution's multiply-adds, N Ho Wo 9 Ci Co (until round 5 the input gradient was counted over its
        # OUTPUT rows N Hi Wi: four times too many for the three stride-2 convolutions, which inflated conv3x3_dgrad's rate by 1.56x)
        PROFILER.end(("conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad")[mode], 2.0 * (N * Ho * Wo) * 9 * Ci * Co, ev0,
                     f"N{N} {Hi}x{Wi}x{Ci}->{Ho}x{Wo}x{Co} s{stride}",
                     float(x.element_size() * (N * Hi * Wi * Ci if mode != 1 else N * Ho * Wo * Co)
                           + (w.element_size() * 9 * Ci * Co if mode != 2 else w.element_size() * N * Ho * Wo * Co)
                           + y.element_size() * (N * Ho * Wo * Co if mode == 0 else (N * Hi * Wi * Ci if mode == 1 else 9 * Ci * Co))
                           + ((rows * (Ci if mode == 1 else Co) // 8 if mask.dtype == torch.uint8 else mask.element_size() * rows * (Ci if mode == 1 else Co))
                              if mask is not None else 0) + (rows * Co // 8 if maskbits_out is not None else 0)
                           + (residual.element_size() * rows * (Ci if mode == 1 else Co) if residual is not None else 0)))


def stem_conv(mode, img, w, y, N, H, W, Ho, Wo, *, alpha=1.0, scale=None, bias=None, act=0, split=0, compute=None):
    """Implicit-GEMM stem convolution (include/detr_hip.h detr_stem_desc): mode 0 forward, mode 2 weight gradient
    (w = dy, y = dw accumulated)."""
    d = StemDesc()
    d.N, d.H, d.W, d.Ho, d.Wo = N, H, W, Ho, Wo
    d.img, d.w, d.y = img.data_ptr(), w.data_ptr(), y.data_ptr()
    d.alpha = alpha
    d.scale, d.bias = ptr(scale), ptr(bias)
    d.act, d.split = act, split
    d.w_dtype = 1 if (mode == 2 and w.dtype == torch.bfloat16) else 0       # bf16 activation storage: dy / the output
    d.y_dtype = 1 if (mode == 0 and y.dtype == torch.bfloat16) else 0
    d.compute = COMPUTE_BF16 if compute is None else int(compute)
    if mode == 2 and WORKSPACE is not None:
        need_workspace(_ws_query("detr_hip_workspace_bytes_stem", ("stem", N, H, W, Ho, Wo, split, d.compute, d.w_dtype), byref(d), mode))
    d.workspace, d.workspace_bytes = (WORKSPACE.data_ptr(), WORKSPACE.numel() * 4) if WORKSPACE is not None else (None, 0)
    ev0 = PROFILER.begin() if PROFILER is not None else None
    _check(load().detr_hip_stem_conv7x7_f32(byref(d), mode, _stream()), "detr_hip_stem_conv7x7_f32")
    if ev0 is not None:
        M = N * Ho * Wo
        PROFILER.end("stem_conv7x7", 2.0 * M * 64 * 147, ev0, f"stem7x7 mode{mode} M{M}",
                     float(4 * N * H * W * 3 + (y if mode == 0 else w).element_size() * M * 64 + 4 * 147 * 64))


def call(name, *args):
    """Generic call by symbol name with the current stream appended."""
    _check(getattr(load(), name)(*args, _stream()), name)


def zero_(t):
    call("detr_hip_memset_zero", t.data_ptr(), t.numel() * t.element_size())
    return t


# ------------------------------------------------------------------------------------------
# attention / LayerNorm descriptors
# ------------------------------------------------------------------------------------------
def attention_dropmask_words(B, H, T, S):
    """uint32 words of the keep-bit buffer of one attention site (detr_hip_attention_dropmask_words)."""
    return int(load().detr_hip_attention_dropmask_words(B, H, T, S))


def attention_dropmask(mask, B, H, T, S, *, dropout_p, dropout_site=0, dropout_step=None):
    """Fills `mask` (int32 / uint32 tensor of attention_dropmask_words(...) words) with the keep bits of the attention-probability
    dropout of (site, *step): what the all-bf16 attention kernels read instead of hashing (detr_hip_attention_dropmask)."""
    d = AttnDesc()
    d.B, d.H, d.T, d.S = B, H, T, S
    d.dropout_p, d.dropout_site, d.dropout_step = dropout_p, dropout_site & 0xFFFFFFFF, ptr(dropout_step)
    if mask.numel() * mask.element_size() < 4 * attention_dropmask_words(B, H, T, S):
        raise ValueError("attention_dropmask: buffer too small")
    d.dropmask = mask.data_ptr()
    _check(load().detr_hip_attention_dropmask(byref(d), _stream()), "detr_hip_attention_dropmask")


def attention_dropmask_many(sites, B, H, *, dropout_p, dropout_step=None):
    """sites: list of (mask tensor, T, S, site id) -- the keep bits of every attention site of a step from ONE launch."""
    arr = (AttnDesc * len(sites))()
    for i, (mask, T, S, site) in enumerate(sites):
        d = arr[i]
        d.B, d.H, d.T, d.S = B, H, T, S
        d.dropout_p, d.dropout_site, d.dropout_step = dropout_p, site & 0xFFFFFFFF, ptr(dropout_step)
        if mask.numel() * mask.element_size() < 4 * attention_dropmask_words(B, H, T, S):
            raise ValueError("attention_dropmask_many: buffer too small")
        d.dropmask = mask.data_ptr()
    _check(load().detr_hip_attention_dropmask_many(arr, len(sites), _stream()), "detr_hip_attention_dropmask_many")


def attention(q, k, v, o, lse, B, H, T, S, *, scale=1.0, dropout_p=0.0, dropout_site=0, dropout_step=None, compute=None,
              d_o=None, dq=None, dk=None, dv=None, delta=None, dropmask=None):
    """Fused attention core (include/detr_hip.h detr_attn_desc).  q / k / v / o (and the gradients) are 2-D views
    [rows, >= H*32] whose row stride is taken from the tensor, so they may be column blocks of a packed projection buffer.
    Forward when d_o is None, otherwise the backward (dq / dk / dv / delta written).
    All-bf16 operands (io_dtype = 1, csrc/attention_dma.hip): every tensor bf16, q pre-multiplied by scale * log2(e), delta with
    2*B*H*T floats, and -- with dropout -- the keep bits of attention_dropmask() in `dropmask`."""
    d = AttnDesc()
    d.B, d.H, d.T, d.S = B, H, T, S
    io16 = q.dtype == torch.bfloat16
    want = torch.bfloat16 if io16 else torch.float32
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        if t.stride(1) != 1 or t.dtype != want:
            raise TypeError(f"attention: {name} must be {want} with a unit column stride")
        setattr(d, name, t.data_ptr())
        setattr(d, "ld" + name, t.stride(0))
    d.lse = lse.data_ptr()
    d.scale, d.dropout_p, d.dropout_site, d.dropout_step = scale, dropout_p, dropout_site & 0xFFFFFFFF, ptr(dropout_step)
    d.compute = COMPUTE_BF16 if compute is None else int(compute)
    d.io_dtype = 1 if io16 else 0
    if io16:
        d.compute = 1
        if dropout_p > 0.0:
            if dropmask is None:
                raise ValueError("attention: bf16 operands with dropout need the keep bits (attention_dropmask)")
            d.dropmask = dropmask.data_ptr()
    ev0 = PROFILER.begin() if PROFILER is not None else None
    esz = 2.0 if io16 else 4.0
    prods = 2.0 * B * H * T * S * 32                   # FLOPs of one [T,S,32] product over all (batch, head) problems
    io = esz * 256 * (2 * B * T + 2 * B * S)           # q, o + k, v rows, once each
    bits = (B * H * T * S / 8.0) if (io16 and dropout_p > 0.0) else 0.0      # keep bits, one layout per kernel
    if d_o is None:
        _check(load().detr_hip_attention_fwd(byref(d), _stream()), "detr_hip_attention_fwd")
        if ev0 is not None:
            PROFILER.end("attention_fwd", 2 * prods, ev0, f"B{B} H{H} T{T} S{S}", io + bits)
        return
    for name, ldn, t in (("d_o", "ldd_o", d_o), ("dq", "lddq", dq), ("dk", "lddk", dk), ("dv", "lddv", dv)):
        if t.stride(1) != 1 or t.dtype != want:
            raise TypeError(f"attention: {name} must be {want} with a unit column stride")
        setattr(d, name, t.data_ptr())
        setattr(d, ldn, t.stride(0))
    if io16 and delta.numel() < 2 * B * H * T:
        raise ValueError("attention: bf16 operands need 2*B*H*T floats of `delta` scratch")
    d.delta = delta.data_ptr()
    _check(load().detr_hip_attention_bwd(byref(d), _stream()), "detr_hip_attention_bwd")
    if ev0 is not None:      # dQ kernel: S, dP, dQ; dK/dV kernel: S, dP, dV, dK (the probabilities are recomputed twice)
        PROFILER.end("attention_bwd", 7 * prods, ev0, f"B{B} H{H} T{T} S{S}", 2.5 * io + 2 * bits)


def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps, *, add=None, y2=None, y16=None):
    d = LayerNormDesc()
    d.rows, d.C, d.eps = x.shape[0], x.shape[1], eps
    d.x, d.gamma, d.beta, d.y, d.mean, d.rstd = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    if y2 is not None:
        d.add, d.add_rows, d.y2 = add.data_ptr(), add.shape[0], y2.data_ptr()
    d.y16 = ptr(y16)
    _check(load().detr_hip_layernorm_fwd(byref(d), _stream()), "detr_hip_layernorm_fwd")


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, *, dx_add=None, dx_drop=None, dropout_p=0.0, dropout_site=0,
                  dropout_step=None, dx_drop16=None, defer=True, dy_add=None):
    """defer=False: finish the gamma / beta reduction at once even while reductions are being queued (a LayerNorm whose
    parameters receive several backward passes per step -- two queued reductions into one output would force a flush)."""
    d = LayerNormDesc()
    d.rows, d.C = x.shape[0], x.shape[1]
    d.dy, d.x, d.gamma, d.mean, d.rstd = dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    d.dx, d.dgamma, d.dbeta = dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr()
    d.dy_add = ptr(dy_add)
    ln_need = _ws_query("detr_hip_workspace_bytes_layernorm", ("ln", d.rows, d.C), byref(d))
    if WORKSPACE is not None:                                       # deterministic gamma / beta reduction
        need_workspace(ln_need)
        d.workspace, d.workspace_bytes = WORKSPACE.data_ptr(), WORKSPACE.numel() * 4
    blocks = None
    if DEFER is not None and defer:  # queue the gamma / beta finish with the split-K reductions (the optimiser is their only reader)
        slab = _defer_slab(ln_need)
        if slab is not None:
            blocks = c_int32(0)
            d.workspace, d.workspace_bytes = slab
            d.defer_blocks_out = ctypes.pointer(blocks)
    d.dx_add = ptr(dx_add)
    if dx_drop is not None or dx_drop16 is not None:
        d.dx_drop, d.dx_drop16 = ptr(dx_drop), ptr(dx_drop16)
        d.dropout_p, d.dropout_site, d.dropout_step = dropout_p, dropout_site & 0xFFFFFFFF, ptr(dropout_step)
    _check(load().detr_hip_layernorm_bwd(byref(d), _stream()), "detr_hip_layernorm_bwd")
    if blocks is not None:
        C = d.C
        rds = []
        for off, out in ((0, dgamma), (C, dbeta)):
            r = ReduceDesc()
            r.ws, r.splits, r.part_stride, r.rows, r.cols = d.workspace + 4 * off, blocks.value, 2 * C, 1, C
            r.C, r.ldc, r.alpha = out.data_ptr(), C, 1.0
            rds.append(r)
        _defer_push(rds)


def copy_table(entries, device):
    """Device table for detr_hip_multi_copy: entries = [(src tensor, dst tensor, mode)], equal byte sizes, multiples of 16."""
    rows = []
    for src, dst, mode in entries:
        nb = src.numel() * src.element_size()
        assert nb == dst.numel() * dst.element_size() and nb % 16 == 0 and src.is_contiguous() and dst.is_contiguous()
        rows.append([src.data_ptr(), dst.data_ptr(), nb // 16, mode])      # (mode, reserved) share one int64 slot: little endian
    return torch.tensor(rows, dtype=torch.int64).to(device)


def multi_copy(table, blocks_per_entry=16):
    call("detr_hip_multi_copy", table.data_ptr(), table.shape[0], blocks_per_entry)
