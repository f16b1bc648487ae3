"""DETR execution engine for MI355X: the forward and the hand-written backward of the whole
hot path (ResNet backbone -> 6+6 transformer -> heads) as an explicit sequence of launches of
the HIP kernels in libdetr_hip.so.  No autograd graph, no allocation in the step: every
activation lives in a named, cached device buffer (static memory plan, hipGraph friendly).

Internal layout: NHWC feature maps, i.e. batch-first token matrices [B*L, 256]; this is the
reference's [L, B, 256] (transformer.py:32-33) with the two transposes removed -- every op of
the transformer is row-wise except attention, which addresses (batch, head) through strides.

Reference lines implemented (paths relative to the reference root):
  backbone   detr_tf/networks/resnet_backbone.py:20-32,116-137, custom_layers.py:21-24
  pos enc    detr_tf/networks/position_embeddings.py:23-50 (constant for a zero mask, detr.py:172)
  transformer detr_tf/networks/transformer.py:29-57,157-179,207-234,285-356
  heads      detr_tf/networks/detr.py:94-114,181-204
  backward   what tape.gradient (detr_tf/optimizers.py:115) computes for the graph above
"""
import math
import os
from ctypes import byref, c_float

import numpy as np
import torch

from . import _hip as hip
from .params import ParamStore, RESNET50_BLOCKS, block_names, bn_conv_pairs, stem_names

D = 256
HEADS = 8
HD = 32
FF = 2048
BN_EPS = 1e-5
KERAS_BN_EPS = 1.001e-5      # tf.keras.applications.resnet BatchNormalization(epsilon=1.001e-5) (tf_backbone=True)
LN_EPS = 1e-5
# bf16 STORAGE of the backbone activations and their gradients in precision="bf16" (default on; DETR_HIP_ACT16=0 keeps
# fp32 storage with bf16 MFMA operands only)
ACT16 = os.environ.get("DETR_HIP_ACT16", "1") != "0"
# bf16 STORAGE of the FFN hidden activation in precision="bf16" (bit-identical: it only feeds GEMM operands; DETR_HIP_H16=0 = fp32)
H16 = os.environ.get("DETR_HIP_H16", "1") != "0"
DEFER_REDUCE = os.environ.get("DETR_HIP_DEFER_REDUCE", "1") != "0"     # queue the weight gradients' split-K reductions (A/B switch)
WGRAD_STREAM = os.environ.get("DETR_HIP_WGRAD_STREAM", "1") != "0"      # backbone weight gradients on a second HIP stream (A/B switch)
MASK_BITS = os.environ.get("DETR_HIP_MASK_BITS", "1") != "0"            # ReLU masks of the block outputs as bits (A/B switch, round 4)
BWD_FUSED = os.environ.get("DETR_HIP_BWD_FUSED", "1") != "0"            # layer1: input + weight gradient of the 64 -> 256 1x1 convolutions in one pass over dY (round 5)
# bf16 compute mode: the LayerNorm behind an attention out-projection (2, default) / also behind the FFN's second Linear (1) runs in
# that GEMM's launch (detr_gemm_desc.ln_*, row-complete 32 x 256 tiles; same bits as the two launches); 0 = off.  Measured, same box:
# 16.53 / 16.56 ms (2) vs 16.53 / 16.61 (0) vs 16.85 / 16.96 (1): the K = 256 launches absorb their LayerNorm at equal time (18 launches
# fewer per step), the K = 2048 ones lose 0.35 ms -- a 32-row tile walks 64 K tiles behind one barrier each
LN_FUSE = int(os.environ.get("DETR_HIP_LN_FUSE", "2"))
# bf16 compute mode, round 6: the attention operands Q / K / V / O / dO / dQ / dK / dV are STORED in bf16 (the projection GEMMs write them, Q with
# scale * log2(e) as its alpha) and the attention core is csrc/attention_dma.hip (LDS-DMA K / V rings per wave, dropout keep flags as bits
# generated once per step).  DETR_HIP_ATTN16=0: the round-3 kernels on fp32 tensors (A/B switch)
ATTN16 = os.environ.get("DETR_HIP_ATTN16", "1") != "0"
# the NEXT step's keep bits generated during this step (eager launches): 2 (default) = next to the assignment kernel alone, behind the matcher's
# cost kernel (pregen_dropmasks_at_matcher: -0.05 ms per step, profiles/r06_ab_results.txt #5); 1 = from the start of the decoder forward
# (measured SLOWER, #3); 0 = at the head of their own step
DROPMASK_AHEAD = int(os.environ.get("DETR_HIP_DROPMASK_AHEAD", "2") or 0)
QK_ALPHA = float(HD) ** -0.5 * 1.4426950408889634      # what the stored bf16 query carries: softmax scale (transformer.py:307) * log2(e)


def mix32(x):
    """csrc/common.h::mix32."""
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def step_seed(base_seed, step_no, rank=0):
    """uint32 seed of one training step: base seed, step counter and data-parallel rank, mixed on the host (the kernels
    mix it once more with the dropout site id, csrc/common.h::drop_key)."""
    return mix32(mix32((base_seed + 0x9E3779B9 * step_no) & 0xFFFFFFFF) ^ ((rank * 0x85EBCA6B) & 0xFFFFFFFF))


def position_embedding_sine_host(H, W, num_pos_features=128, temperature=10000.0, eps=1e-6):
    """position_embeddings.py:23-50 for an all-False mask: [H*W, 256] fp32, [pos_y | pos_x]."""
    y = (np.arange(1, H + 1, dtype=np.float32) / np.float32(H + eps) * np.float32(2 * math.pi)).astype(np.float32)
    x = (np.arange(1, W + 1, dtype=np.float32) / np.float32(W + eps) * np.float32(2 * math.pi)).astype(np.float32)
    k = np.arange(num_pos_features, dtype=np.float32)
    dim_t = np.power(np.float32(temperature), (2 * np.floor(k / 2) / np.float32(num_pos_features)).astype(np.float32)).astype(np.float32)
    py = (y[:, None] / dim_t[None, :]).astype(np.float32)
    px = (x[:, None] / dim_t[None, :]).astype(np.float32)
    ey = np.empty_like(py)
    ex = np.empty_like(px)
    ey[:, 0::2], ey[:, 1::2] = np.sin(py[:, 0::2]), np.cos(py[:, 1::2])
    ex[:, 0::2], ex[:, 1::2] = np.sin(px[:, 0::2]), np.cos(px[:, 1::2])
    pos = np.concatenate([np.broadcast_to(ey[:, None, :], (H, W, num_pos_features)),
                          np.broadcast_to(ex[None, :, :], (H, W, num_pos_features))], axis=2)
    return np.ascontiguousarray(pos.reshape(H * W, 2 * num_pos_features), dtype=np.float32)


class DetrEngine:
    def __init__(self, device="cuda:0", blocks=RESNET50_BLOCKS, num_enc=6, num_dec=6, num_queries=100,
                 num_classes=92, nb_class=None, seed=0, tf_backbone=False):
        hip.load()
        self.device = torch.device(device)
        hip.ensure_workspace(self.device)
        if DEFER_REDUCE:
            hip.ensure_defer_workspace(self.device)        # allocated here, never inside a graph capture
        self.blocks = tuple(blocks)
        self.num_enc, self.num_dec, self.Q = num_enc, num_dec, num_queries
        self.nb_class = nb_class
        self.C = num_classes if nb_class is None else nb_class
        # tf_backbone=True (detr.py:146-148): tf.keras.applications ResNet50 -- ResNet v1 (the stride of a stage sits on the
        # FIRST 1x1 conv of its first block), convs with a trainable bias, BatchNormalization(eps 1.001e-5) in inference mode
        self.tf_backbone = bool(tf_backbone)
        self.bn_eps = KERAS_BN_EPS if self.tf_backbone else BN_EPS
        self.P = ParamStore(self.device, blocks, num_enc, num_dec, num_queries, num_classes, nb_class, seed, tf_backbone=tf_backbone)
        self._stem = stem_names(self.tf_backbone)
        self._pairs = bn_conv_pairs(self.blocks, self.tf_backbone)
        self._bufs = {}
        self._buf_gen = 0                # bumped whenever buf() replaces (frees) a buffer: recorded graphs of older generations are invalid
        self._pos_cache = {}
        self._shape = None
        self.bn_scale, self.bn_shift = {}, {}
        self._weights_version, self._built = 0, {}
        self.P.on_change = self._params_changed   # ParamStore.load / load_dict: refold the frozen BN, bump the weights version
        self.fold_bn()
        self.compute = 0                 # 0 = exact fp32 MFMA (parity mode); 1 = bf16 MFMA, fp32 storage (config C3)
        # fp32 storage mode only: the GEMMs / convolutions / stem run on the bf16 matrix pipe at fp32 accuracy (detr_gemm_desc.compute = 2,
        # csrc/gemm_core.h: mma_ktile_split3) instead of on v_mfma_f32_32x32x2_f32; everything else of the fp32 mode is unchanged
        self.f32_split = False
        self.dropout_p = 0.1             # Transformer(dropout=0.1) transformer.py:9 -- active when training=True
        self.dropout_seed = 0x5EED       # base seed; mixed with the step counter and the data-parallel rank
        self.dp_rank = 0                 # parallel.DataParallel sets it: every rank draws its own masks
        self._step_no = 0
        self._drop = (0.0, 0)
        self._step_seed = 0
        self._seed_dev = torch.zeros(8, dtype=torch.int32, device=self.device)   # [0] = uint32 seed of the current training step
        self._cross = {}
        self._graph_replay = False       # True while a captured step is being recorded / replayed (training.GraphedTrainStep)
        self.phase_events = None         # list of (name, torch.cuda.Event) while a caller times the phases of a step (bench.py --phase-events)

    # Derived weight copies (BN-folded kernels, the bf16 shadow, the gathered cross-attention weights) are stamped with the
    # weights version they were built from.  EVERY mutation of the parameters bumps the version -- the optimiser after an
    # apply, fold_bn, and ParamStore.load / load_dict themselves (through the store's on_change hook), so a direct
    # `engine.P.load(...)` cannot leave a recorded eval graph or a bf16 shadow stale -- and a copy is rebuilt exactly when
    # it is next needed, also when fp32 and bf16 passes are interleaved.
    @property
    def gemm_mode(self):
        """detr_gemm_desc.compute of this pass's GEMM / conv / stem launches: 1 = bf16 MFMA, 0 = exact fp32 MFMA, 2 = f32x3."""
        return 1 if self.compute == 1 else (2 if self.f32_split else 0)

    def bump_weights_version(self):
        self._weights_version += 1

    def _stale(self, key):
        if self._built.get(key) == self._weights_version:
            return False
        self._built[key] = self._weights_version
        return True

    def advance_dropout_step(self):
        """Next training step: a new uint32 step seed (base seed, step counter, DP rank) is written to DEVICE memory; every
        dropout site derives its key from it inside the kernels, so a captured hipGraph sees the new masks on replay."""
        self._step_no += 1
        self._step_seed = step_seed(self.dropout_seed, self._step_no, self.dp_rank)
        # slot 1: the seed of the step after this one (the keep bits of the next step's attention sites are generated while this
        # step's decoder / matcher leave most of the chip idle, _pregen_dropmasks)
        hip.call("detr_hip_set_u32x8", self._seed_dev.data_ptr(), self._step_seed, step_seed(self.dropout_seed, self._step_no + 1, self.dp_rank),
                 0, 0, 0, 0, 0, 0)
        self._drop = (float(self.dropout_p), self._step_seed)
        return self._step_seed

    def phase(self, name):
        """Phase marker of the launch sequence: a timing event on the current stream when `phase_events` is a list (measurement only)."""
        if self.phase_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.phase_events.append((name, ev))

    # ---- buffers ------------------------------------------------------------------------------
    def buf(self, name, shape, dtype=torch.float32):
        """Named device buffer of the static memory plan.  A request with another shape / dtype REPLACES the buffer (the
        old storage is freed: alternating train / validation shapes do not add up) and bumps `buf_generation`; every
        recorded hipGraph remembers the generation it was captured at and is dropped when it no longer matches
        (training.GraphedTrainStep, DetrModel._eval_forward) -- a replay would write through the freed addresses."""
        key = name
        t = self._bufs.get(key)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            if t is not None:
                self._buf_gen += 1
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    @property
    def buf_generation(self):
        """Changes whenever device memory a recorded hipGraph may address has been replaced: an engine buffer (buf) or the
        binding's shared split-reduction scratch (hip.need_workspace)."""
        return self._buf_gen + hip.WS_GENERATION

    def buffer_bytes(self):
        """Bytes held by the named buffers (activations, scratch, derived weight copies)."""
        return sum(t.numel() * t.element_size() for t in self._bufs.values())

    def fold_bn(self):
        """custom_layers.py:21-23: scale = w * rsqrt(var + eps), shift = b - mean * scale (frozen)."""
        for p, c in self.P.bn.items():
            raw = self.P.bn_raw[p]
            sc = self.buf(f"bnscale:{p}", (c,))
            sh = self.buf(f"bnshift:{p}", (c,))
            hip.call("detr_hip_bn_fold_f32", raw[0].data_ptr(), raw[1].data_ptr(), raw[2].data_ptr(), raw[3].data_ptr(),
                     sc.data_ptr(), sh.data_ptr(), c, c_float(self.bn_eps))
            self.bn_scale[p] = sc
            # a conv bias under the frozen BN moves into the shift: scale*(conv + b) + shift0 (rebuilt by _refresh_bias_shift)
            self.bn_shift[p] = self.buf(f"bnshift_eff:{p}", (c,)) if self.tf_backbone else sh
        self.bump_weights_version()
        self._fold_table = None
        self._shift_table = None

    def _refresh_bias_shift(self):
        """tf_backbone=True: effective shift = bias * scale + (beta - mean * scale) for every conv, one launch."""
        if self._shift_table is None:
            rows = []
            for bn_name, conv_name in self._pairs:
                c = self.P.bn[bn_name]
                rows.append([self.P.views[f"{conv_name}/bias"].data_ptr(), self.bn_scale[bn_name].data_ptr(),
                             self._bufs[f"bnshift:{bn_name}"].data_ptr(), self.bn_shift[bn_name].data_ptr(), c])
            self._shift_table = torch.tensor(rows, dtype=torch.int64).to(self.device)
        hip.call("detr_hip_fma_vec_group", self._shift_table.data_ptr(), self._shift_table.shape[0])

    def _params_changed(self):
        self.fold_bn()

    def load_params(self, params):
        return self.P.load_dict(params)       # (the store's on_change hook refolds the frozen BN and bumps the weights version)

    def _w(self, name):
        """Weight OPERAND of a GEMM: the fp32 tensor, or -- while the bf16-compute kernels are active -- its bf16 shadow
        (half the bytes, no conversion in the kernel).  Biases, LayerNorm vectors and gradients stay fp32."""
        if hip.COMPUTE_BF16 == 1 and self.P.views16 is not None:
            return self.P.views16[name]
        return self.P.views[name]

    def _refresh_shadow(self):
        """bf16 compute mode: one flat fp32 -> bf16 conversion of all parameters per optimiser step (~60 us)."""
        flat16 = self.P.shadow16()
        hip.call("detr_hip_cvt_bf16", self.P.flat.data_ptr(), flat16.data_ptr(), self.P.flat.numel())
        self._refold_group()

    def _refold_group(self):
        """Frozen-BN fold of every bf16 conv kernel (all but the stem) in one launch: a device table of
        (kernel, bn scale, bf16 out, n/4, cols/4) entries, built once -- the tensors it points at never move."""
        if getattr(self, "_fold_table", None) is None:
            rows = []
            for bn_name, conv in self._pairs[1:]:               # (the stem kernel takes the fp32 copy)
                conv_name = f"{conv}/kernel"
                w = self.P.views[conv_name]
                co = w.shape[-1]
                ws16 = self.buf(f"ws16:{conv_name}", w.shape, torch.bfloat16)
                rows.append([w.data_ptr(), self.bn_scale[bn_name].data_ptr(), ws16.data_ptr(), w.numel() // 4, co // 4])
            self._fold_table = torch.tensor(rows, dtype=torch.int64).to(self.device)
            self._fold_keep = [self.bn_scale, self._bufs]
        hip.call("detr_hip_scale_cols_bf16_group", self._fold_table.data_ptr(), self._fold_table.shape[0])

    def _scaled_kernel(self, conv_name, bn_name):
        """kernel * bn scale per output channel (the frozen-BN fold into the conv); bf16 compute mode: the bf16 copy."""
        w = self.P.views[conv_name]
        co = w.shape[-1]
        if self.compute == 1 and conv_name != f"{self._stem['conv']}/kernel":      # (the stem kernel takes the fp32 copy)
            return self.buf(f"ws16:{conv_name}", w.shape, torch.bfloat16)     # filled by _refold_group()
        ws = self.buf(f"ws:{conv_name}", w.shape)
        if self._stale(f"ws:{conv_name}"):
            hip.call("detr_hip_scale_cols_f32", w.data_ptr(), self.bn_scale[bn_name].data_ptr(), ws.data_ptr(),
                     w.numel() // co, co)
        return ws

    def _conv1x1_fwd(self, x, M, cin, cout, conv_name, bn_name, out, residual=None, act=1, maskbits_out=None):
        """1x1 conv + frozen BN (+ residual) (+ ReLU) as one GEMM (resnet_backbone.py:119-121,128-135).
        B = scaled kernel [cin][cout] (N contiguous; in bf16 mode it is staged as a transpose-read LDS image, which
        made the former transposed weight copy unnecessary).  maskbits_out: also emit (out > 0) as one byte per 8 channels."""
        ws = self._scaled_kernel(conv_name, bn_name)
        shift = self.bn_shift[bn_name]
        ldr = cout if residual is not None else 0
        hip.gemm(M, cout, cin, x, cin, 1, ws, cout, 0, out, cout, bias=shift, residual=residual, ldr=ldr, act=act,
                 maskbits_out=maskbits_out)

    # ---- small helpers ----------------------------------------------------------------------------
    @staticmethod
    def _wgrad(M_out, N_out, K_red, A, lda, Bm, ldb, C, ldc, scale=None, alpha=1.0):
        """C[M_out,N_out] += alpha * scale[n] * A'^T-style reduction (both operands reduction-major)."""
        sk = hip.pick_split_k(M_out, N_out, K_red)
        if sk > 1:
            hip.gemm(M_out, N_out, K_red, A, lda, 0, Bm, ldb, 0, C, ldc, alpha=alpha, scale=scale, split_k=sk)
        else:
            hip.gemm(M_out, N_out, K_red, A, lda, 0, Bm, ldb, 0, C, ldc, alpha=alpha, scale=scale, residual=C, ldr=ldc)

    def _colsum(self, x2d, out, alpha=1.0):
        """out += alpha * column sums of x2d, in a FIXED order (round 5: the atomic form made input_proj/bias the one gradient of
        the step that differed between two runs, scripts/experiments/determinism_full.py)."""
        rows, cols = x2d.shape
        n = int(hip.load().detr_hip_colsum_det_scratch_floats(rows, cols))
        scratch = self.buf(f"colsum:{rows}x{cols}", (n,))
        hip.call("detr_hip_colsum_det_f32", x2d.data_ptr(), out.data_ptr(), rows, cols, x2d.stride(0), c_float(alpha),
                 scratch.data_ptr(), n)

    def _ln_fwd(self, x, pfx, y, tag, add=None, y2=None, y16=None):
        """LayerNormalization(eps 1e-5) transformer.py:151-152; optional fused y2 = y + add[r % rows(add)] -- the
        `+ pos` / `+ query_pos` operand of the next attention block (transformer.py:161-163,209,219); optional bf16 twin."""
        rows = x.shape[0]
        mean = self.buf(f"{tag}:mean", (rows,))
        rstd = self.buf(f"{tag}:rstd", (rows,))
        hip.layernorm_fwd(x, self.P.views[f"{pfx}/gamma"], self.P.views[f"{pfx}/beta"], y, mean, rstd, LN_EPS, add=add, y2=y2,
                          y16=y16)

    def _ln_spec(self, rows, pfx, y, tag, add=None, y2=None, y16=None, ffn=False):
        """The same LayerNorm as an epilogue of the GEMM that produces its input (hip.linear_fwd(..., ln=spec)); None when this
        pass runs them as two launches (fp32 compute mode, DETR_HIP_LN_FUSE)."""
        if self.compute != 1 or LN_FUSE == 0 or (ffn and LN_FUSE == 2) or (ffn and not self.ffn16):
            return None
        return dict(gamma=self.P.views[f"{pfx}/gamma"], beta=self.P.views[f"{pfx}/beta"], y=y, mean=self.buf(f"{tag}:mean", (rows,)),
                    rstd=self.buf(f"{tag}:rstd", (rows,)), eps=LN_EPS, add=add, y2=y2, y16=y16)

    # ---- second launch stream ----------------------------------------------------------------------------------------------
    # Work that does not feed the critical chain is issued on a second HIP stream behind an event of the main stream that
    # follows its producer (a graph edge when the pass is captured) and is joined where its result is read:
    #   backward: every weight gradient (it feeds nothing but the bucket exchange / the optimiser) and the data gradient of a
    #             stage's projection shortcut; forward: the projection shortcut and the decoder's shared K / V projection.
    # They share the chip with the chain: the decoder's 800-row kernels fill a fifth of the CUs, the backbone's HBM-bound 1x1
    # GEMMs leave the MFMA pipes idle.  The tensors a side launch reads are never rewritten within the same pass (_sx: one
    # scratch tensor per site instead of one per shape; the backbone loop guards its recycled tensors with per-block events),
    # so the only joins are in front of a reduction flush, a bucket hand-over, the consumer of a shortcut / of K, V, and the
    # end of the pass.  Off (one stream): DETR_HIP_WGRAD_STREAM=0, and any backward with immediate split-K reductions.
    def _side_begin(self, on):
        self._side_on = bool(on)
        hip.AFTER_FLUSH = self._side_sync if self._side_on else None
        if self._side_on:
            self._side_main = torch.cuda.current_stream()
            if getattr(self, "_wg_stream", None) is None:
                self._wg_stream = torch.cuda.Stream(device=self._side_main.device)

    def _side_sync(self):
        """After a reduction flush the slab pool is recycled: whichever stream ran the flush, the other one must not write a
        recycled slab before the flush has read it."""
        if getattr(self, "_side_on", False):
            self._side_main.wait_stream(self._wg_stream)
            self._wg_stream.wait_stream(self._side_main)

    def _side(self, fn):
        if not getattr(self, "_side_on", False):
            fn()
            return
        ev = torch.cuda.Event()
        ev.record(self._side_main)
        self._wg_stream.wait_event(ev)
        with torch.cuda.stream(self._wg_stream):
            fn()

    def _side_join(self):
        if getattr(self, "_side_on", False):
            self._side_main.wait_stream(self._wg_stream)

    def _sx(self, site):
        """Suffix that makes a scratch tensor private to one site while the second stream is in use."""
        return f":{site}" if getattr(self, "_side_on", False) else ""

    def _ln_bwd(self, dy, x, pfx, dx, tag, dx_add=None, drop_site=None, want16=False, dy_add=None):
        """Backward of _ln_fwd w.r.t. x (+ dx_add); the incoming gradient is dy (+ dy_add).  drop_site: also return dropout_bwd(dx)
        -- the gradient through the Dropout in front of the residual add that feeds this LayerNorm -- as a second output of the
        same launch.  want16: that second output as bf16 (it only feeds GEMM operands of the bf16-compute FFN backward)."""
        dp, _ = self._drop
        dx_drop = dx_drop16 = None
        if drop_site is not None:
            if want16:
                dx_drop16 = self.buf(f"scratch:drop16:{dx.shape[0]}{self._sx(tag)}", dx.shape, torch.bfloat16)
            elif dp > 0.0:
                dx_drop = self.buf(f"scratch:drop:{dx.shape[0]}{self._sx(tag)}", dx.shape)
        hip.layernorm_bwd(dy, x, self.P.views[f"{pfx}/gamma"], self._bufs[f"{tag}:mean"], self._bufs[f"{tag}:rstd"], dx,
                          self.P.gviews[f"{pfx}/gamma"], self.P.gviews[f"{pfx}/beta"], dx_add=dx_add, dx_drop=dx_drop,
                          dropout_p=dp, dropout_site=(drop_site or 0), dropout_step=self._seed_dev, dx_drop16=dx_drop16,
                          dy_add=dy_add)
        if dx_drop16 is not None:
            return dx_drop16
        return dx if dx_drop is None else dx_drop

    def _add_bcast(self, x, p, out):
        hip.call("detr_hip_add_bcast_f32", x.data_ptr(), p.data_ptr(), out.data_ptr(), x.numel(), p.numel())

    # ---- attention --------------------------------------------------------------------------------
    # MultiHeadAttention.call transformer.py:285-356.  The three projections write column blocks of ONE packed buffer
    # QKV [rows, 768] = [Q | K | V]; q and k share their input (src + pos / tgt + query_pos), so Q and K are ONE GEMM
    # with N = 512 against rows 0..511 of in_proj_kernel.  The query scaling head_dim**-0.5 (:307) is folded into the
    # attention kernels (detr_attn_desc.scale).  In the backward the packed dQKV buffer turns the three data gradients
    # and the three residual-style adds into ONE GEMM with K = 768:  d_x = dQKV @ in_proj_kernel + d_residual.
    @property
    def attn16(self):
        return self.compute == 1 and ATTN16

    def _dropmask(self, site, B, T, S, which=None):
        """Keep bits of the attention-probability dropout of `site` (hip.attention_dropmask): one buffer per site and step parity,
        filled once per step by _gen_dropmasks() / one step ahead by _pregen_dropmasks()."""
        which = getattr(self, "_mask_set", 0) if which is None else which
        return self.buf(f"dropmask:{which}:{site}", (hip.attention_dropmask_words(B, HEADS, T, S),), torch.int32)

    def _mask_sites(self, L):
        Q = self.Q
        sites = [(16 * i, L, L) for i in range(self.num_enc)]
        for i in range(self.num_dec):
            ds = 16 * (32 + i)
            sites += [(ds, Q, Q), (ds + 2, Q, L)]
        return sites

    def _launch_dropmasks(self, B, L, which, seed_slot):
        dp, _ = self._drop
        main = torch.cuda.current_stream()
        if getattr(self, "_mask_stream", None) is None:
            self._mask_stream = torch.cuda.Stream(device=main.device)
        bufs = [(self._dropmask(site, B, T, S, which), T, S, site) for site, T, S in self._mask_sites(L)]      # (allocated on the main stream)
        self._mask_stream.wait_stream(main)
        with torch.cuda.stream(self._mask_stream):
            hip.attention_dropmask_many(bufs, B, HEADS, dropout_p=dp, dropout_step=self._seed_dev[seed_slot:])

    def _gen_dropmasks(self, B, L):
        """The keep-bit buffers of this step's attention sites: already there when the previous step generated them one step ahead
        (_pregen_dropmasks; eager launches only), otherwise generated now on a stream of their own behind the seed write (they depend
        on nothing else) and joined in front of the first encoder attention."""
        dp, _ = self._drop
        if not (self.attn16 and dp > 0.0):
            return
        if os.environ.get("DETR_HIP_DROPMASK_STALE") == "1" and getattr(self, "_mask_once", False):
            return          # timing experiment only (profiles/r06_ab_results.txt): the keep bits of the first step are reused -- what their generation costs the step
        self._mask_once = True
        self._mask_set = self._step_no & 1
        tags = self.__dict__.setdefault("_mask_tags", {})
        key = (self._step_seed, B, L, self.Q, dp)
        if not (tags.get(self._mask_set) == key and not self._graph_replay):
            self._launch_dropmasks(B, L, self._mask_set, 0)
            tags[self._mask_set] = None if self._graph_replay else key
        self._mask_pending = True

    def _pregen_dropmasks(self, B, L, now=False):
        """Called when the decoder forward begins (800-row kernels, then the matcher: most CUs idle for ~1.5 ms): the NEXT step's keep
        bits, from its seed in slot 1 of the device seed block, into the other buffer set.  Eager launches only (a recorded graph keeps
        the generation at the head of its own step).  Off by default: measured +0.07 ms per step (DETR_HIP_DROPMASK_AHEAD=1 enables it; profiles/r06_ab_results.txt #3)."""
        dp, _ = self._drop
        if not (self.attn16 and dp > 0.0) or self._graph_replay or not DROPMASK_AHEAD or os.environ.get("DETR_HIP_DROPMASK_STALE") == "1":
            return
        if DROPMASK_AHEAD == 2 and not now:
            self._pregen_args = (B, L)          # launched by the set loss between its cost and assignment kernels
            return
        nxt = (self._step_no + 1) & 1
        self._launch_dropmasks(B, L, nxt, 1)
        self._mask_tags[nxt] = (step_seed(self.dropout_seed, self._step_no + 1, self.dp_rank), B, L, self.Q, dp)

    def pregen_dropmasks_at_matcher(self):
        """DETR_HIP_DROPMASK_AHEAD=2: the next step's keep bits next to the assignment kernel alone (one wave per problem, ~50 CUs for
        ~0.7 ms), not next to the decoder's chain of short launches (which the co-running generator slowed: profiles/r06_ab_results.txt #3)."""
        args, self._pregen_args = getattr(self, "_pregen_args", None), None
        if args is not None:
            self._pregen_dropmasks(*args, now=True)

    def _join_dropmasks(self):
        if getattr(self, "_mask_pending", False):
            torch.cuda.current_stream().wait_stream(self._mask_stream)
            self._mask_pending = False

    def _self_attn_fwd(self, tag, pfx, qk_in, v_in, B, T, out, site, ln=None):
        W, bias = self._w(f"{pfx}/in_proj_kernel"), self.P.views[f"{pfx}/in_proj_bias"]
        dp, _ = self._drop
        lse = self.buf(f"{tag}:lse", (B * HEADS, T))
        if self.attn16:
            # bf16 operands: three members of one grouped launch (Q alone carries the alpha), bf16 O straight into the out-projection
            QKV = self.buf(f"{tag}:QKV16", (B * T, 3 * D), torch.bfloat16)
            hip.gemm_group([hip.linear_fwd_call(qk_in, W[0:D], bias[0:D], QKV[:, 0:D], alpha=QK_ALPHA),          # :294-300, :307
                            hip.linear_fwd_call(qk_in, W[D:2 * D], bias[D:2 * D], QKV[:, D:2 * D]),
                            hip.linear_fwd_call(v_in, W[2 * D:], bias[2 * D:], QKV[:, 2 * D:])])                  # :302-304
            O = self.buf(f"{tag}:O16", (B * T, D), torch.bfloat16)
            self._join_dropmasks()
            hip.attention(QKV[:, 0:D], QKV[:, D:2 * D], QKV[:, 2 * D:], O, lse, B, HEADS, T, T, scale=float(HD) ** -0.5, dropout_p=dp,
                          dropout_site=site, dropout_step=self._seed_dev, dropmask=self._dropmask(site, B, T, T) if dp > 0.0 else None)
        else:
            QKV = self.buf(f"{tag}:QKV", (B * T, 3 * D))
            hip.gemm_group([hip.linear_fwd_call(qk_in, W[0:2 * D], bias[0:2 * D], QKV[:, 0:2 * D]),        # :294-300
                            hip.linear_fwd_call(v_in, W[2 * D:], bias[2 * D:], QKV[:, 2 * D:])])          # :302-304
            O = self.buf(f"{tag}:O", (B * T, D))
            hip.attention(QKV[:, 0:D], QKV[:, D:2 * D], QKV[:, 2 * D:], O, lse, B, HEADS, T, T, scale=float(HD) ** -0.5,
                          dropout_p=dp, dropout_site=site, dropout_step=self._seed_dev)                    # :307-345
        hip.linear_fwd(O, self._w(f"{pfx}/out_proj_kernel"), self.P.views[f"{pfx}/out_proj_bias"], out, residual=v_in,
                       dropout_p=dp, dropout_seed=site + 1, dropout_step=self._seed_dev, ln=ln)        # :346-347 + :169 (+ the LayerNorm behind it)

    def _self_attn_bwd(self, tag, pfx, d_out, d_res, qk_in, v_in, B, T, d_x, site, acc_qk=None):
        """d_out: gradient of the out-projection output (dropout backward already applied), d_res: gradient of the
        block output (residual path).  Writes d_x = gradient w.r.t. v_in, including the q / k branch when qk_in is
        v_in + const (d_x = None: not needed).  acc_qk: buffer that ACCUMULATES the gradient of the q / k input alone
        (the decoder's query_pos gradient)."""
        G = self.P.gviews
        W, gW, gb = self._w(f"{pfx}/in_proj_kernel"), G[f"{pfx}/in_proj_kernel"], G[f"{pfx}/in_proj_bias"]
        a16 = self.attn16
        adt = torch.bfloat16 if a16 else torch.float32
        QKV, O = self._bufs[f"{tag}:QKV16" if a16 else f"{tag}:QKV"], self._bufs[f"{tag}:O16" if a16 else f"{tag}:O"]
        dO = self.buf(f"scratch:dO:{B * T}:{int(a16)}", (B * T, D), adt)
        hip.linear_dgrad(d_out, self._w(f"{pfx}/out_proj_kernel"), dO)
        dQKV = self.buf(f"scratch:dQKV:{B * T}:{int(a16)}{self._sx(tag)}", (B * T, 3 * D), adt)
        delta = self.buf(f"scratch:delta:{B * T}", (2 * B * HEADS, T))       # (bf16 operands: delta / scale | lse * log2 e - log2 scale)
        dp, _ = self._drop
        hip.attention(QKV[:, 0:D], QKV[:, D:2 * D], QKV[:, 2 * D:], O, self._bufs[f"{tag}:lse"], B, HEADS, T, T,
                      scale=float(HD) ** -0.5, dropout_p=dp, dropout_site=site, dropout_step=self._seed_dev,
                      d_o=dO, dq=dQKV[:, 0:D], dk=dQKV[:, D:2 * D], dv=dQKV[:, 2 * D:], delta=delta,
                      dropmask=self._dropmask(site, B, T, T) if (a16 and dp > 0.0) else None)
        # weight gradients (bias gradients fused: row sums of dy^T), one grouped launch (bf16 operands: the out-projection's pair has
        # other storage types than the in-projections' and is a launch of its own)
        wg_out = hip.linear_wgrad_call(d_out, O, G[f"{pfx}/out_proj_kernel"], bias_grad=G[f"{pfx}/out_proj_bias"])
        wg_in = [hip.linear_wgrad_call(dQKV[:, 0:2 * D], qk_in, gW[0:2 * D], bias_grad=gb[0:2 * D]),
                 hip.linear_wgrad_call(dQKV[:, 2 * D:], v_in, gW[2 * D:], bias_grad=gb[2 * D:])]
        if a16:
            self._side(lambda: (hip.gemm(*wg_out[0], **wg_out[1]), hip.gemm_group(wg_in)))
        else:
            self._side(lambda: hip.gemm_group([wg_out] + wg_in))
        calls = []
        if acc_qk is not None:
            calls.append(hip.linear_dgrad_call(dQKV[:, 0:2 * D], W[0:2 * D], acc_qk, residual=acc_qk))
        if d_x is not None:
            calls.append(hip.linear_dgrad_call(dQKV, W, d_x, residual=d_res))
        if len(calls) == 1:
            hip.gemm(*calls[0][0], **calls[0][1])
        elif calls:
            hip.gemm_group(calls)

    # bf16 compute mode: the tensors that ONLY feed GEMM operands of the FFN -- its input twin x16 (written by the LayerNorm
    # that produces x), the hidden activation h, its gradient dh and the incoming gradient twin d_y16 (written by the
    # LayerNorm backward) -- are STORED in bf16: the loaders would round them anyway (bit-identical products), the two
    # [rows, 2048] tensors move half the bytes, and both weight gradients become one grouped launch of the all-bf16 variant.
    @property
    def ffn16(self):
        return self.compute == 1 and H16

    def _ffn_fwd(self, tag, pfx, x, out_pre_ln, seed=0, x16=None, ln=None):
        V = self.P.views
        dp, _ = self._drop
        h = self.buf(f"{tag}:h", (x.shape[0], FF), torch.bfloat16 if self.ffn16 else torch.float32)
        hip.linear_fwd(x16 if x16 is not None else x, self._w(f"{pfx}/linear1/kernel"), V[f"{pfx}/linear1/bias"], h, act=1,
                       dropout_p=dp, dropout_seed=seed, dropout_step=self._seed_dev)    # :172-174
        hip.linear_fwd(h, self._w(f"{pfx}/linear2/kernel"), V[f"{pfx}/linear2/bias"], out_pre_ln, residual=x, dropout_p=dp,
                       dropout_seed=seed + 1, dropout_step=self._seed_dev, ln=ln)      # :175-176 (+ the LayerNorm behind it)

    def _ffn_bwd(self, tag, pfx, d_y, d_f, x, dx, x16=None):
        """d_f: grad of (drop(linear2(drop(relu(linear1(x))))) + x); d_y = dropout_bwd(d_f) (from the LayerNorm backward
        launch; bf16 in bf16 compute mode); dx = full gradient w.r.t. x."""
        G = self.P.gviews
        h = self._bufs[f"{tag}:h"]                 # post-ReLU, post-dropout hidden activation
        dp, _ = self._drop
        dh = self.buf(f"scratch:dh:{h.shape[0]}:{int(self.ffn16)}{self._sx(tag)}", h.shape,
                      torch.bfloat16 if self.ffn16 else torch.float32)
        # (h > 0) is both the ReLU and the keep mask of the hidden dropout; its 1/(1-p) scale goes in alpha
        hip.linear_dgrad(d_y, self._w(f"{pfx}/linear2/kernel"), dh, mask=h, alpha=(1.0 / (1.0 - dp)) if dp > 0.0 else 1.0)
        self._side(lambda: hip.gemm_group([
            hip.linear_wgrad_call(d_y, h, G[f"{pfx}/linear2/kernel"], bias_grad=G[f"{pfx}/linear2/bias"]),
            hip.linear_wgrad_call(dh, x16 if x16 is not None else x, G[f"{pfx}/linear1/kernel"],
                                  bias_grad=G[f"{pfx}/linear1/bias"])]))
        hip.linear_dgrad(dh, self._w(f"{pfx}/linear1/kernel"), dx, residual=d_f)

    # ---- layer-invariant decoder cross-attention K / V (transformer.py:221-223: memory is the same for every layer) ------
    def _cross_tables(self):
        """Wkv [2*nd*256, 256] = [Wk_0 .. Wk_nd-1 ; Wv_0 .. Wv_nd-1] (rows 256..767 of every multihead_attn in_proj_kernel) and
        its bias: a derived weight copy like the BN-folded conv kernels, gathered by ONE multi-copy launch per weights
        version; the gradient of the copy is scattered back by one more."""
        nd = self.num_dec
        key = ("cross", self.compute)
        if key not in self._cross:
            wdt = torch.bfloat16 if self.compute == 1 else torch.float32
            Wkv = self.buf(f"cross:Wkv:{self.compute}", (2 * nd * D, D), wdt)
            bkv = self.buf("cross:bkv", (2 * nd * D,))
            gW = self.buf("cross:gWkv", (2 * nd * D, D))
            gb = self.buf("cross:gbkv", (2 * nd * D,))
            gather, scatter = [], []
            src_views = self.P.views16 if self.compute == 1 else self.P.views
            for i in range(nd):
                pfx = f"transformer/decoder/layer_{i}/multihead_attn"
                w, b = src_views[f"{pfx}/in_proj_kernel"], self.P.views[f"{pfx}/in_proj_bias"]
                g, gbias = self.P.gviews[f"{pfx}/in_proj_kernel"], self.P.gviews[f"{pfx}/in_proj_bias"]
                for part, row0 in ((0, i * D), (1, (nd + i) * D)):             # part 0 = K rows, 1 = V rows
                    lo = (1 + part) * D
                    gather.append((w[lo:lo + D], Wkv[row0:row0 + D], 0))
                    gather.append((b[lo:lo + D], bkv[row0:row0 + D], 0))
                    scatter.append((gW[row0:row0 + D], g[lo:lo + D], 1))
                    scatter.append((gb[row0:row0 + D], gbias[lo:lo + D], 1))
            self._cross[key] = dict(Wkv=Wkv, bkv=bkv, gW=gW, gb=gb, gather=hip.copy_table(gather, self.device),
                                    scatter=hip.copy_table(scatter, self.device))
        return self._cross[key]

    # ---- forward ----------------------------------------------------------------------------------
    def forward(self, images, training=False):
        """See _forward_impl.  GEMM / conv compute mode: self.compute (0 = exact fp32, 1 = bf16 MFMA)."""
        hip.COMPUTE_BF16 = self.gemm_mode
        if self.compute == 1 and (self._stale("shadow16") or self.P.views16 is None):
            self._refresh_shadow()
        if self.tf_backbone and self._stale("bias_shift"):
            self._refresh_bias_shift()
        self._side_begin(WGRAD_STREAM and images.is_cuda)
        try:
            return self._forward_impl(images, training)
        finally:
            try:
                self._side_join()
            finally:
                self._side_on = False
                hip.COMPUTE_BF16 = 0

    def backward(self, d_logits, d_boxes, backbone=True, on_bucket=None):
        hip.COMPUTE_BF16 = self.gemm_mode
        cb = on_bucket
        if DEFER_REDUCE:
            # the weight gradients are only read by the bucket exchange / the optimiser: their split-K reductions are queued
            # and run 16 per launch (hip.flush_reduces) instead of one ~9 us latency-bound launch behind every GEMM
            hip.begin_deferred_reduces(self.device)
            if on_bucket:
                def cb(i):
                    self._side_join()
                    hip.flush_reduces()
                    on_bucket(i)
        elif on_bucket:
            def cb(i):
                self._side_join()
                on_bucket(i)
        # (immediate split-K reductions share ONE slab workspace between all launches: single stream then)
        self._side_begin(WGRAD_STREAM and DEFER_REDUCE and d_logits.is_cuda)
        try:
            return self._backward_impl(d_logits, d_boxes, backbone, cb)
        finally:
            try:
                self._side_join()
                self._side_on = False
                hip.flush_reduces(end=True)
            finally:
                hip.COMPUTE_BF16 = 0

    def _forward_impl(self, images, training=False):
        """images: CUDA fp32 NHWC [B,H,W,3] (already normalised, processing.py:12-16).
        Returns (logits [Lv,B,Q,C], boxes [Lv,B,Q,4]) views of engine buffers."""
        assert images.is_cuda and images.dtype == torch.float32 and images.dim() == 4 and images.shape[3] == 3
        images = images.contiguous()
        if training and self.dropout_p > 0.0:
            if not self._graph_replay:          # a captured step is fed its seed by training.GraphedTrainStep before every replay
                self.advance_dropout_step()
            self._drop = (float(self.dropout_p), self._step_seed)
        else:
            self._drop = (0.0, 0)
        dp, dseed = self._drop
        B, H, W, _ = images.shape
        self._shape = (B, H, W)
        if dp > 0.0:
            hf, wf = H, W
            for _ in range(5):                                   # stem conv, max-pool, layer2..4: each ceil(n / 2)
                hf, wf = (hf - 1) // 2 + 1, (wf - 1) // 2 + 1
            self._gen_dropmasks(B, hf * wf)
        self.images = images
        V = self.P.views
        self.phase("fwd backbone")
        # ---------------- stem (resnet_backbone.py:11-26) ----------------
        H1, W1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        M1 = B * H1 * W1
        ws = self._scaled_kernel(f"{self._stem['conv']}/kernel", self._stem["bn"])
        adt = torch.bfloat16 if (self.compute == 1 and ACT16) else torch.float32      # storage type of backbone activations
        self._adt = adt
        stem = self.buf("stem:out", (B, H1, W1, 64), adt)
        # implicit GEMM: the 7x7x3 patches are gathered from the image by the A loader (stem_conv.hip), no im2col buffer
        hip.stem_conv(0, images, ws, stem, B, H, W, H1, W1, bias=self.bn_shift[self._stem["bn"]], act=1)
        H2, W2 = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        pool = self.buf("stem:pool", (B, H2, W2, 64), adt)
        amax = self.buf("stem:amax", (B, H2, W2, 64), torch.uint8)
        hip.call("detr_hip_maxpool3x3s2_fwd_bf16" if adt == torch.bfloat16 else "detr_hip_maxpool3x3s2_fwd_f32", stem.data_ptr(),
                 pool.data_ptr(), amax.data_ptr(), B, H1, W1, 64, H2, W2)
        # ---------------- residual stages (resnet_backbone.py:116-137) ----------------
        x, h, w, cin = pool, H2, W2, 64
        self._block_meta = []
        for li, nb in enumerate(self.blocks):
            d1, d2 = 64 * 2 ** li, 256 * 2 ** li
            for b in range(nb):
                n = block_names(li, b, self.tf_backbone)
                p = n["tag"]
                stride = 2 if (b == 0 and li > 0) else 1
                ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
                M_in, M_out = B * h * w, B * ho * wo
                xs = x
                if stride == 2 and (b == 0):
                    # stride-2 1x1 convs read every second pixel: gather them once (a plain 16-byte-chunk copy: bf16 tensors
                    # pass as cin/2 "float" channels)
                    xs = self.buf(f"{p}:xs", (B, ho, wo, cin), adt)
                    hip.call("detr_hip_subsample2_fwd_f32", x.data_ptr(), xs.data_ptr(), B, h, w,
                             cin // 2 if adt == torch.bfloat16 else cin, ho, wo)
                if self.tf_backbone:            # ResNet v1 (keras.applications): the stride sits on the FIRST 1x1 conv
                    x1, M1, h1, w1, s2 = xs, M_out, ho, wo, 1
                else:                           # reference backbone (resnet_backbone.py:104-105): on the 3x3 conv
                    x1, M1, h1, w1, s2 = x, M_in, h, w, stride
                if b == 0:
                    # the projection shortcut only meets the main branch at conv3's residual operand: second stream
                    # (an HBM-bound 1x1 GEMM next to the MFMA-bound 3x3 conv)
                    idn = self.buf(f"{p}:idn", (B, ho, wo, d2), adt)
                    self._side(lambda: self._conv1x1_fwd(xs, M_out, cin, d2, f"{n['down']}/kernel", n["bnd"], idn, act=0))
                else:
                    idn = x
                y1 = self.buf(f"{p}:y1", (B, h1, w1, d1), adt)
                self._conv1x1_fwd(x1, M1, cin, d1, f"{n['conv1']}/kernel", n["bn1"], y1)
                y2 = self.buf(f"{p}:y2", (B, ho, wo, d1), adt)
                hip.conv3x3(0, y1, self._scaled_kernel(f"{n['conv2']}/kernel", n["bn2"]), y2, B, h1, w1, d1, ho, wo, d1,
                            s2, bias=self.bn_shift[n["bn2"]], act=1)
                if b == 0:
                    self._side_join()
                out = self.buf(f"{p}:out", (B, ho, wo, d2), adt)
                # training, bf16 activation storage: the block output's ReLU mask as BITS (1 byte per 8 channels) next to the tensor --
                # the backward only needs the sign, and the GEMMs that apply it are HBM streams (1/16 of the mask bytes)
                # (only where the consumer is the STREAMING kernel -- the conv1 input gradient of a following block of layer1-3: there the
                #  mask stream is a third of the bytes; in the tile engine's / the 3x3 kernels' epilogues the byte loads measured 2-9 %
                #  SLOWER than the 16-byte mask loads, so y1 / y2 and the layer4 blocks keep their bf16 masks)
                nxt_d1 = 64 * 2 ** (li if b + 1 < nb else li + 1)        # d1 of the next block
                last = (li + 1 == len(self.blocks)) and (b + 1 == nb)
                obits = self.buf(f"{p}:out_bits", (M_out, d2 // 8), torch.uint8) \
                    if (training and MASK_BITS and adt == torch.bfloat16 and not last and nxt_d1 <= 256) else None
                self._conv1x1_fwd(y2, M_out, d1, d2, f"{n['conv3']}/kernel", n["bn3"], out, residual=idn, maskbits_out=obits)
                self._block_meta.append(dict(p=p, n=n, x=x, xs=xs, x1=x1, y1=y1, y2=y2, out=out, h=h, w=w, ho=ho, wo=wo, h1=h1, w1=w1,
                                             M1=M1, s2=s2, cin=cin, d1=d1, d2=d2, stride=stride, first=(b == 0), out_bits=obits))
                x, h, w, cin = out, ho, wo, d2
        feat, Hf, Wf = x, h, w
        L = Hf * Wf
        self._feat_meta = (feat, Hf, Wf, L)
        self.phase("fwd encoder")
        # ---------------- input_proj + positional encoding (detr.py:172-175) ----------------
        src = self.buf("enc:src0", (B * L, D))
        hip.gemm(B * L, D, 2048, feat, 2048, 1, self._w("input_proj/kernel"), D, 0, src, D, bias=V["input_proj/bias"])
        key = (Hf, Wf)
        if key not in self._pos_cache:
            self._pos_cache[key] = torch.from_numpy(position_embedding_sine_host(Hf, Wf)).to(self.device)
        pos = self._pos_cache[key]
        self.pos = pos
        # ---------------- encoder (transformer.py:157-179) ----------------
        x = src
        qk = self.buf("enc0:qk", (B * L, D))
        self._add_bcast(x, pos, qk)                              # src + pos (:161-163); later layers get it from LayerNorm 2
        for i in range(self.num_enc):
            pfx, tag = f"transformer/encoder/layer_{i}", f"enc{i}"
            a = self.buf(f"{tag}:a", (B * L, D))
            x1 = self.buf(f"{tag}:x1", (B * L, D))
            x1h = self.buf(f"{tag}:x1h", (B * L, D), torch.bfloat16) if self.ffn16 else None
            ln1 = self._ln_spec(B * L, f"{pfx}/norm1", x1, f"{tag}:ln1", y16=x1h)
            self._self_attn_fwd(f"{tag}:sa", f"{pfx}/self_attn", qk, x, B, L, a, site=16 * i, ln=ln1)
            if ln1 is None:
                self._ln_fwd(a, f"{pfx}/norm1", x1, f"{tag}:ln1", y16=x1h)
            f = self.buf(f"{tag}:f", (B * L, D))
            x2 = self.buf(f"{tag}:x2", (B * L, D))
            # x2 + pos = the q / k input of the next layer, or `memory + pos` of the decoder (:219), in the same launch
            qk = self.buf(f"enc{i + 1}:qk" if i + 1 < self.num_enc else "dec:mem_pos", (B * L, D))
            ln2 = self._ln_spec(B * L, f"{pfx}/norm2", x2, f"{tag}:ln2", add=pos, y2=qk, ffn=True)
            self._ffn_fwd(tag, pfx, x1, f, seed=16 * i + 2, x16=x1h, ln=ln2)
            if ln2 is None:
                self._ln_fwd(f, f"{pfx}/norm2", x2, f"{tag}:ln2", add=pos, y2=qk)
            x = x2
        memory, mem_pos = x, qk
        if self.num_enc == 0:
            mem_pos = qk                                          # = src + pos
        self.phase("fwd decoder")
        if training:
            self._pregen_dropmasks(B, L)
        # ---------------- decoder (transformer.py:207-234, :104-133) ----------------
        Q, nd = self.Q, self.num_dec
        qpos = V["query_embed/kernel"]
        tgt = self.buf("dec:tgt0", (B * Q, D))
        hip.zero_(tgt)                                           # transformer.py:45
        hs = self.buf("dec:hs", (nd, B * Q, D))
        t3all = self.buf("dec:t3all", (nd, B * Q, D))           # the layer outputs, contiguous: ONE final-norm launch over all levels
        dp, _ = self._drop
        # K / V projections of `memory` for ALL decoder layers at once: memory is layer-invariant (:221-223)
        ct = self._cross_tables()
        if self._stale(f"cross:{self.compute}"):
            hip.multi_copy(ct["gather"])
        a16 = self.attn16
        KV = self.buf("dec:KV16" if a16 else "dec:KV", (B * L, 2 * nd * D), torch.bfloat16 if a16 else torch.float32)
        # (second stream: the first decoder self-attention block -- 800-row kernels -- does not need it)
        self._side(lambda: hip.gemm_group([hip.linear_fwd_call(mem_pos, ct["Wkv"][0:nd * D], ct["bkv"][0:nd * D], KV[:, 0:nd * D]),
                                           hip.linear_fwd_call(memory, ct["Wkv"][nd * D:], ct["bkv"][nd * D:], KV[:, nd * D:])]))
        qin = self.buf("dec0:qin", (B * Q, D))
        self._add_bcast(tgt, qpos, qin)                          # tgt + query_pos (:209); later layers: from LayerNorm 3
        for i in range(nd):
            pfx, tag = f"transformer/decoder/layer_{i}", f"dec{i}"
            ds = 16 * (32 + i)
            a1 = self.buf(f"{tag}:a1", (B * Q, D))
            t1 = self.buf(f"{tag}:t1", (B * Q, D))
            q2 = self.buf(f"{tag}:q2", (B * Q, D))
            ln1 = self._ln_spec(B * Q, f"{pfx}/norm1", t1, f"{tag}:ln1", add=qpos, y2=q2)
            self._self_attn_fwd(f"{tag}:sa", f"{pfx}/self_attn", qin, tgt, B, Q, a1, site=ds, ln=ln1)
            if ln1 is None:
                self._ln_fwd(a1, f"{pfx}/norm1", t1, f"{tag}:ln1", add=qpos, y2=q2)      # q2 = t1 + query_pos (:219)
            # cross attention: Q from this layer, K / V column blocks of the shared projection buffer
            cp = f"{pfx}/multihead_attn"
            Wc, bc = self._w(f"{cp}/in_proj_kernel"), V[f"{cp}/in_proj_bias"]
            Qc = self.buf(f"{tag}:ca:Q16" if a16 else f"{tag}:ca:Q", (B * Q, D), torch.bfloat16 if a16 else torch.float32)
            hip.linear_fwd(q2, Wc[0:D], bc[0:D], Qc, alpha=QK_ALPHA if a16 else 1.0)
            if i == 0:
                self._side_join()                                # K / V of all layers
            Oc = self.buf(f"{tag}:ca:O16" if a16 else f"{tag}:ca:O", (B * Q, D), torch.bfloat16 if a16 else torch.float32)
            lse = self.buf(f"{tag}:ca:lse", (B * HEADS, Q))
            hip.attention(Qc, KV[:, i * D:(i + 1) * D], KV[:, (nd + i) * D:(nd + i + 1) * D], Oc, lse, B, HEADS, Q, L,
                          scale=float(HD) ** -0.5, dropout_p=dp, dropout_site=ds + 2, dropout_step=self._seed_dev,
                          dropmask=self._dropmask(ds + 2, B, Q, L) if (a16 and dp > 0.0) else None)
            a2 = self.buf(f"{tag}:a2", (B * Q, D))
            t2 = self.buf(f"{tag}:t2", (B * Q, D))
            t2h = self.buf(f"{tag}:t2h", (B * Q, D), torch.bfloat16) if self.ffn16 else None
            ln2 = self._ln_spec(B * Q, f"{pfx}/norm2", t2, f"{tag}:ln2", y16=t2h)
            hip.linear_fwd(Oc, self._w(f"{cp}/out_proj_kernel"), V[f"{cp}/out_proj_bias"], a2, residual=t1, dropout_p=dp,
                           dropout_seed=ds + 3, dropout_step=self._seed_dev, ln=ln2)      # :226 (+ norm2)
            if ln2 is None:
                self._ln_fwd(a2, f"{pfx}/norm2", t2, f"{tag}:ln2", y16=t2h)
            f = self.buf(f"{tag}:f", (B * Q, D))
            t3 = self._bufs[f"{tag}:t3"] = t3all[i]
            qin = self.buf(f"dec{i + 1}:qin", (B * Q, D)) if i + 1 < nd else None        # next layer's tgt + query_pos
            ln3 = self._ln_spec(B * Q, f"{pfx}/norm3", t3, f"{tag}:ln3", add=qpos if qin is not None else None, y2=qin, ffn=True)
            self._ffn_fwd(tag, pfx, t2, f, seed=ds + 4, x16=t2h, ln=ln3)
            if ln3 is None:
                self._ln_fwd(f, f"{pfx}/norm3", t3, f"{tag}:ln3", add=qpos if qin is not None else None, y2=qin)
            tgt = t3
        # the shared decoder norm of every level (:121-125) in one launch (round 4: it was one 800-row launch per layer, forward and backward)
        self._ln_fwd(t3all.view(nd * B * Q, D), "transformer/decoder/norm", hs.view(nd * B * Q, D), "dec:lnf")
        self.phase("fwd heads")
        # ---------------- heads (detr.py:181-204 / :94-114) ----------------
        hip.COMPUTE_BF16 = 0          # the heads, like LayerNorm / softmax / the set loss, always run in exact fp32
        Lv = self.num_dec
        R = Lv * B * Q
        hs2 = hs.view(R, D)
        logits = self.buf("head:logits", (R, self.C))
        boxes = self.buf("head:boxes", (R, 4))
        t_a, t_b = self.buf("head:t1", (R, D)), self.buf("head:t2", (R, D))
        if self.nb_class is None:
            hip.linear_fwd(hs2, V["class_embed/kernel"], V["class_embed/bias"], logits)
            hip.linear_fwd(hs2, V["bbox_embed_0/kernel"], V["bbox_embed_0/bias"], t_a, act=1)
            hip.linear_fwd(t_a, V["bbox_embed_1/kernel"], V["bbox_embed_1/bias"], t_b, act=1)
            hip.linear_fwd(t_b, V["bbox_embed_2/kernel"], V["bbox_embed_2/bias"], boxes, act=2)
        else:   # Keras Dense kernels are (in, out)
            def dense(x, name, out, act):
                k = V[f"{name}/kernel"]
                hip.gemm(x.shape[0], k.shape[1], k.shape[0], x, x.stride(0), 1, k, k.shape[1], 0, out, out.stride(0),
                         bias=V[f"{name}/bias"], act=act)
            dense(hs2, "cls_layer", logits, 0)
            dense(hs2, "pos_layer/dense_0", t_a, 1)
            dense(t_a, "pos_layer/dense_1", t_b, 1)
            dense(t_b, "pos_layer/dense_2", boxes, 2)
        self.phase("set loss")
        return logits.view(Lv, B, Q, self.C), boxes.view(Lv, B, Q, 4)

    # ---- backward ---------------------------------------------------------------------------------
    def zero_grad(self):
        hip.zero_(self.P.grad)

    def _backward_impl(self, d_logits, d_boxes, backbone=True, on_bucket=None):
        """d_logits [Lv,B,Q,C], d_boxes [Lv,B,Q,4] (contiguous CUDA fp32): gradients of the scalar
        loss w.r.t. the head outputs.  Parameter gradients are ACCUMULATED into P.grad.
        on_bucket(i) is called when gradient bucket i (ParamStore.bucket_bounds) is final."""
        B, H, W = self._shape
        V, G = self.P.views, self.P.gviews
        Q, Lv = self.Q, self.num_dec
        R = Lv * B * Q
        hs = self._bufs["dec:hs"]
        hs2 = hs.view(R, D)
        dl = d_logits.reshape(R, self.C)
        db = d_boxes.reshape(R, 4)
        boxes, t_a, t_b = self._bufs["head:boxes"], self._bufs["head:t1"], self._bufs["head:t2"]
        self.phase("bwd heads")
        # ---------------- heads ----------------
        hip.COMPUTE_BF16 = 0
        dz3 = self.buf("scratch:dz3", (R, 4))
        hip.call("detr_hip_sigmoid_bwd_f32", db.data_ptr(), boxes.data_ptr(), dz3.data_ptr(), R * 4)
        d_hs = self.buf("scratch:d_hs", (R, D))
        dt_b, dt_a = self.buf("scratch:dt_b", (R, D)), self.buf("scratch:dt_a", (R, D))
        if self.nb_class is None:
            hip.linear_wgrad(dz3, t_b, G["bbox_embed_2/kernel"], bias_grad=G["bbox_embed_2/bias"])
            hip.linear_dgrad(dz3, V["bbox_embed_2/kernel"], dt_b, mask=t_b)
            hip.linear_wgrad(dt_b, t_a, G["bbox_embed_1/kernel"], bias_grad=G["bbox_embed_1/bias"])
            hip.linear_dgrad(dt_b, V["bbox_embed_1/kernel"], dt_a, mask=t_a)
            hip.linear_wgrad(dt_a, hs2, G["bbox_embed_0/kernel"], bias_grad=G["bbox_embed_0/bias"])
            hip.linear_dgrad(dt_a, V["bbox_embed_0/kernel"], d_hs)
            hip.linear_wgrad(dl, hs2, G["class_embed/kernel"], bias_grad=G["class_embed/bias"])
            hip.linear_dgrad(dl, V["class_embed/kernel"], d_hs, residual=d_hs)
        else:
            def dense_bwd(dy, x, name, dx, mask=None, residual=None):
                k, gk = V[f"{name}/kernel"], G[f"{name}/kernel"]      # (in, out)
                n_in, n_out = k.shape
                self._wgrad(n_in, n_out, x.shape[0], x, x.stride(0), dy, dy.stride(0), gk, n_out)
                self._colsum(dy, G[f"{name}/bias"])
                hip.gemm(dy.shape[0], n_in, n_out, dy, dy.stride(0), 1, k, n_out, 1, dx, dx.stride(0), residual=residual,
                         ldr=(dx.stride(0) if residual is not None else 0), mask=mask,
                         ldmask=(mask.stride(0) if mask is not None else 0))
            dense_bwd(dz3, t_b, "pos_layer/dense_2", dt_b, mask=t_b)
            dense_bwd(dt_b, t_a, "pos_layer/dense_1", dt_a, mask=t_a)
            dense_bwd(dt_a, hs2, "pos_layer/dense_0", d_hs)
            dense_bwd(dl, hs2, "cls_layer", d_hs, residual=d_hs)
        d_hs3 = d_hs.view(Lv, B * Q, D)
        hip.COMPUTE_BF16 = self.gemm_mode
        self.phase("bwd decoder")
        # ---------------- decoder ----------------
        feat, Hf, Wf, L = self._feat_meta
        nd = self.num_dec
        memory = self._bufs[f"enc{self.num_enc - 1}:x2"] if self.num_enc > 0 else self._bufs["enc:src0"]
        mem_pos = self._bufs["dec:mem_pos"] if self.num_enc > 0 else self._bufs["enc0:qk"]
        a16 = self.attn16
        adt16 = torch.bfloat16 if a16 else torch.float32
        KV = self._bufs["dec:KV16" if a16 else "dec:KV"]
        ct = self._cross_tables()
        qpos, g_qpos = V["query_embed/kernel"], G["query_embed/kernel"]
        BQ = B * Q
        dp, _ = self._drop
        dKV = self.buf(f"scratch:dKV:{int(a16)}", (B * L, 2 * nd * D), adt16)      # every column block is written by its layer's attention backward
        acc_qpos = self.buf("scratch:acc_qpos", (BQ, D))             # sum over layers / sites of d(tgt + query_pos): query_pos gradient
        hip.zero_(acc_qpos)
        d_next = None                       # gradient flowing into t3 of layer i from layer i+1
        # the shared decoder norm, all levels at once: d_t3n[i] = gradient of level i's head input w.r.t. t3 of layer i
        d_t3n = self.buf("scratch:d_t3n", (nd, BQ, D))
        self._ln_bwd(d_hs3.view(nd * BQ, D), self._bufs["dec:t3all"].view(nd * BQ, D), "transformer/decoder/norm", d_t3n.view(nd * BQ, D), "dec:lnf")
        for i in reversed(range(nd)):
            pfx, tag = f"transformer/decoder/layer_{i}", f"dec{i}"
            cp = f"{pfx}/multihead_attn"
            t1, t2, t3 = self._bufs[f"{tag}:t1"], self._bufs[f"{tag}:t2"], self._bufs[f"{tag}:t3"]
            a1, a2, f = self._bufs[f"{tag}:a1"], self._bufs[f"{tag}:a2"], self._bufs[f"{tag}:f"]
            qin, q2 = self._bufs[f"{tag}:qin"], self._bufs[f"{tag}:q2"]
            tgt = self._bufs[f"dec{i - 1}:t3"] if i > 0 else self._bufs["dec:tgt0"]
            ds = 16 * (32 + i)
            d_f = self.buf(f"scratch:d_f{self._sx(tag)}", (BQ, D))
            # d_t3 = d_t3n[i] + d_next: the two branches are summed by the norm3 backward itself
            d_y = self._ln_bwd(d_t3n[i], f, f"{pfx}/norm3", d_f, f"{tag}:ln3", drop_site=ds + 5, want16=self.ffn16, dy_add=d_next)
            d_t2 = self.buf("scratch:d_t2", (BQ, D))
            self._ffn_bwd(tag, pfx, d_y, d_f, t2, d_t2, x16=self._bufs.get(f"{tag}:t2h") if self.ffn16 else None)
            d_a2 = self.buf(f"scratch:d_a2{self._sx(tag)}", (BQ, D))
            d_out = self._ln_bwd(d_t2, a2, f"{pfx}/norm2", d_a2, f"{tag}:ln2", drop_site=ds + 3)
            # ---- cross attention
            Wc = self._w(f"{cp}/in_proj_kernel")
            Qc, Oc = self._bufs[f"{tag}:ca:Q16" if a16 else f"{tag}:ca:Q"], self._bufs[f"{tag}:ca:O16" if a16 else f"{tag}:ca:O"]
            dO = self.buf(f"scratch:dO:{BQ}:{int(a16)}", (BQ, D), adt16)
            hip.linear_dgrad(d_out, self._w(f"{cp}/out_proj_kernel"), dO)
            dQc = self.buf(f"scratch:dQc:{int(a16)}{self._sx(tag)}", (BQ, D), adt16)
            delta = self.buf(f"scratch:delta:{BQ}", (2 * B * HEADS, Q))
            hip.attention(Qc, KV[:, i * D:(i + 1) * D], KV[:, (nd + i) * D:(nd + i + 1) * D], Oc, self._bufs[f"{tag}:ca:lse"],
                          B, HEADS, Q, L, scale=float(HD) ** -0.5, dropout_p=dp, dropout_site=ds + 2, dropout_step=self._seed_dev,
                          d_o=dO, dq=dQc, dk=dKV[:, i * D:(i + 1) * D], dv=dKV[:, (nd + i) * D:(nd + i + 1) * D], delta=delta,
                          dropmask=self._dropmask(ds + 2, B, Q, L) if (a16 and dp > 0.0) else None)
            self._side(lambda d_out=d_out, Oc=Oc, dQc=dQc, q2=q2, cp=cp: hip.gemm_group([
                hip.linear_wgrad_call(d_out, Oc, G[f"{cp}/out_proj_kernel"], bias_grad=G[f"{cp}/out_proj_bias"]),
                hip.linear_wgrad_call(dQc, q2, G[f"{cp}/in_proj_kernel"][0:D], bias_grad=G[f"{cp}/in_proj_bias"][0:D])]))
            d_t1 = self.buf("scratch:d_t1", (BQ, D))
            # q2 = t1 + query_pos ; a2 = attn + t1: the q gradient goes to t1 (with the residual path) and to the query_pos sum
            hip.gemm_group([hip.linear_dgrad_call(dQc, Wc[0:D], acc_qpos, residual=acc_qpos),
                            hip.linear_dgrad_call(dQc, Wc[0:D], d_t1, residual=d_a2)])
            d_a1 = self.buf(f"scratch:d_a1{self._sx(tag)}", (BQ, D))
            d_out = self._ln_bwd(d_t1, a1, f"{pfx}/norm1", d_a1, f"{tag}:ln1", drop_site=ds + 1)
            # ---- self attention: qin = tgt + query_pos feeds q and k, tgt feeds v and the residual
            d_tgt = self.buf(f"scratch:d_tgt{i & 1}", (BQ, D)) if i > 0 else None      # layer 0: tgt is the constant zero target
            self._self_attn_bwd(f"{tag}:sa", f"{pfx}/self_attn", d_out, d_a1, qin, tgt, B, Q, d_tgt, site=ds, acc_qk=acc_qpos)
            d_next = d_tgt
        self._colsum(acc_qpos.view(B, Q * D), g_qpos.view(Q * D))      # d query_pos (sum over the batch)
        # ---- K / V projections of all layers: weight gradient of the gathered copy (scattered back into the per-layer
        #      in_proj gradients by one launch) and ONE data gradient GEMM with K = 2*nd*256
        hip.zero_(ct["gW"])
        hip.zero_(ct["gb"])
        hip.gemm_group([hip.linear_wgrad_call(dKV[:, 0:nd * D], mem_pos, ct["gW"][0:nd * D], bias_grad=ct["gb"][0:nd * D]),
                        hip.linear_wgrad_call(dKV[:, nd * D:], memory, ct["gW"][nd * D:], bias_grad=ct["gb"][nd * D:])])
        self._side_join()
        hip.flush_reduces()                            # the scatter reads the gathered gradient
        hip.multi_copy(ct["scatter"])
        d_mem = self.buf("scratch:d_mem", (B * L, D))
        hip.linear_dgrad(dKV, ct["Wkv"], d_mem)        # d(memory + pos) through K and d(memory) through V land on the same tensor
        self.phase("bwd encoder")
        # ---------------- encoder ----------------
        d_x = d_mem
        for i in reversed(range(self.num_enc)):
            pfx, tag = f"transformer/encoder/layer_{i}", f"enc{i}"
            x_in = self._bufs[f"enc{i - 1}:x2"] if i > 0 else self._bufs["enc:src0"]
            qk, a, x1, f = (self._bufs[f"{tag}:{n}"] for n in ("qk", "a", "x1", "f"))
            d_f = self.buf(f"scratch:e_d_f{self._sx(tag)}", (B * L, D))
            d_y = self._ln_bwd(d_x, f, f"{pfx}/norm2", d_f, f"{tag}:ln2", drop_site=16 * i + 3, want16=self.ffn16)
            d_x1 = self.buf("scratch:e_d_x1", (B * L, D))
            self._ffn_bwd(tag, pfx, d_y, d_f, x1, d_x1, x16=self._bufs.get(f"{tag}:x1h") if self.ffn16 else None)
            d_a = self.buf(f"scratch:e_d_a{self._sx(tag)}", (B * L, D))
            d_out = self._ln_bwd(d_x1, a, f"{pfx}/norm1", d_a, f"{tag}:ln1", drop_site=16 * i + 1)
            d_xn = self.buf(f"scratch:e_d_x{i & 1}", (B * L, D))
            self._self_attn_bwd(f"{tag}:sa", f"{pfx}/self_attn", d_out, d_a, qk, x_in, B, L, d_xn, site=16 * i)
            d_x = d_xn
        if on_bucket:
            on_bucket(0)
        self.phase("bwd backbone")
        # ---------------- input_proj ----------------
        self._wgrad(2048, D, B * L, feat, 2048, d_x, D, G["input_proj/kernel"], D)
        self._colsum(d_x, G["input_proj/bias"])
        if not backbone:
            if on_bucket:
                for i in (1, 2, 3):
                    on_bucket(i)
            return
        adt = self._adt
        g = self.buf("scratch:g_feat", feat.shape, adt)
        hip.gemm(B * L, 2048, D, d_x, D, 1, self._w("input_proj/kernel"), D, 1, g, 2048, mask=feat, ldmask=2048)
        # ---------------- residual stages ----------------
        n_blocks = len(self._block_meta)
        tfb = self.tf_backbone
        f32c = 2 if adt == torch.bfloat16 else 1        # the subsample kernels copy 16-byte chunks: bf16 passes as C/2 "floats"

        def bias_grad(conv, bn, dz, rows, cols):
            """tf_backbone: db = scale * column sums of the gradient at the conv + BN output (the folded BN scales the bias)."""
            if tfb:
                hip.call("detr_hip_colsum_scaled", dz.data_ptr(), 1 if dz.dtype == torch.bfloat16 else 0, rows, cols, cols,
                         self.bn_scale[bn].data_ptr(), G[f"{conv}/bias"].data_ptr())

        # Weight gradients of a block do not feed the data-gradient chain (g -> dz2 -> dz1 -> gx): with WGRAD_STREAM they are
        # issued on a second stream, each behind an event of the main stream that follows its producer, so a chip-filling
        # but HBM-bound data-gradient GEMM and an MFMA/LDS-bound weight gradient share the CUs (inside the captured graph the
        # fork / join events become plain graph edges).  The scratch tensors the side stream reads (dz2, dz1: double-buffered
        # by block parity; gx) are only overwritten after the side work that read them has been waited for.
        ws_on = self._side_on
        main, wside = (self._side_main, self._wg_stream) if ws_on else (None, None)
        side_done = {}
        on_side = self._side

        def side_mark(bi):
            if ws_on:
                ev = torch.cuda.Event()
                ev.record(wside)
                side_done[bi] = ev

        def side_wait(bi):
            ev = side_done.pop(bi, None)
            if ev is not None:
                main.wait_event(ev)

        for bi in reversed(range(n_blocks)):
            m = self._block_meta[bi]
            side_wait(bi + 2)
            par = f":{bi & 1}" if ws_on else ""
            p, n, x, xs, x1, y1, y2 = m["p"], m["n"], m["x"], m["xs"], m["x1"], m["y1"], m["y2"]
            h, w, ho, wo, cin, d1, d2, stride = m["h"], m["w"], m["ho"], m["wo"], m["cin"], m["d1"], m["d2"], m["stride"]
            h1, w1, M1, s2 = m["h1"], m["w1"], m["M1"], m["s2"]
            M_in, M_out = B * h * w, B * ho * wo
            wk = "ws16" if self.compute == 1 else "ws"       # scaled kernels of the forward (bf16 shadow in bf16 mode)
            ws1 = self._bufs[f"{wk}:{n['conv1']}/kernel"]
            ws2 = self._bufs[f"{wk}:{n['conv2']}/kernel"]
            ws3 = self._bufs[f"{wk}:{n['conv3']}/kernel"]
            # conv3: g is the gradient w.r.t. (bn3(conv3(y2)) + identity), already ReLU-masked
            strided = m["first"] and stride == 2
            wsd = self._bufs[f"{wk}:{n['down']}/kernel"] if m["first"] else None
            idg, idg_ready = g, None
            # layer1's 64 -> 256 convolutions on bf16 tensors: g (274 MB at B = 8, 800 x 1333) is read once per convolution, not twice
            fused_ok = BWD_FUSED and self.compute == 1 and adt == torch.bfloat16 and d2 == 256 and g.is_cuda
            fuse3, fuse_down = fused_ok and d1 == 64, False
            if m["first"] and not (tfb and strided):
                # data gradient of the projection shortcut: only the block's last GEMM reads it (as its residual operand)
                idg = self.buf(f"scratch:idg:{cin}:{h}", (B, h, w, cin), adt)

                # (layer1, bf16: the shortcut's input gradient and weight gradient in ONE pass over g -- csrc/bwd_fused.hip)
                fuse_down = fused_ok and not strided and cin == 64

                def dg_down(g=g, idg=idg):
                    if strided:
                        dxs = self.buf(f"scratch:dxs:{cin}:{ho}", (B, ho, wo, cin), adt)
                        hip.gemm(M_out, cin, d2, g, d2, 1, wsd, d2, 1, dxs, cin)
                        hip.call("detr_hip_subsample2_bwd_f32", dxs.data_ptr(), idg.data_ptr(), B, h, w, cin // f32c, ho, wo)
                    elif fuse_down:
                        hip.conv1x1_bwd_fused(g.view(M_out, d2), xs.view(M_out, cin), wsd.view(cin, d2), idg.view(M_out, cin), G[f"{n['down']}/kernel"].view(cin, d2),
                                              self.buf(f"bwdfused:slabs:down:{M_out}", (hip.conv1x1_bwd_fused_scratch_floats(M_out),)),
                                              scale=self.bn_scale[n["bnd"]], use_mask=False)
                    else:
                        hip.gemm(M_out, cin, d2, g, d2, 1, wsd, d2, 1, idg, cin)
                on_side(dg_down)
                if ws_on:
                    idg_ready = torch.cuda.Event()
                    idg_ready.record(wside)

            def wg3(g=g, fuse3=fuse3, fuse_down=fuse_down):
                if not fuse3:
                    self._wgrad(d1, d2, M_out, y2, d1, g, d2, G[f"{n['conv3']}/kernel"], d2, scale=self.bn_scale[n["bn3"]])
                bias_grad(n["conv3"], n["bn3"], g, M_out, d2)
                if m["first"]:
                    if not fuse_down:
                        self._wgrad(cin, d2, M_out, xs, cin, g, d2, G[f"{n['down']}/kernel"], d2, scale=self.bn_scale[n["bnd"]])
                    bias_grad(n["down"], n["bnd"], g, M_out, d2)
            on_side(wg3)
            dz2 = self.buf(f"scratch:dz2:{d1}:{ho}{par}", (B, ho, wo, d1), adt)
            if fuse3:
                hip.conv1x1_bwd_fused(g.view(M_out, d2), y2.view(M_out, d1), ws3.view(d1, d2), dz2.view(M_out, d1), G[f"{n['conv3']}/kernel"].view(d1, d2),
                                      self.buf(f"bwdfused:slabs:{M_out}", (hip.conv1x1_bwd_fused_scratch_floats(M_out),)),
                                      scale=self.bn_scale[n["bn3"]], use_mask=True)
            else:
                hip.gemm(M_out, d1, d2, g, d2, 1, ws3, d2, 1, dz2, d1, mask=y2, ldmask=d1)
            # conv2 (3x3)

            def wg2():
                hip.conv3x3(2, y1, dz2, G[f"{n['conv2']}/kernel"], B, h1, w1, d1, ho, wo, d1, s2, scale=self.bn_scale[n["bn2"]])
                bias_grad(n["conv2"], n["bn2"], dz2, M_out, d1)
            on_side(wg2)
            dz1 = self.buf(f"scratch:dz1:{d1}:{h1}{par}", (B, h1, w1, d1), adt)
            hip.conv3x3(1, dz2, ws2, dz1, B, h1, w1, d1, ho, wo, d1, s2, mask=y1)
            # conv1

            def wg1():
                self._wgrad(cin, d1, M1, x1, cin, dz1, d1, G[f"{n['conv1']}/kernel"], d1, scale=self.bn_scale[n["bn1"]])
                bias_grad(n["conv1"], n["bn1"], dz1, M1, d1)
            on_side(wg1)
            side_mark(bi)
            side_wait(bi + 1)               # gx of this parity was read (as g) by the side work of block bi + 1
            is_first_block = bi == 0
            gx = self.buf(f"scratch:gx:{cin}:{h}:{bi & 1}", (B, h, w, cin), adt)
            mask = None if is_first_block else x           # x = ReLU output of the previous block
            xbits = None if is_first_block else self._block_meta[bi - 1].get("out_bits")      # ... or its sign bits
            if tfb and strided:
                # both branches read the subsampled input: d_xs = g @ Wd^T + dz1 @ W1^T (ReLU-masked at the sampled pixels),
                # scattered back into the zero-filled full-resolution gradient
                dxs = self.buf(f"scratch:dxs:{cin}:{ho}", (B, ho, wo, cin), adt)
                hip.gemm(M_out, cin, d2, g, d2, 1, wsd, d2, 1, dxs, cin)
                hip.gemm(M_out, cin, d1, dz1, d1, 1, ws1, d1, 1, dxs, cin, residual=dxs, ldr=cin,
                         mask=(None if is_first_block else xs), ldmask=(0 if is_first_block else cin))
                hip.call("detr_hip_subsample2_bwd_f32", dxs.data_ptr(), gx.data_ptr(), B, h, w, cin // f32c, ho, wo)
            else:
                if idg_ready is not None:
                    main.wait_event(idg_ready)
                if xbits is not None:
                    hip.gemm(M_in, cin, d1, dz1, d1, 1, ws1, d1, 1, gx, cin, residual=idg, ldr=cin, mask=xbits, ldmask=xbits.stride(0))
                else:
                    hip.gemm(M_in, cin, d1, dz1, d1, 1, ws1, d1, 1, gx, cin, residual=idg, ldr=cin, mask=mask,
                             ldmask=(cin if mask is not None else 0))
            g = gx
            if on_bucket:
                if p == block_names(3, 0, tfb)["tag"]:
                    on_bucket(1)
                elif p == block_names(2, 0, tfb)["tag"]:
                    on_bucket(2)
        self._side_join()
        # ---------------- stem ----------------
        stem, pool, amax = (self._bufs[f"stem:{n}"] for n in ("out", "pool", "amax"))
        H1, W1 = stem.shape[1], stem.shape[2]
        H2, W2 = pool.shape[1], pool.shape[2]
        d_stem = self.buf("scratch:d_stem", stem.shape, adt)
        hip.call("detr_hip_maxpool3x3s2_bwd_bf16" if adt == torch.bfloat16 else "detr_hip_maxpool3x3s2_bwd_f32", g.data_ptr(),
                 amax.data_ptr(), stem.data_ptr(), d_stem.data_ptr(), B, H1, W1, 64, H2, W2)
        M1 = B * H1 * W1
        hip.stem_conv(2, self.images, d_stem, G[f"{self._stem['conv']}/kernel"], B, self._shape[1], self._shape[2], H1, W1,
                      scale=self.bn_scale[self._stem["bn"]], split=max(1, min(512, M1 // 4096)))
        if self.tf_backbone:
            hip.call("detr_hip_colsum_scaled", d_stem.data_ptr(), 1 if d_stem.dtype == torch.bfloat16 else 0, B * H1 * W1, 64, 64,
                     self.bn_scale[self._stem["bn"]].data_ptr(), G[f"{self._stem['conv']}/bias"].data_ptr())
        if on_bucket:
            on_bucket(3)
