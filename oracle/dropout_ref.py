"""ORACLE (test infrastructure only) -- the counter-hash dropout masks of the HIP path, restated in numpy.

The reference uses Keras Dropout(0.1) in training mode (transformer.py:149,196,248: on the attention
probabilities :341, after every attention block and inside the FFN :169-176,216-232).  TensorFlow's RNG
stream cannot be reproduced (and TF is absent), so parity under dropout is defined with the MASKS of the
HIP path: one keyed hash32(key(site, step seed), idx >> 1) per element pair, its low (even idx) / high (odd idx) 16 bits
compared with p * 2^16; kept values scaled by 1/(1-p) (csrc/common.h::drop_key, drop_hash, drop_keep).  With these masks the oracle and the device compute the same function."""
import numpy as np
import torch

M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    """csrc/common.h::mix32 on uint64-held 32-bit values."""
    x = np.uint64(x) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def step_seed(base_seed, step_no, rank=0):
    """The per-step seed the engine writes to device memory (detr_tf/engine.py::_step_seed): base seed, step counter
    and data-parallel rank mixed on the host."""
    s = int(mix32((int(base_seed) + 0x9E3779B9 * int(step_no)) & 0xFFFFFFFF))
    return int(mix32(s ^ ((int(rank) * 0x85EBCA6B) & 0xFFFFFFFF)))


def drop_key(site, step):
    """csrc/common.h::drop_key: key of dropout site `site` in the training step whose seed is `step`."""
    return int(mix32((int(step) + int(site) * 0x9E3779B9) & 0xFFFFFFFF))


def drop_hash(key, idx):
    """csrc/common.h::drop_hash (idx = element-pair index, uint64)."""
    idx = idx.astype(np.uint64)
    key = np.uint64(key & 0xFFFFFFFF)
    key2 = (key * np.uint64(0x85EBCA6B) + np.uint64(0xC2B2AE35)) & M32
    x = (idx & M32) ^ key
    x ^= ((idx >> np.uint64(32)) * np.uint64(0x9E3779B9)) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= key2
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def keep_mask(key, idx, p):
    """keep flags of elements `idx` for the dropout key `key` (= drop_key(site, step seed))."""
    idx = np.asarray(idx).astype(np.uint64)
    thresh = np.uint64(int(np.float32(p) * np.float32(65536.0)))
    h = drop_hash(key, idx >> np.uint64(1))
    half = np.where((idx & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return half >= thresh


def attn_index(BH, T, S):
    """Element index of the attention-probability dropout (attention_f32.hip): rows padded to an even length."""
    Sp = (S + 1) & ~1
    r, k = np.meshgrid(np.arange(BH * T), np.arange(S), indexing="ij")
    return (r * Sp + k).reshape(BH, T, S)


class Dropper:
    """drop(seed, x, layout): layout "lbc" = sequence-first activations [L,B,C] whose device twin is the
    batch-first matrix [B*L, C] (element index (b*L+l)*C+c); "flat" = row-major index of x itself; "attn" =
    attention probabilities [B*H, T, S] with the row stride padded to an even length (attn_index)."""

    def __init__(self, p, step_seed_value):
        """step_seed_value: the uint32 per-step seed (engine._drop[1] / step_seed()); the first argument of a call is the
        dropout SITE id (16 * layer + k, the numbering of oracle/detr_ref.py)."""
        self.p, self.base = float(p), int(step_seed_value)

    def __call__(self, seed_off, x, layout):
        if self.p <= 0.0:
            return x
        shp = tuple(x.shape)
        if layout == "lbc":
            L, B, C = shp
            l, b, c = np.meshgrid(np.arange(L), np.arange(B), np.arange(C), indexing="ij")
            idx = (b * L + l) * C + c
        elif layout == "attn":
            idx = attn_index(*shp)
        else:
            idx = np.arange(int(np.prod(shp))).reshape(shp)
        keep = keep_mask(drop_key(seed_off, self.base), idx, self.p)
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(self.p))
        m = torch.from_numpy(keep.astype(np.float32) * float(scale)).to(x.dtype)
        return x * m
