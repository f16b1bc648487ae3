"""GPU parity tests of the all-bf16 attention core (detr_attn_desc.io_dtype = 1, csrc/attention_dma.hip; reference
detr_tf/networks/transformer.py:285-356) through the C ABI, against an fp64 torch reference that uses the oracle's dropout masks
(oracle/dropout_ref.py -- the same keyed counter hash the device evaluates), and of the keep-bit generator against those masks
bit for bit.  Operands are rounded to bf16 on the host first, so the only differences left are the kernels' own roundings
(P, dS, O, dQ, dK, dV in bf16; fp32 accumulation): tolerances are stated per check."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
H, HD = 8, 32
D = H * HD
LOG2E = 1.4426950408889634


def close(a, b, rtol, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert err <= rtol * scale, f"{what}: max abs err {err:.3e} > {rtol * scale:.3e} (scale {scale:.3e})"


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _unpack_bits(words, rows, cols_pad, cols):
    """[rows, cols_pad / 32 tiles...] helper: words [tiles, rows] uint32 with bit j = column tile * 32 + j -> bool [rows, cols]."""
    w = words.astype(np.uint32)                                   # [tiles, rows]
    bits = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).astype(bool)      # [tiles, rows, 32]
    return bits.transpose(1, 0, 2).reshape(rows, cols_pad)[:, :cols]


@pytest.mark.parametrize("B,T,S,p", [(1, 100, 1050, 0.1), (2, 70, 333, 0.1), (1, 37, 5, 0.25), (1, 64, 64, 0.1), (1, 130, 97, 0.5)])
def test_attention_dropmask_bits_equal_the_oracle_masks(hip, B, T, S, p):
    """detr_hip_attention_dropmask: one word per (query, 32-key tile), bit = key, against oracle/dropout_ref.keep_mask -- the function the
    fp32 kernels and the round-3 bf16 kernels evaluate per element."""
    from oracle import dropout_ref as DR
    site, step = 91, 0xC0FFEE11
    stepd = torch.tensor([step - (1 << 32)] + [0] * 7, dtype=torch.int32, device=DEV)
    words = hip.attention_dropmask_words(B, H, T, S)
    nqt, nkt = -(-T // 32), -(-S // 32)
    assert words == B * H * nqt * nkt * 32
    mask = torch.zeros(words, dtype=torch.int32, device=DEV)
    hip.attention_dropmask(mask, B, H, T, S, dropout_p=p, dropout_site=site, dropout_step=stepd)
    torch.cuda.synchronize()
    m = mask.cpu().numpy().view(np.uint32)
    keep = DR.keep_mask(DR.drop_key(site, step), DR.attn_index(B * H, T, S), p)          # [BH, T, S]
    mq = m.reshape(B * H, nkt, nqt * 32)
    for bh in range(B * H):
        got_q = _unpack_bits(mq[bh][:, :T], T, nkt * 32, S)                              # rows = queries, columns = keys
        assert np.array_equal(got_q, keep[bh]), f"keep bits differ (problem {bh})"


def test_attention_dropmask_many_sites_in_one_launch(hip):
    """detr_hip_attention_dropmask_many: the sites of a step (different T, S, site ids) from one launch equal the single-site calls."""
    B, p, step = 2, 0.1, 0x1234ABCD
    stepd = torch.tensor([step] + [0] * 7, dtype=torch.int32, device=DEV)
    shapes = [(96, 96, 0), (96, 96, 16), (37, 96, 512), (37, 37, 514), (300, 40, 7)]
    many = [torch.zeros(hip.attention_dropmask_words(B, H, T, S), dtype=torch.int32, device=DEV) for T, S, _ in shapes]
    hip.attention_dropmask_many([(m, T, S, site) for m, (T, S, site) in zip(many, shapes)], B, H, dropout_p=p, dropout_step=stepd)
    for m, (T, S, site) in zip(many, shapes):
        one = torch.zeros_like(m)
        hip.attention_dropmask(one, B, H, T, S, dropout_p=p, dropout_site=site, dropout_step=stepd)
        assert torch.equal(m, one), f"site {site}"


def _reference(q, k, v, do, scale, keep, p):
    B, T = q.shape[0], q.shape[1]
    qh, kh, vh = (t.view(B, -1, H, HD).transpose(1, 2) for t in (q, k, v))
    sc = (qh * scale) @ kh.transpose(-1, -2)
    w = torch.softmax(sc, dim=-1)
    if keep is not None:
        w = torch.where(keep, w / (1.0 - p), torch.zeros_like(w))
    o = (w @ vh).transpose(1, 2).reshape(B, T, D)
    o.backward(do)
    return o.detach(), torch.logsumexp(sc, dim=-1).detach()


def _run(hip, B, T, S, p, split, seed, spike=False, packed=False):
    from oracle import dropout_ref as DR
    torch.manual_seed(seed)
    site, step, scale = 77, 0xBEEF1234, HD ** -0.5
    # the stored query operand is bf16(scale * log2(e) * q) -- the projection GEMM's epilogue rounds ONCE, after the alpha --, so the
    # reference takes q := stored / (scale * log2 e) as the exact unscaled query
    qs = (torch.randn(B, T, D, dtype=torch.float64) * 2.0 * (scale * LOG2E)).to(torch.bfloat16)
    q = (qs.to(torch.float64) / (scale * LOG2E)).requires_grad_(True)
    k = _bf(torch.randn(B, S, D, dtype=torch.float64)).requires_grad_(True)
    v = _bf(torch.randn(B, S, D, dtype=torch.float64)).requires_grad_(True)
    if spike:                      # one late key dominates one query of head 0: the running reference must move late in the stream
        with torch.no_grad():
            k[0, S - 1, :HD] = _bf(q[0, T // 2, :HD].detach() * 4.0)
    do = _bf(torch.randn(B, T, D, dtype=torch.float64))
    keep = None
    if p > 0.0:
        keep = torch.from_numpy(DR.keep_mask(DR.drop_key(site, step), DR.attn_index(B * H, T, S).reshape(B, H, T, S), p))
    o_ref, lse_ref = _reference(q, k, v, do, scale, keep, p)
    stepd = torch.tensor([step - (1 << 32)] + [0] * 7, dtype=torch.int32, device=DEV)
    bf = lambda t: t.detach().to(torch.bfloat16)
    if packed:                     # operands as column blocks of packed buffers with their own row strides
        qbuf = torch.full((B * T, 3 * D), 3.0, dtype=torch.bfloat16, device=DEV)
        kvbuf = torch.full((B * S, 5 * D), -2.0, dtype=torch.bfloat16, device=DEV)
        qbuf[:, D:2 * D] = qs.view(-1, D).to(DEV)
        kvbuf[:, D:2 * D] = bf(k).view(-1, D).to(DEV)
        kvbuf[:, 3 * D:4 * D] = bf(v).view(-1, D).to(DEV)
        qd, kd, vd = qbuf[:, D:2 * D], kvbuf[:, D:2 * D], kvbuf[:, 3 * D:4 * D]
        dqb, dkvb = torch.full_like(qbuf, 5.0), torch.full_like(kvbuf, 5.0)
        dqd, dkd, dvd = dqb[:, 0:D], dkvb[:, 2 * D:3 * D], dkvb[:, 4 * D:]
    else:
        qd, kd, vd = qs.view(-1, D).to(DEV), bf(k).view(-1, D).to(DEV), bf(v).view(-1, D).to(DEV)
        dqd, dkd, dvd = (torch.full_like(t, 3.0) for t in (qd, kd, vd))
    dod = bf(do).view(-1, D).to(DEV)
    od = torch.full((B * T, D), 7.0, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(B * H, T, device=DEV)
    stats = torch.zeros(2 * B * H * T, device=DEV)
    mask = None
    if p > 0.0:
        mask = torch.zeros(hip.attention_dropmask_words(B, H, T, S), dtype=torch.int32, device=DEV)
        hip.attention_dropmask(mask, B, H, T, S, dropout_p=p, dropout_site=site, dropout_step=stepd)
    hip.set_tuning("DETR_HIP_ATTN_SPLIT", split if split else None)
    try:
        kw = dict(scale=scale, dropout_p=p, dropout_site=site, dropout_step=stepd, dropmask=mask)
        hip.attention(qd, kd, vd, od, lse, B, H, T, S, **kw)
        # the backward reads the bf16 O the forward wrote
        hip.attention(qd, kd, vd, od, lse, B, H, T, S, d_o=dod, dq=dqd, dk=dkd, dv=dvd, delta=stats, **kw)
        torch.cuda.synchronize()
    finally:
        hip.set_tuning("DETR_HIP_ATTN_SPLIT", None)
    tag = f"B{B} T{T} S{S} p{p} split{split}"
    # forward: P and O are rounded to bf16 (2^-9 relative each), sums in fp32
    close(od.float().view(B, T, D), o_ref, rtol=1.0e-2, what=f"bf16-io attention fwd ({tag})")
    close(lse.view(B, H, T), lse_ref, rtol=2e-3, what=f"bf16-io attention lse ({tag})")
    # backward: on top of the bf16 roundings of P / dS / the outputs, delta = rowsum(dO * O) is formed from the bf16 O the forward stored
    # (2^-9 relative on a number of size |dO| |O|), which a peaked row (the spiked key: p ~ 1) passes straight into dS.  Measured on these
    # shapes: 1.0-2.3 % of the largest gradient entry
    close(dqd.float().reshape(B, T, D), q.grad, rtol=3.0e-2, what=f"bf16-io attention dq ({tag})")
    close(dkd.float().reshape(B, S, D), k.grad, rtol=3.0e-2, what=f"bf16-io attention dk ({tag})")
    close(dvd.float().reshape(B, S, D), v.grad, rtol=3.0e-2, what=f"bf16-io attention dv ({tag})")
    if packed:                     # nothing outside the addressed column blocks was touched
        assert bool((dqb[:, D:] == 5.0).all()) and bool((dkvb[:, :2 * D] == 5.0).all()) and bool((dkvb[:, 3 * D:4 * D] == 5.0).all())
    return od, lse, dqd, dkd, dvd


@pytest.mark.parametrize("B,T,S,p", [(2, 1050, 1050, 0.1), (2, 1050, 1050, 0.0), (2, 100, 1050, 0.1), (3, 100, 100, 0.1), (1, 37, 5, 0.0),
                                     (1, 300, 1344, 0.1), (2, 70, 333, 0.1), (1, 64, 32, 0.1), (1, 65, 33, 0.0)])
def test_attention_bf16_io_vs_fp64(hip, B, T, S, p):
    """Forward (O, LSE) and backward (dQ w.r.t. the unscaled q, dK, dV) on the step's shapes and on ragged ones (T, S not multiples
    of 32 / 64; S < 32; a single key tile), with and without dropout, launch heuristic."""
    _run(hip, B, T, S, p, split=0, seed=B * 77 + T + S, spike=True)


@pytest.mark.parametrize("split", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("B,T,S,p", [(1, 200, 1050, 0.1), (1, 1050, 200, 0.1), (2, 70, 40, 0.0), (1, 300, 300, 0.1)])
def test_attention_bf16_io_runs_per_workgroup(hip, split, B, T, S, p):
    """The streamed dimension cut into 1 .. 8 runs per workgroup (DETR_HIP_ATTN_SPLIT; runs capped at 4 tiles each, so small shapes
    collapse to fewer): the merge of the partial softmaxes, the partial dQ / dK / dV sums and runs that own no tiles."""
    _run(hip, B, T, S, p, split=split, seed=split * 13 + T + S, spike=True)


def test_attention_bf16_io_packed_operands(hip):
    """q / k / v / gradients as column blocks of packed [rows, 768] / [rows, 1280] buffers (the engine's layouts)."""
    _run(hip, 2, 100, 1050, 0.1, split=0, seed=5, packed=True)
    _run(hip, 1, 333, 100, 0.0, split=0, seed=6, packed=True)


def test_attention_bf16_io_is_deterministic_and_split_invariant_in_distribution(hip):
    """Two runs of the same call are bit-identical (no atomics anywhere; fixed merge order)."""
    a = _run(hip, 2, 300, 1050, 0.1, split=0, seed=11)
    b = _run(hip, 2, 300, 1050, 0.1, split=0, seed=11)
    for x, y, name in zip(a, b, ("o", "lse", "dq", "dk", "dv")):
        assert torch.equal(x, y), f"{name} differs between two identical calls"


def test_attention_bf16_io_rejects_bad_descriptors(hip):
    """Host-side validation of the new descriptor fields."""
    B, T, S = 1, 64, 64
    q = torch.zeros(B * T, D, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros_like(q)
    lse = torch.zeros(B * H, T, device=DEV)
    with pytest.raises(ValueError):        # dropout without keep bits
        hip.attention(q, q, q, o, lse, B, H, T, S, dropout_p=0.1)
    qbad = torch.zeros(B * T, D + 4, dtype=torch.bfloat16, device=DEV)[:, :D]       # row stride not a multiple of 8
    with pytest.raises(RuntimeError):
        hip.attention(qbad, q, q, o, lse, B, H, T, S)
    with pytest.raises(ValueError):        # backward scratch too small
        hip.attention(q, q, q, o, lse, B, H, T, S, d_o=q, dq=o, dk=o, dv=o, delta=torch.zeros(B * H * T, device=DEV))
