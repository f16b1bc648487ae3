"""GPU parity tests of every HIP kernel, called through the C ABI (ctypes), against plain
torch-CPU fp32/fp64 references of the same op and against the oracle (SciPy for the matcher).
Tolerances are stated per test; GEMM-class kernels use the exact-f32 MFMA so only the summation
order differs from the CPU (rel 2e-5 of the row scale)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def g(t):
    return t.to(DEV).contiguous()


@pytest.fixture(autouse=True)
def _reset_library_tuning():
    """The library reads its DETR_HIP_* tuning variables once (at load) and on detr_hip_reload_tuning(): a test that switches
    one through hip.set_tuning() must not leak it into the next test."""
    import os
    yield
    from detr_tf import _hip
    leaked = [k for k in ("DETR_HIP_STEM_ROWS", "DETR_HIP_CONV_HALO", "DETR_HIP_DGRAD_S2_CLASSES", "DETR_HIP_GEMM_STREAM", "DETR_HIP_GEMM_TILE", "DETR_HIP_ATTN_SPLIT", "DETR_HIP_GEMM_K64", "DETR_HIP_EPI_WIDE",
                          "DETR_HIP_GEMM_RING", "DETR_HIP_RING_NS", "DETR_HIP_RING_BN", "DETR_HIP_RING_WGS", "DETR_HIP_RING_ROWS", "DETR_HIP_RING_ABLATE", "DETR_HIP_CONV_DMA", "DETR_HIP_WGRAD_FUSED")
              if k in os.environ]
    for k in leaked:
        _hip.set_tuning(k, None)


def close(a, b, rtol=2e-5, atol=None, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    scale = float(b.abs().max()) + 1e-30
    atol = rtol * scale if atol is None else atol
    err = float((a - b).abs().max())
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.3e} (scale {scale:.3e})"


# ------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 92, 256), (1000, 256, 147), (77, 40, 33), (8400, 64, 64),
                                   (520, 2048, 256), (130, 32, 100)])
@pytest.mark.parametrize("ak,bk", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_layouts(hip, M, N, K, ak, bk):
    torch.manual_seed(M * 7 + N * 3 + K + ak * 2 + bk)
    A = torch.randn(M, K)
    B = torch.randn(K, N)
    ref = A.double() @ B.double()
    lda = (K if ak else M) + 4
    ldb = (K if bk else N) + 4
    Am = torch.zeros((M, lda) if ak else (K, lda))
    Bm = torch.zeros((N, ldb) if bk else (K, ldb))
    if ak:
        Am[:, :K] = A
    else:
        Am[:, :M] = A.t()
    if bk:
        Bm[:, :K] = B.t()
    else:
        Bm[:, :N] = B
    Ad, Bd = g(Am), g(Bm)
    C = torch.full((M, N + 4), 7.0, device=DEV)
    hip.gemm(M, N, K, Ad, lda, ak, Bd, ldb, bk, C, N + 4)
    torch.cuda.synchronize()
    close(C[:, :N], ref, what=f"gemm {M}x{N}x{K} ak={ak} bk={bk}")
    assert float((C[:, N:] - 7.0).abs().max()) == 0.0, "gemm wrote outside its columns"


def test_gemm_unaligned_scalar_path(hip):
    torch.manual_seed(1)
    M, N, K = 70, 50, 37
    A = torch.randn(M, K)
    B = torch.randn(N, K)
    Ad, Bd = g(A), g(B)          # ld = 37 -> not a multiple of 4 -> scalar loads
    C = torch.zeros(M, N, device=DEV)
    hip.gemm(M, N, K, Ad, K, 1, Bd, K, 1, C, N)
    close(C, A.double() @ B.double().t(), what="gemm scalar path")


def test_gemm_epilogue(hip):
    torch.manual_seed(2)
    M, N, K = 333, 200, 96
    A, W = torch.randn(M, K), torch.randn(N, K)
    scale, bias = torch.rand(N) + 0.5, torch.randn(N)
    R, Mk = torch.randn(M, N), torch.randn(M, N)
    acc = A.double() @ W.double().t()
    for act in (0, 1, 2):
        ref = (acc * scale.double() + bias.double()) * 0.25 + R.double()
        if act == 1:
            ref = ref.clamp_min(0)
        elif act == 2:
            ref = torch.sigmoid(ref)
        ref = torch.where(Mk.double() > 0, ref, torch.zeros_like(ref))
        C = torch.zeros(M, N, device=DEV)
        hip.gemm(M, N, K, g(A), K, 1, g(W), K, 1, C, N, alpha=0.25, scale=g(scale), bias=g(bias), residual=g(R), ldr=N,
                 mask=g(Mk), ldmask=N, act=act)
        close(C, ref, what=f"gemm epilogue act={act}")


def test_gemm_split_k_accumulates(hip):
    torch.manual_seed(3)
    M, N, K = 64, 256, 5000
    A, B = torch.randn(K, M), torch.randn(K, N)
    scale = torch.rand(N) + 0.5
    C0 = torch.randn(M, N)
    C = g(C0.clone())
    hip.gemm(M, N, K, g(A), M, 0, g(B), N, 0, C, N, alpha=0.5, scale=g(scale), split_k=37)
    ref = C0.double() + 0.5 * (A.double().t() @ B.double()) * scale.double()
    close(C, ref, rtol=5e-5, what="gemm split-k")


def test_split_k_deterministic_workspace_path(hip):
    """split-K through the caller-provided workspace (partials + reduce launch): same result as the
    atomic path, bit-identical run to run; also the conv3x3 wgrad split reduction."""
    torch.manual_seed(13)
    M, N, K = 147, 64, 30000
    A, B = torch.randn(K, 160), torch.randn(K, N)
    scale = torch.rand(N) + 0.5
    C0 = torch.randn(M, N)
    Ad, Bd, sd = g(A), g(B), g(scale)
    ws = torch.empty(8 * 1024 * 1024, device=DEV)
    outs = []
    for rep in range(2):
        C = g(C0.clone())
        hip.gemm(M, N, K, Ad, 160, 0, Bd, N, 0, C, N, alpha=0.5, scale=sd, split_k=97, workspace=ws)
        outs.append(C.clone())
    ref = C0.double() + 0.5 * (A[:, :M].double().t() @ B.double()) * scale.double()
    close(outs[0], ref, rtol=5e-5, what="gemm split-k (workspace)")
    assert torch.equal(outs[0], outs[1]), "workspace split-K must be deterministic"
    old = hip.WORKSPACE
    try:
        hip.WORKSPACE = ws
        Nn, H, W, Ci, Co = 2, 30, 41, 64, 64
        x = torch.randn(Nn, H, W, Ci, dtype=torch.float64)
        w = torch.zeros(3, 3, Ci, Co, dtype=torch.float64, requires_grad=True)
        z = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), None, stride=1, padding=1).permute(0, 2, 3, 1)
        dz = torch.randn_like(z)
        z.backward(dz)
        xd, dzd = g(x.float()), g(dz.float())
        res = []
        for rep in range(2):
            dwd = torch.zeros(3, 3, Ci, Co, device=DEV)
            hip.conv3x3(2, xd, dzd, dwd, Nn, H, W, Ci, H, W, Co, 1, scale=sd, split=7)
            res.append(dwd)
        close(res[0], w.grad * scale.double(), rtol=5e-5, what="conv3x3 wgrad (workspace)")
        assert torch.equal(res[0], res[1])
    finally:
        hip.WORKSPACE = old


@pytest.mark.parametrize("M,N,K,sk,compute", [(256, 256, 8400, 32, 0), (92, 256, 4800, 8, 0), (4, 256, 4800, 1, 0),
                                               (2048, 256, 800, 3, 0), (256, 256, 8400, 32, 1), (768, 256, 801, 1, 1),
                                               (128, 512, 20000, 24, 1), (92, 256, 4800, 8, 1)])
def test_gemm_fused_rowsum_bias_gradient(hip, M, N, K, sk, compute):
    """weight-gradient GEMM dW = dy^T x with the bias gradient (column sums of dy = row sums of the A operand) fused
    into the same launch: exact fp32 sums in both compute modes, split and unsplit, ragged M, deterministic."""
    torch.manual_seed(M + N + K + sk)
    dy, x = torch.randn(K, M), torch.randn(K, N)
    dyd, xd = g(dy), g(x)
    ws = torch.empty(16 * 1024 * 1024, device=DEV)
    b0 = torch.randn(M)
    outs = []
    for rep in range(2):
        dw, db = torch.zeros(M, N, device=DEV), g(b0.clone())
        kw = dict(split_k=sk) if sk > 1 else dict(residual=dw, ldr=N)
        hip.gemm(M, N, K, dyd, M, 0, xd, N, 0, dw, N, alpha=0.5, workspace=ws, compute=compute, rowsum_a=db, rowsum_alpha=0.5, **kw)
        outs.append((dw.clone(), db.clone()))
    ref_w = 0.5 * ((_bf(dy) if compute else dy.double()).t() @ (_bf(x) if compute else x.double()))
    close(outs[0][0], ref_w, rtol=5e-5, what="fused-rowsum gemm dW")
    close(outs[0][1], b0.double() + 0.5 * dy.double().sum(0), rtol=2e-5, what="fused-rowsum bias gradient")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "must be deterministic"


@pytest.mark.parametrize("M,N,K,sk", [(256, 2048, 8400, 8), (2048, 256, 8400, 8), (256, 256, 800, 6), (512, 256, 801, 1)])
def test_gemm_fused_rowsum_with_bf16_stored_operands(hip, M, N, K, sk):
    """The same fused bias gradient when the A operand (dy) and / or the B operand (x) are STORED in bf16 (the FFN tensors
    of the bf16 transformer: LoaderMNth shares LoaderMNt's unit map): dW from the bf16 values, db = fp32 sum of them."""
    torch.manual_seed(M + N + K)
    dy16, x16 = torch.randn(K, M).to(torch.bfloat16), torch.randn(K, N).to(torch.bfloat16)
    ws = torch.empty(16 * 1024 * 1024, device=DEV)
    ref_w = dy16.double().t() @ x16.double()
    ref_b = dy16.double().sum(0)
    for a16, b16 in ((True, True), (True, False), (False, True)):
        A = g(dy16 if a16 else dy16.float())
        Bm = g(x16 if b16 else x16.float())
        outs = []
        for rep in range(2):
            dw, db = torch.zeros(M, N, device=DEV), torch.zeros(M, device=DEV)
            kw = dict(split_k=sk) if sk > 1 else dict(residual=dw, ldr=N)
            hip.gemm(M, N, K, A, M, 0, Bm, N, 0, dw, N, workspace=ws, compute=1, rowsum_a=db, **kw)
            outs.append((dw.clone(), db.clone()))
        close(outs[0][0], ref_w, rtol=5e-5, what=f"bf16-stored wgrad a16={a16} b16={b16}")
        close(outs[0][1], ref_b, rtol=2e-5, what=f"bf16-stored bias gradient a16={a16} b16={b16}")
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # and through the grouped launch the engine uses for the two FFN weight gradients
    dws = [torch.zeros(M, N, device=DEV), torch.zeros(N, M, device=DEV)]
    dbs = [torch.zeros(M, device=DEV), torch.zeros(N, device=DEV)]
    hip.ensure_workspace(DEV)
    old = hip.COMPUTE_BF16
    hip.COMPUTE_BF16 = 1
    try:
        hip.gemm_group([hip.linear_wgrad_call(g(dy16), g(x16), dws[0], bias_grad=dbs[0]),
                        hip.linear_wgrad_call(g(x16), g(dy16), dws[1], bias_grad=dbs[1])])
    finally:
        hip.COMPUTE_BF16 = old
    close(dws[0], ref_w, rtol=5e-5, what="grouped bf16-stored wgrad 0")
    close(dws[1], ref_w.t(), rtol=5e-5, what="grouped bf16-stored wgrad 1")
    close(dbs[0], ref_b, rtol=2e-5, what="grouped bf16-stored bias gradient 0")
    close(dbs[1], x16.double().sum(0), rtol=2e-5, what="grouped bf16-stored bias gradient 1")


def test_gemm_batched_attention_layout(hip):
    """scores[b,h] = Q_h K_h^T and O = P V_h on the [B, L, heads*32] layout used by the transformer."""
    torch.manual_seed(4)
    Bn, H, T, S, hd = 2, 8, 100, 150, 32
    D = H * hd
    q, k, v = torch.randn(Bn, T, D), torch.randn(Bn, S, D), torch.randn(Bn, S, D)
    Sp = 152
    sc = torch.zeros(Bn * H, T, Sp, device=DEV)
    qd, kd, vd = g(q), g(k), g(v)
    hip.gemm(T, S, hd, qd, D, 1, kd, D, 1, sc, Sp, batch=Bn * H, batch_inner=H, sA=(T * D, hd), sB=(S * D, hd),
             sC=(H * T * Sp, T * Sp))
    ref = torch.einsum("bthd,bshd->bhts", q.view(Bn, T, H, hd).double(), k.view(Bn, S, H, hd).double())
    close(sc[:, :, :S].view(Bn, H, T, S), ref, what="batched QK^T")
    p = torch.softmax(ref, -1).float()
    pd = torch.zeros(Bn * H, T, Sp, device=DEV)
    pd[:, :, :S] = p.view(Bn * H, T, S).to(DEV)
    o = torch.zeros(Bn, T, D, device=DEV)
    hip.gemm(T, hd, S, pd, Sp, 1, vd, D, 0, o, D, batch=Bn * H, batch_inner=H, sA=(H * T * Sp, T * Sp), sB=(S * D, hd),
             sC=(T * D, hd))
    oref = torch.einsum("bhts,bshd->bthd", p.double(), v.view(Bn, S, H, hd).double()).reshape(Bn, T, D)
    close(o, oref, what="batched PV")


@pytest.mark.parametrize("B,T,S", [(2, 1050, 1050), (2, 100, 1050), (3, 100, 100), (1, 37, 5), (2, 300, 1344), (1, 12, 12)])
def test_fused_attention_fwd_bwd(hip, B, T, S):
    """Fused attention (scores never stored) vs an fp64 torch reference, incl. ragged tiles, S < 32 and a
    spiked key that moves the running max late in the stream (online-softmax rescale branch)."""
    torch.manual_seed(B * 1000 + T + S)
    H, hd = 8, 32
    D = H * hd
    q = torch.randn(B, T, D, dtype=torch.float64) * 0.6
    k = torch.randn(B, S, D, dtype=torch.float64)
    v = torch.randn(B, S, D, dtype=torch.float64)
    k[0, S - 1, :hd] = q[0, T // 2, :hd] * 6.0          # last key dominates one query of head 0
    q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    qh, kh, vh = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2), dim=-1)
    o = (p @ vh).transpose(1, 2).reshape(B, T, D)
    do = torch.randn(B, T, D, dtype=torch.float64)
    o.backward(do)
    lse_ref = torch.logsumexp(qh @ kh.transpose(-1, -2), dim=-1)
    qd, kd, vd, dod = (t.view(-1, D) for t in (g(q.detach().float()), g(k.detach().float()), g(v.detach().float()), g(do.float())))
    od = torch.full((B * T, D), 7.0, device=DEV)
    lse = torch.zeros(B * H, T, device=DEV)
    hip.attention(qd, kd, vd, od, lse, B, H, T, S, compute=0)
    close(od.view(B, T, D), o, rtol=2e-5, what="attention fwd")
    close(lse.view(B, H, T), lse_ref, rtol=1e-5, what="attention lse")
    dq, dk, dv = (torch.zeros_like(t).view(B, -1, D) for t in (qd, kd, vd))
    delta = torch.zeros(B * H, T, device=DEV)
    hip.attention(qd, kd, vd, od, lse, B, H, T, S, compute=0, d_o=dod, dq=dq.view(-1, D), dk=dk.view(-1, D), dv=dv.view(-1, D),
                  delta=delta)
    close(dq, q.grad, rtol=5e-5, what="attention dq")
    close(dk, k.grad, rtol=5e-5, what="attention dk")
    close(dv, v.grad, rtol=5e-5, what="attention dv")


@pytest.mark.parametrize("B,T,S,p", [(2, 1050, 1050, 0.0), (2, 100, 1050, 0.1), (3, 100, 100, 0.0), (1, 37, 5, 0.0),
                                     (1, 300, 1344, 0.0), (2, 70, 333, 0.1)])
def test_fused_attention_bf16_mfma(hip, B, T, S, p):
    """bf16-MFMA attention core (precision="bf16"): Q, K, V, P, dO, dS rounded to bf16, everything else fp32.  Checked
    against the fp64 reference (with the hash dropout masks of the oracle when p > 0) at bf16-level tolerances; the
    transposed-fragment paths (ds_read_b64_tr_b16) would be off by O(1) if a row / column mapping were wrong."""
    from oracle import dropout_ref as DR
    torch.manual_seed(B * 77 + T + S)
    H, hd, seed = 8, 32, 991
    D = H * hd
    q = (torch.randn(B, T, D, dtype=torch.float64) * 0.6).requires_grad_(True)
    k = torch.randn(B, S, D, dtype=torch.float64).requires_grad_(True)
    v = torch.randn(B, S, D, dtype=torch.float64).requires_grad_(True)
    qh, kh, vh = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    sc = qh @ kh.transpose(-1, -2)
    w = torch.softmax(sc, dim=-1)
    if p > 0.0:
        keep = torch.from_numpy(DR.keep_mask(DR.drop_key(seed, 0), DR.attn_index(B * H, T, S).reshape(B, H, T, S), p))
        w = torch.where(keep, w / (1.0 - p), torch.zeros_like(w))
    o = (w @ vh).transpose(1, 2).reshape(B, T, D)
    do = torch.randn(B, T, D, dtype=torch.float64)
    o.backward(do)
    qd, kd, vd, dod = (t.view(-1, D) for t in (g(q.detach().float()), g(k.detach().float()), g(v.detach().float()), g(do.float())))
    od, lse = torch.full((B * T, D), 7.0, device=DEV), torch.zeros(B * H, T, device=DEV)
    hip.attention(qd, kd, vd, od, lse, B, H, T, S, compute=1, dropout_p=p, dropout_site=seed)
    close(od.view(B, T, D), o, rtol=1.5e-2, what="bf16 attention fwd")
    close(lse.view(B, H, T), torch.logsumexp(sc, dim=-1), rtol=3e-3, what="bf16 attention lse")
    dq, dk, dv = (torch.zeros_like(t).view(B, -1, D) for t in (qd, kd, vd))
    delta = torch.zeros(B * H, T, device=DEV)
    hip.attention(qd, kd, vd, od, lse, B, H, T, S, compute=1, dropout_p=p, dropout_site=seed, d_o=dod, dq=dq.view(-1, D),
                  dk=dk.view(-1, D), dv=dv.view(-1, D), delta=delta)
    close(dq, q.grad, rtol=2.5e-2, what="bf16 attention dq")
    close(dk, k.grad, rtol=2.5e-2, what="bf16 attention dk")
    close(dv, v.grad, rtol=2.5e-2, what="bf16 attention dv")


@pytest.mark.parametrize("N,H,W,compute", [(2, 37, 53, 0), (1, 128, 160, 0), (2, 37, 53, 1), (3, 64, 96, 1)])
def test_stem_conv_implicit_gemm(hip, N, H, W, compute):
    """7x7/s2 stem convolution as an implicit GEMM (no im2col buffer): forward with folded BN + ReLU and the weight
    gradient (split over the output pixels, deterministic) vs torch conv2d on the zero-padded image."""
    torch.manual_seed(N + H + W + compute)
    hip.ensure_workspace(DEV)                        # the split reduction goes through the shared workspace
    rnd = (lambda *sh: _bf(torch.randn(*sh))) if compute else (lambda *sh: torch.randn(*sh, dtype=torch.float64))
    img = rnd(N, H, W, 3)
    w = (rnd(7, 7, 3, 64) / 12.0).requires_grad_(True)
    scale, shift = torch.rand(64, dtype=torch.float64) + 0.5, torch.randn(64, dtype=torch.float64)
    z = F.conv2d(F.pad(img.permute(0, 3, 1, 2), (3, 3, 3, 3)), w.permute(3, 2, 0, 1), None, stride=2).permute(0, 2, 3, 1)
    Ho, Wo = z.shape[1], z.shape[2]
    y = torch.relu(z * scale + shift)
    imgd, sd = g(img.float()), g(scale.float())
    ws = g((w.detach() * scale).float().reshape(147, 64))          # BN scale folded into the kernel, as the engine does
    yd = torch.zeros(N, Ho, Wo, 64, device=DEV)
    hip.stem_conv(0, imgd, ws, yd, N, H, W, Ho, Wo, bias=g(shift.float()), act=1, compute=compute)
    close(yd, y, rtol=(3e-3 if compute else 2e-5), what="stem conv fwd")     # bf16: w*scale is re-rounded by the kernel
    dz = rnd(N, Ho, Wo, 64)
    (z * dz).sum().backward()
    dzd = g(dz.float())
    res = []
    for rep in range(2):
        dw = torch.zeros(147, 64, device=DEV)
        hip.stem_conv(2, imgd, dzd, dw, N, H, W, Ho, Wo, scale=sd, split=7, compute=compute)
        res.append(dw)
    close(res[0].view(7, 7, 3, 64), w.grad * scale, rtol=5e-5, what="stem conv wgrad")
    assert torch.equal(res[0], res[1])
    dw1 = torch.zeros(147, 64, device=DEV)
    hip.stem_conv(2, imgd, dzd, dw1, N, H, W, Ho, Wo, scale=sd, split=1, compute=compute)
    close(dw1.view(7, 7, 3, 64), w.grad * scale, rtol=5e-5, what="stem conv wgrad (unsplit)")


@pytest.mark.parametrize("N,H,W", [(2, 37, 53), (1, 64, 300), (3, 23, 131), (1, 9, 8)])
def test_stem_conv_row_staged_forward(hip, monkeypatch, N, H, W):
    """The row-staging stem forward (bf16 output: persistent workgroups, kernel tensor resident in LDS, 7 input rows of a
    64-pixel output run staged once, kernel rows padded 21 -> 24) against fp64 on the same bf16-rounded operands (one bf16
    rounding of the output) and against the gathering kernel (DETR_HIP_STEM_ROWS=2), which sums the same products in
    another order: <= 1 bf16 ulp apart on a small fraction of the outputs.  Shapes: ragged last tile, several tiles per row,
    several images, an image narrower than one tile."""
    torch.manual_seed(N + H + W)
    img = _bf(torch.randn(N, H, W, 3))
    w = _bf(torch.randn(7, 7, 3, 64) / 12.0)
    shift = torch.randn(64, dtype=torch.float64)
    z = F.conv2d(F.pad(img.permute(0, 3, 1, 2), (3, 3, 3, 3)), w.permute(3, 2, 0, 1), None, stride=2).permute(0, 2, 3, 1)
    Ho, Wo = z.shape[1], z.shape[2]
    ref = torch.relu(z + shift)
    imgd, ws, sd = g(img.float()), g(w.float().reshape(147, 64)), g(shift.float())
    outs = {}
    for mode in ("0", "2"):
        hip.set_tuning("DETR_HIP_STEM_ROWS", mode)
        y = torch.full((N, Ho, Wo, 64), 7.0, device=DEV, dtype=torch.bfloat16)
        hip.stem_conv(0, imgd, ws, y, N, H, W, Ho, Wo, bias=sd, act=1, compute=1)
        torch.cuda.synchronize()
        outs[mode] = y.float().cpu().double()
    hip.set_tuning("DETR_HIP_STEM_ROWS", None)
    rows, gather = outs["0"], outs["2"]
    scale = float(ref.abs().max())
    err = (rows - ref).abs()
    assert float((err / (ref.abs() + 1e-2 * scale)).max()) < 2.0 ** -8 * 1.1, "row-staged stem: more than one bf16 rounding from fp64"
    diff = (rows - gather).abs()
    assert float((diff > 0).double().mean()) < 5e-3, float((diff > 0).double().mean())
    assert float((diff / (gather.abs() + 1e-2 * scale)).max()) <= 2.0 ** -7
    # weight gradient with a bf16 dy: the row-staging kernel (all 147 kernel rows per workgroup, persistent over units of 32
    # pixels, per-workgroup slabs) against fp64 and against the gathering kernel
    hip.ensure_workspace(DEV)
    dy = _bf(torch.randn(N, Ho, Wo, 64))
    bn = torch.rand(64, dtype=torch.float64) + 0.5
    wref = torch.einsum("nhwk,nhwc->kc", F.unfold(F.pad(img.permute(0, 3, 1, 2), (3, 3, 3, 3)), 7, stride=2)
                        .view(N, 3, 49, Ho, Wo).permute(0, 3, 4, 2, 1).reshape(N, Ho, Wo, 147), dy) * bn
    dyd, bnd = g(dy.float()).to(torch.bfloat16), g(bn.float())
    res = {}
    for mode in ("0", "2"):
        hip.set_tuning("DETR_HIP_STEM_ROWS", mode)
        pair = []
        for rep in range(2):
            dw = torch.zeros(147, 64, device=DEV)
            hip.stem_conv(2, imgd, dyd, dw, N, H, W, Ho, Wo, scale=bnd, split=7, compute=1)
            pair.append(dw)
        assert torch.equal(pair[0], pair[1]), "stem weight gradient is not deterministic"
        res[mode] = pair[0]
    hip.set_tuning("DETR_HIP_STEM_ROWS", None)
    close(res["0"], wref, rtol=3e-3, what="stem wgrad (staged rows, bf16 operands)")
    close(res["2"], wref, rtol=3e-3, what="stem wgrad (gathering kernel, bf16 operands)")
    close(res["0"], res["2"].double().cpu(), rtol=1e-4, what="stem wgrad: staged rows vs gathering kernel")


def test_linear_helpers(hip):
    torch.manual_seed(5)
    M, K, N = 420, 256, 92
    x = torch.randn(M, K, dtype=torch.float64, requires_grad=True)
    W = torch.randn(N, K, dtype=torch.float64, requires_grad=True)
    b = torch.randn(N, dtype=torch.float64)
    y = torch.relu(x @ W.t() + b)
    dy = torch.randn(M, N, dtype=torch.float64)
    dz = dy * (y > 0)
    y.backward(dy)
    xd, Wd, bd = g(x.detach().float()), g(W.detach().float()), g(b.float())
    yd = torch.zeros(M, N, device=DEV)
    hip.linear_fwd(xd, Wd, bd, yd, act=1)
    close(yd, y, what="linear fwd")
    dzd = g(dz.float())
    dxd = torch.zeros(M, K, device=DEV)
    hip.linear_dgrad(dzd, Wd, dxd)
    close(dxd, x.grad, what="linear dgrad")
    dWd = torch.zeros(N, K, device=DEV)
    hip.linear_wgrad(dzd, xd, dWd)
    close(dWd, W.grad, rtol=5e-5, what="linear wgrad")


# ------------------------------------------------------------------------------------------
# conv 3x3 / stem / pooling
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,Ci,Co,stride", [(2, 13, 17, 64, 64, 1), (1, 20, 27, 128, 128, 2), (2, 9, 11, 16, 32, 1),
                                                (1, 50, 67, 64, 64, 1), (1, 25, 42, 256, 256, 2)])
def test_conv3x3_all_modes(hip, N, H, W, Ci, Co, stride):
    torch.manual_seed(N + H + W + Ci + stride)
    x = torch.randn(N, H, W, Ci, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(3, 3, Ci, Co, dtype=torch.float64) / (3 * Ci ** 0.5)).requires_grad_(True)
    scale, shift = torch.rand(Co, dtype=torch.float64) + 0.5, torch.randn(Co, dtype=torch.float64)
    z = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), None, stride=stride, padding=1).permute(0, 2, 3, 1)
    Ho, Wo = z.shape[1], z.shape[2]
    y = torch.relu(z * scale + shift)
    xd, wd = g(x.detach().float()), g(w.detach().float())
    yd = torch.zeros(N, Ho, Wo, Co, device=DEV)
    hip.conv3x3(0, xd, wd, yd, N, H, W, Ci, Ho, Wo, Co, stride, scale=g(scale.float()), bias=g(shift.float()), act=1)
    close(yd, y, what="conv3x3 fwd")
    # backward of z (pre-BN) given dz
    dz = torch.randn_like(z)
    z.backward(dz)
    dzd = g(dz.float())
    mask = torch.randn(N, H, W, Ci)
    dxd = torch.zeros(N, H, W, Ci, device=DEV)
    hip.conv3x3(1, dzd, wd, dxd, N, H, W, Ci, Ho, Wo, Co, stride, mask=g(mask))
    close(dxd, x.grad * (mask.double() > 0), what="conv3x3 dgrad(+mask)")
    dwd = torch.zeros(3, 3, Ci, Co, device=DEV)
    hip.conv3x3(2, xd, dzd, dwd, N, H, W, Ci, Ho, Wo, Co, stride, scale=g(scale.float()))
    close(dwd, w.grad * scale, rtol=5e-5, what="conv3x3 wgrad(+scale)")


def test_maxpool_fp32_fwd_bwd(hip):
    torch.manual_seed(11)
    N, Ho, Wo = 2, 19, 25
    y = g(torch.relu(torch.randn(N, Ho, Wo, 64)))
    # max pool over the zero padded map
    x = y.view(N, Ho, Wo, 64)
    Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
    xc = x.cpu().double().requires_grad_(True)
    pref = F.max_pool2d(F.pad(xc.permute(0, 3, 1, 2), (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1)
    pool = torch.zeros(N, Hp, Wp, 64, device=DEV)
    amax = torch.zeros(N, Hp, Wp, 64, device=DEV, dtype=torch.uint8)
    hip.call("detr_hip_maxpool3x3s2_fwd_f32", x.data_ptr(), pool.data_ptr(), amax.data_ptr(), N, Ho, Wo, 64, Hp, Wp)
    close(pool, pref, atol=0.0, what="maxpool fwd")
    dy = torch.randn(N, Hp, Wp, 64)
    pref.backward(dy.double())
    dx = torch.zeros(N, Ho, Wo, 64, device=DEV)
    dyd = g(dy)
    hip.call("detr_hip_maxpool3x3s2_bwd_f32", dyd.data_ptr(), amax.data_ptr(), x.data_ptr(), dx.data_ptr(), N, Ho, Wo,
             64, Hp, Wp)
    # gradients at x == 0 are dropped by the fused ReLU mask; torch routes ties at 0 arbitrarily
    refdx = xc.grad * (xc.detach() > 0)
    close(dx, refdx, atol=1e-6, what="maxpool bwd (+relu mask)")


def test_subsample2(hip):
    x = torch.randn(2, 9, 13, 16)
    Ho, Wo = 5, 7
    y = torch.zeros(2, Ho, Wo, 16, device=DEV)
    xd = g(x)
    hip.call("detr_hip_subsample2_fwd_f32", xd.data_ptr(), y.data_ptr(), 2, 9, 13, 16, Ho, Wo)
    assert torch.equal(y.cpu(), x[:, ::2, ::2, :])
    dy = torch.randn(2, Ho, Wo, 16)
    dx = torch.full((2, 9, 13, 16), 3.0, device=DEV)
    dyd = g(dy)
    hip.call("detr_hip_subsample2_bwd_f32", dyd.data_ptr(), dx.data_ptr(), 2, 9, 13, 16, Ho, Wo)
    ref = torch.zeros(2, 9, 13, 16)
    ref[:, ::2, ::2, :] = dy
    assert torch.equal(dx.cpu(), ref)


# ------------------------------------------------------------------------------------------
# row kernels
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,C", [(8400, 256), (37, 256), (5, 64), (130, 1024)])
def test_layernorm_fwd_bwd(hip, rows, C):
    torch.manual_seed(rows + C)
    x = (torch.randn(rows, C, dtype=torch.float64) * 2 + 0.3).requires_grad_(True)
    gam = (torch.rand(C, dtype=torch.float64) + 0.5).requires_grad_(True)
    bet = torch.randn(C, dtype=torch.float64).requires_grad_(True)
    y = F.layer_norm(x, (C,), gam, bet, 1e-5)
    dy = torch.randn(rows, C, dtype=torch.float64)
    y.backward(dy)
    xd, gd, bd = g(x.detach().float()), g(gam.detach().float()), g(bet.detach().float())
    yd = torch.zeros(rows, C, device=DEV)
    mean, rstd = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
    hip.layernorm_fwd(xd, gd, bd, yd, mean, rstd, 1e-5)
    close(yd, y, rtol=1e-5, what="layernorm fwd")
    dxd = torch.zeros(rows, C, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dyd = g(dy.float())
    ws_keep, hip.WORKSPACE = hip.WORKSPACE, None
    try:
        hip.layernorm_bwd(dyd, xd, gd, mean, rstd, dxd, dg, db)                     # no workspace: atomic fallback
    finally:
        hip.WORKSPACE = ws_keep
    close(dxd, x.grad, rtol=2e-5, what="layernorm dx")
    close(dg, gam.grad, rtol=5e-5, what="layernorm dgamma")
    close(db, bet.grad, rtol=5e-5, what="layernorm dbeta")
    hip.ensure_workspace(DEV)                                                     # deterministic workspace path
    res = []
    for rep in range(2):
        dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        hip.layernorm_bwd(dyd, xd, gd, mean, rstd, dxd, dg2, db2)
        res.append((dg2, db2))
    close(res[0][0], gam.grad, rtol=5e-5, what="layernorm dgamma (workspace)")
    close(res[0][1], bet.grad, rtol=5e-5, what="layernorm dbeta (workspace)")
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # dy_add (round 4): two gradient branches summed inside the launch == the launch on their sum
    part = g(torch.randn(rows, C))
    rest = dyd - part
    dx2, dg3, db3 = torch.zeros(rows, C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    hip.layernorm_bwd(rest, xd, gd, mean, rstd, dx2, dg3, db3, dy_add=part)
    close(dx2, x.grad, rtol=4e-5, what="layernorm dx (dy + dy_add)")
    close(dg3, gam.grad, rtol=1e-4, what="layernorm dgamma (dy + dy_add)")
    close(db3, bet.grad, rtol=1e-4, what="layernorm dbeta (dy + dy_add)")
    # queued finish (detr_layernorm_desc.defer_blocks_out + detr_hip_splitk_reduce_many): same partials, same summation order
    hip.begin_deferred_reduces(DEV)
    try:
        dg3, db3 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        hip.layernorm_bwd(dyd, xd, gd, mean, rstd, dxd, dg3, db3)
        assert len(hip.DEFER) == 2 and float(dg3.abs().max()) == 0.0, "the gamma / beta finish ran early"
    finally:
        hip.flush_reduces(end=True)
    if rows >= 64:       # >= 8 partial blocks: the grouped reduction sums them in the finish kernel's order
        assert torch.equal(dg3, res[0][0]) and torch.equal(db3, res[0][1])
    close(dg3, gam.grad, rtol=5e-5, what="layernorm dgamma (queued finish)")
    close(db3, bet.grad, rtol=5e-5, what="layernorm dbeta (queued finish)")


def test_small_elementwise(hip):
    torch.manual_seed(9)
    # NOTE: every device tensor is bound to a local before its pointer is taken (a temporary would be
    # freed -- and its block reused by the next allocation -- before the kernel runs)
    x = torch.randn(8400, 300)
    xd = g(x)
    out = torch.zeros(300, device=DEV)
    hip.call("detr_hip_colsum_f32", xd.data_ptr(), out.data_ptr(), 8400, 300, 300, ctypes.c_float(0.5))
    close(out, 0.5 * x.double().sum(0), rtol=2e-5, what="colsum")
    # the deterministic form (round 5): fixed summation order -> two runs give the same bits, whatever the row count
    n = int(hip.load().detr_hip_colsum_det_scratch_floats(8400, 300))
    scr = torch.empty(n, device=DEV)
    outs = []
    for _ in range(3):
        o2 = torch.full((300,), 0.25, device=DEV)
        hip.call("detr_hip_colsum_det_f32", xd.data_ptr(), o2.data_ptr(), 8400, 300, 300, ctypes.c_float(0.5), scr.data_ptr(), n)
        outs.append(o2.cpu())
    close(outs[0], 0.25 + 0.5 * x.double().sum(0), rtol=2e-5, what="colsum_det")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    a, p = torch.randn(4, 50, 256), torch.randn(50, 256)
    ad, pd = g(a), g(p)
    o = torch.zeros(4, 50, 256, device=DEV)
    hip.call("detr_hip_add_bcast_f32", ad.data_ptr(), pd.data_ptr(), o.data_ptr(), a.numel(), p.numel())
    assert torch.equal(o.cpu(), a + p)
    b = torch.randn(4, 50, 256)
    bd_ = g(b)
    hip.call("detr_hip_add_f32", ad.data_ptr(), bd_.data_ptr(), o.data_ptr(), a.numel())
    assert torch.equal(o.cpu(), a + b)
    y = torch.rand(1000)
    dy = torch.randn(1000)
    yd, dyd, ymd = g(y), g(dy), g(y - 0.5)
    dz = torch.zeros(1000, device=DEV)
    hip.call("detr_hip_sigmoid_bwd_f32", dyd.data_ptr(), yd.data_ptr(), dz.data_ptr(), 1000)
    close(dz, dy * y * (1 - y), rtol=1e-6, what="sigmoid bwd")
    hip.call("detr_hip_relu_mask_f32", dyd.data_ptr(), ymd.data_ptr(), dz.data_ptr(), 1000)
    assert torch.equal(dz.cpu(), torch.where(y - 0.5 > 0, dy, torch.zeros(())))
    w, sc = torch.randn(576, 64), torch.rand(64)
    wd_, scd = g(w), g(sc)
    wo = torch.zeros(576, 64, device=DEV)
    hip.call("detr_hip_scale_cols_f32", wd_.data_ptr(), scd.data_ptr(), wo.data_ptr(), 576, 64)
    assert torch.equal(wo.cpu(), w * sc)
    bw, bb, bm, bv = torch.rand(64) + .5, torch.randn(64), torch.randn(64), torch.rand(64) + .5
    bwd_, bbd, bmd, bvd = g(bw), g(bb), g(bm), g(bv)
    s_, h_ = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    hip.call("detr_hip_bn_fold_f32", bwd_.data_ptr(), bbd.data_ptr(), bmd.data_ptr(), bvd.data_ptr(), s_.data_ptr(),
             h_.data_ptr(), 64, ctypes.c_float(1e-5))
    rs = bw * torch.rsqrt(bv + 1e-5)
    close(s_, rs, rtol=1e-6, what="bn scale")
    close(h_, bb - bm * rs, rtol=1e-6, what="bn shift")
    acc, gg = torch.randn(999), torch.randn(999)
    accd, ggd = g(acc), g(gg)
    hip.call("detr_hip_axpy_f32", accd.data_ptr(), ggd.data_ptr(), ctypes.c_float(2.0), 999)
    close(accd, acc + 2 * gg, rtol=1e-6, what="axpy")
    z = torch.ones(1001, device=DEV)
    hip.zero_(z)
    assert float(z.abs().sum()) == 0.0


# ------------------------------------------------------------------------------------------
# set loss: cost, assignment (vs SciPy), loss + gradients (vs oracle autograd)
# ------------------------------------------------------------------------------------------
def _setloss_desc(hip, logits, boxes, t_bbox, t_class, bg):
    """logits [Lv,B,Q,C], boxes [Lv,B,Q,4] contiguous CUDA tensors."""
    Lv, B, Q, C = logits.shape
    d = hip.SetLossDesc()
    d.levels, d.B, d.Q, d.C, d.R = Lv, B, Q, C, t_bbox.shape[1]
    d.logits, d.sL_l, d.sL_b, d.sL_q = logits.data_ptr(), B * Q * C, Q * C, C
    d.boxes, d.sB_l, d.sB_b, d.sB_q = boxes.data_ptr(), B * Q * 4, Q * 4, 4
    d.t_bbox, d.t_class = t_bbox.data_ptr(), t_class.data_ptr()
    d.background_class = bg
    return d


@pytest.mark.parametrize("Q,n_list", [(100, [7, 1, 99, 0, 20]), (300, [99, 5]), (64, [64, 3])])
def test_assign_matches_scipy(hip, Q, n_list):
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(Q)
    R = 100
    B = len(n_list)
    Lv = 3
    P = Lv * B
    ldc = R - 1
    t_bbox = np.zeros((B, R, 4), np.float32)
    for b, n in enumerate(n_list):
        t_bbox[b, 0, 0] = n
    cost = rng.normal(size=(P, Q, ldc)).astype(np.float32)
    cd, tbd = g(torch.tensor(cost)), g(torch.tensor(t_bbox))
    tfp = torch.full((P, Q), -7, device=DEV, dtype=torch.int32)
    pft = torch.full((P, ldc), -7, device=DEV, dtype=torch.int32)
    st = torch.full((P,), -7, device=DEV, dtype=torch.int32)
    hip.call("detr_hip_assign_f32", cd.data_ptr(), P, Q, ldc, tbd.data_ptr(), B, R, tfp.data_ptr(), pft.data_ptr(),
             st.data_ptr())
    torch.cuda.synchronize()
    tfp, pft, st = tfp.cpu().numpy(), pft.cpu().numpy(), st.cpu().numpy()
    assert (st == 0).all()
    for p in range(P):
        n = n_list[p % B]
        rows, cols = linear_sum_assignment(cost[p, :, :n])
        got = pft[p, :n]
        exp = np.full(n, -1)
        exp[cols] = rows
        assert np.array_equal(got, exp), f"problem {p} (Q={Q}, n={n}) differs from SciPy"
        assert (pft[p, n:] == -1).all()
        inv = np.full(Q, -1)
        inv[rows] = cols
        assert np.array_equal(tfp[p], inv)


@pytest.mark.parametrize("Q,n,seed", [(100, 7, 0), (100, 99, 1), (100, 1, 2), (64, 20, 3)])
def test_hungarian_matching_six_tuple_vs_scipy(hip, Q, n, seed):
    """The single-image wrapper with the reference's signature and SIX-tuple return convention
    (hungarian_matching.py:163-203).  The reference's names are swapped twice (np_tf_linear_sum_assignment calls SciPy's
    ROW = prediction indices `target_indices`, :29-31, and hungarian_matching returns `pred_indices` first, :203), so the
    caller's `t_indices, p_indices, t_selector, p_selector` (loss.py:118) receive: SciPy's column (= target) indices,
    SciPy's row (= prediction) indices ascending, a bool[n] over targets, a bool[Q] over predictions."""
    from scipy.optimize import linear_sum_assignment
    from detr_tf.loss.hungarian_matching import hungarian_matching
    from oracle import set_loss_ref as L
    rng = np.random.default_rng(seed)
    C = 92
    t_bbox = np.zeros((100, 4), np.float32)
    t_class = np.zeros((100, 1), np.int64)
    t_bbox[0, 0] = n
    t_bbox[1:1 + n, :2] = rng.uniform(0.2, 0.8, (n, 2))
    t_bbox[1:1 + n, 2:] = rng.uniform(0.05, 0.4, (n, 2))
    t_class[1:1 + n, 0] = rng.integers(1, 91, n)
    p_bbox = np.concatenate([rng.uniform(0.1, 0.9, (Q, 2)), rng.uniform(0.05, 0.5, (Q, 2))], 1).astype(np.float32)
    p_class = rng.normal(size=(Q, C)).astype(np.float32)
    got = hungarian_matching(g(torch.from_numpy(t_bbox)), g(torch.from_numpy(t_class)), g(torch.from_numpy(p_bbox)),
                             g(torch.from_numpy(p_class)), slice_preds=True)
    assert len(got) == 6
    t_idx, p_idx, t_sel, p_sel, tb, tc = got
    cost = L.cost_matrix(torch.from_numpy(t_bbox[1:1 + n]), torch.from_numpy(t_class[1:1 + n, 0]), torch.from_numpy(p_bbox),
                         torch.from_numpy(p_class)).numpy()
    rows, cols = linear_sum_assignment(cost)                       # rows = predictions (ascending), cols = targets
    assert t_idx.dtype == torch.int64 and p_idx.dtype == torch.int64
    assert np.array_equal(p_idx.cpu().numpy(), rows) and np.array_equal(t_idx.cpu().numpy(), cols)
    assert t_sel.dtype == torch.bool and tuple(t_sel.shape) == (n,) and bool(t_sel.all())
    want_psel = np.zeros(Q, bool)
    want_psel[rows] = True
    assert p_sel.dtype == torch.bool and np.array_equal(p_sel.cpu().numpy(), want_psel)
    assert np.array_equal(tb.cpu().numpy(), t_bbox[1:1 + n]) and np.array_equal(tc.cpu().numpy(), t_class[1:1 + n, 0])
    # the oracle's own wrapper agrees (it returns the five used entries in the caller's order)
    oti, opi, osel, _, _ = L.hungarian_matching(torch.from_numpy(t_bbox), torch.from_numpy(t_class), torch.from_numpy(p_bbox),
                                                torch.from_numpy(p_class))
    assert np.array_equal(oti.numpy(), cols) and np.array_equal(opi.numpy(), rows) and np.array_equal(osel.numpy(), want_psel)


def test_assign_ties_and_invalid(hip):
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(5)
    Q, R, B, P = 100, 100, 4, 4
    ldc = R - 1
    t_bbox = np.zeros((B, R, 4), np.float32)
    t_bbox[:, 0, 0] = [10, 30, 5, 8]
    cost = np.round(rng.normal(size=(P, Q, ldc)) * 2).astype(np.float32) / 2     # heavy ties
    cost[2, 3, 1] = np.nan
    cost[3, :, 2] = np.inf                                                        # infeasible column
    tfp = torch.zeros((P, Q), device=DEV, dtype=torch.int32)
    pft = torch.zeros((P, ldc), device=DEV, dtype=torch.int32)
    st = torch.zeros((P,), device=DEV, dtype=torch.int32)
    cd, tbd = g(torch.tensor(cost)), g(torch.tensor(t_bbox))
    hip.call("detr_hip_assign_f32", cd.data_ptr(), P, Q, ldc, tbd.data_ptr(), B, R, tfp.data_ptr(), pft.data_ptr(),
             st.data_ptr())
    st, pft = st.cpu().numpy(), pft.cpu().numpy()
    assert st[0] == 0 and st[1] == 0 and st[2] == 1 and st[3] == 1      # SciPy raises ValueError for both
    for p in (0, 1):
        n = int(t_bbox[p, 0, 0])
        rows, cols = linear_sum_assignment(cost[p, :, :n])
        got = pft[p, :n]
        assert len(set(got.tolist())) == n and (got >= 0).all()
        assert abs(cost[p, got, np.arange(n)].sum() - cost[p, rows, cols].sum()) < 1e-6   # same optimum under ties


@pytest.mark.parametrize("Q,B,bg", [(100, 4, 91), (300, 2, 91)])
def test_set_loss_vs_oracle(hip, Q, B, bg):
    from oracle import set_loss_ref as L
    torch.manual_seed(Q + B)
    Lv, C = 6, 92
    logits = torch.randn(Lv, B, Q, C) * 2
    boxes = torch.rand(Lv, B, Q, 4) * torch.tensor([1.0, 1.0, 0.6, 0.6]) + torch.tensor([0.0, 0.0, 0.02, 0.02])
    boxes[0, 0, :5, 2] = 1.9   # boxes clipped on both sides: exercises the clip gradient
    tb, tc = L.make_targets(B, seed=77)
    tbt, tct = torch.tensor(tb), torch.tensor(tc)
    # oracle (fp64 for the gradient reference, fp32 for the cost matrix)
    lg64 = logits.double().requires_grad_(True)
    bx64 = boxes.double().requires_grad_(True)
    m_out = {"pred_logits": lg64[Lv - 1], "pred_boxes": bx64[Lv - 1],
             "aux": [{"pred_logits": lg64[i], "pred_boxes": bx64[i]} for i in range(Lv - 1)]}
    total, losses = L.get_losses(m_out, tbt.double(), tct, bg)
    total.backward()
    lgd, bxd, tbd, tcd = g(logits), g(boxes), g(tbt), g(tct.reshape(B, -1))
    d = _setloss_desc(hip, lgd, bxd, tbd, tcd, bg)
    P, R = Lv * B, tb.shape[1]
    cost = torch.zeros(P, Q, R - 1, device=DEV)
    hip.call("detr_hip_match_cost_f32", ctypes.byref(d), cost.data_ptr())
    for lv in range(Lv):
        for b in range(B):
            tbs, tcs = L.strip_header(tbt[b], tct[b])
            cref = L.cost_matrix(tbs, tcs, boxes[lv, b], logits[lv, b])
            close(cost[lv * B + b, :, :tbs.shape[0]], cref, atol=2e-6, what=f"cost lv{lv} b{b}")
    tfp = torch.zeros((P, Q), device=DEV, dtype=torch.int32)
    pft = torch.zeros((P, R - 1), device=DEV, dtype=torch.int32)
    st = torch.zeros((P,), device=DEV, dtype=torch.int32)
    hip.call("detr_hip_assign_f32", cost.data_ptr(), P, Q, R - 1, tbd.data_ptr(), B, R, tfp.data_ptr(), pft.data_ptr(),
             st.data_ptr())
    assert int(st.abs().sum()) == 0
    sums = torch.zeros(Lv * 10, device=DEV)
    hip.call("detr_hip_set_loss_sums_f32", ctypes.byref(d), tfp.data_ptr(), sums.data_ptr())
    out = torch.zeros(Lv, 6, device=DEV)
    tot = torch.zeros(1, device=DEV)
    hip.call("detr_hip_set_loss_finalize_f32", sums.data_ptr(), Lv, out.data_ptr(), tot.data_ptr())
    names = ["label_cost", "true_neg", "true_pos", "pos_accuracy", "giou_loss", "l1_loss"]
    out = out.cpu()
    for lv in range(Lv):
        suf = "" if lv == Lv - 1 else f"_{lv}"
        for k, nm in enumerate(names):
            ref = float(losses[nm + suf])
            assert abs(float(out[lv, k]) - ref) <= 2e-5 * max(1.0, abs(ref)), (lv, nm, float(out[lv, k]), ref)
    assert abs(float(tot) - float(total)) <= 1e-5 * abs(float(total))      # loss matches the oracle to 1e-5 rel
    dl, dbx = torch.zeros_like(lgd), torch.zeros_like(bxd)
    hip.call("detr_hip_set_loss_grad_f32", ctypes.byref(d), tfp.data_ptr(), sums.data_ptr(), ctypes.c_float(0.5),
             dl.data_ptr(), dbx.data_ptr())
    close(dl, 0.5 * lg64.grad, rtol=2e-5, what="d_logits")
    close(dbx, 0.5 * bx64.grad, rtol=5e-5, what="d_boxes")


# ------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------
def test_clip_adam_vs_oracle(hip):
    from oracle import optim_ref as O
    rng = np.random.default_rng(3)
    shapes = [(64, 3, 7), (5,), (300, 40), (1,), (4097,)]
    groups = [0, 0, 1, 2, 1]
    params = {i: rng.normal(size=s).astype(np.float32) for i, s in enumerate(shapes)}
    sizes = [int(np.prod(s)) for s in shapes]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    chunk = 1024
    ct, cs = [], []
    for t, (o, n) in enumerate(zip(offs[:-1], sizes)):
        for st in range(0, n, chunk):
            ct.append(t)
            cs.append(o + st)
    flat = g(torch.tensor(np.concatenate([params[i].ravel() for i in range(len(shapes))])))
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    ctd = g(torch.tensor(ct, dtype=torch.int32))
    csd = g(torch.tensor(cs, dtype=torch.int64))
    sed = g(torch.tensor(offs[1:], dtype=torch.int64))
    tgd = g(torch.tensor(groups, dtype=torch.int32))
    lrs = [1e-2, 3e-3, 1e-3]
    opts = [O.Adam(lr, clipnorm=0.1) for lr in lrs]
    import math
    for step in range(1, 4):
        scale = [1e-3, 10.0, 1.0][step - 1]    # below / above the clip threshold
        grads = {i: (rng.normal(size=s) * scale).astype(np.float32) for i, s in enumerate(shapes)}
        gflat = g(torch.tensor(np.concatenate([grads[i].ravel() for i in range(len(shapes))])))
        sumsq = torch.zeros(len(ct), device=DEV)              # per-chunk partials; the per-tensor sum is formed in a fixed order
        hip.call("detr_hip_sumsq_segments_f32", gflat.data_ptr(), ctd.data_ptr(), csd.data_ptr(), sed.data_ptr(), len(ct),
                 chunk, sumsq.data_ptr())
        hyper = torch.tensor([lr * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step) for lr in lrs] + [0.1, 0.9, 0.999, 1e-7, 0],
                             dtype=torch.float32, device=DEV)
        hip.call("detr_hip_clip_adam_f32", flat.data_ptr(), gflat.data_ptr(), m.data_ptr(), v.data_ptr(), ctd.data_ptr(),
                 csd.data_ptr(), sed.data_ptr(), tgd.data_ptr(), sumsq.data_ptr(), hyper.data_ptr(), len(ct), chunk)
        for gi in range(3):
            opts[gi].apply({i: grads[i] for i in range(len(shapes)) if groups[i] == gi}, params)
        ref = np.concatenate([params[i].ravel() for i in range(len(shapes))])
        close(flat, torch.tensor(ref), rtol=1e-5, what=f"adam step {step}")


# ------------------------------------------------------------------------------------------
# dropout (training mode): device masks == the numpy restatement of the counter hash
# ------------------------------------------------------------------------------------------
def test_dropout_gemm_epilogue_and_elementwise(hip):
    from oracle import dropout_ref as DR
    torch.manual_seed(21)
    M, N, K, p, seed = 333, 256, 64, 0.1, 0xABCDE
    x, W, b, R = torch.randn(M, K), torch.randn(N, K), torch.randn(N), torch.randn(M, N)
    xd, Wd, bd, Rd = g(x), g(W), g(b), g(R)
    step = 0x1234567
    stepd = torch.tensor([step] + [0] * 7, dtype=torch.int32, device=DEV)         # the per-step seed lives in device memory
    keep = torch.from_numpy(DR.keep_mask(DR.drop_key(seed, step), np.arange(M * N).reshape(M, N), p))
    assert 0.88 < float(keep.float().mean()) < 0.92
    keep0 = torch.from_numpy(DR.keep_mask(DR.drop_key(seed, 0), np.arange(M * N).reshape(M, N), p))
    assert abs(float((keep ^ keep0).float().mean()) - 0.18) < 0.02                 # another step: an unrelated mask
    scale = 1.0 / (1.0 - p)
    lin = x.double() @ W.double().t() + b.double()
    # with a residual: x + drop(f(x))
    y = torch.zeros(M, N, device=DEV)
    hip.linear_fwd(xd, Wd, bd, y, residual=Rd, dropout_p=p, dropout_seed=seed, dropout_step=stepd)
    close(y, torch.where(keep, lin * scale, torch.zeros_like(lin)) + R.double(), what="gemm dropout before residual")
    # without: drop(relu(f(x)))
    hip.linear_fwd(xd, Wd, bd, y, act=1, dropout_p=p, dropout_seed=seed, dropout_step=stepd)
    close(y, torch.where(keep, lin.clamp_min(0) * scale, torch.zeros_like(lin)), what="gemm dropout after relu")
    gsrc = torch.randn(M, N)
    gd, out = g(gsrc), torch.zeros(M, N, device=DEV)
    hip.call("detr_hip_dropout_f32", gd.data_ptr(), out.data_ptr(), M * N, ctypes.c_float(p), seed, stepd.data_ptr())
    close(out, torch.where(keep, gsrc.double() * scale, torch.zeros(M, N, dtype=torch.float64)), rtol=1e-6, what="dropout_f32")


@pytest.mark.parametrize("B,T,S,compute", [(2, 100, 333, 0), (1, 70, 1050, 0), (2, 100, 333, 1)])
def test_fused_attention_dropout(hip, B, T, S, compute):
    """Dropout on the probabilities with the keyed counter-hash masks (site id as kernel argument, step seed in DEVICE
    memory), softmax scale folded into the kernel (transformer.py:307), and q / k / v / gradients addressed as column
    blocks of PACKED buffers with their own row strides (the engine's [rows, 768] / [rows, layers*256] layouts)."""
    from oracle import dropout_ref as DR
    torch.manual_seed(T + S)
    H, hd, p, site, step, scale = 8, 32, 0.1, 77, 0xBEEF1234, 32 ** -0.5
    D = H * hd
    q = (torch.randn(B, T, D, dtype=torch.float64) * 1.5).requires_grad_(True)
    k = torch.randn(B, S, D, dtype=torch.float64).requires_grad_(True)
    v = torch.randn(B, S, D, dtype=torch.float64).requires_grad_(True)
    qh, kh, vh = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    w = torch.softmax((qh * scale) @ kh.transpose(-1, -2), dim=-1)             # [B,H,T,S]
    keep = torch.from_numpy(DR.keep_mask(DR.drop_key(site, step), DR.attn_index(B * H, T, S).reshape(B, H, T, S), p))
    wd = torch.where(keep, w / (1.0 - p), torch.zeros_like(w))
    o = (wd @ vh).transpose(1, 2).reshape(B, T, D)
    do = torch.randn(B, T, D, dtype=torch.float64)
    o.backward(do)
    stepd = torch.tensor([step - (1 << 32)] + [0] * 7, dtype=torch.int32, device=DEV)
    qbuf = torch.full((B * T, 3 * D), 3.0, device=DEV)                         # q in columns 256..511 of a [rows, 768] buffer
    kvbuf = torch.full((B * S, 5 * D), -2.0, device=DEV)                       # k in block 1, v in block 3 of a [rows, 1280] buffer
    qbuf[:, D:2 * D] = q.detach().float().view(-1, D).to(DEV)
    kvbuf[:, D:2 * D] = k.detach().float().view(-1, D).to(DEV)
    kvbuf[:, 3 * D:4 * D] = v.detach().float().view(-1, D).to(DEV)
    qd, kd, vd = qbuf[:, D:2 * D], kvbuf[:, D:2 * D], kvbuf[:, 3 * D:4 * D]
    dod = g(do.float()).view(-1, D)
    od, lse = torch.zeros(B * T, D, device=DEV), torch.zeros(B * H, T, device=DEV)
    hip.attention(qd, kd, vd, od, lse, B, H, T, S, scale=scale, dropout_p=p, dropout_site=site, dropout_step=stepd, compute=compute)
    tol = 3e-5 if compute == 0 else 1.5e-2
    close(od.view(B, T, D), o, rtol=tol, what="attention+dropout fwd")
    dqb, dkvb = torch.full_like(qbuf, 5.0), torch.full_like(kvbuf, 5.0)
    delta = torch.zeros(B * H, T, device=DEV)
    hip.attention(qd, kd, vd, od, lse, B, H, T, S, scale=scale, dropout_p=p, dropout_site=site, dropout_step=stepd, compute=compute,
                  d_o=dod, dq=dqb[:, 0:D], dk=dkvb[:, 2 * D:3 * D], dv=dkvb[:, 4 * D:], delta=delta)
    gt = 5e-5 if compute == 0 else 2e-2
    close(dqb[:, 0:D].reshape(B, T, D), q.grad, rtol=gt, what="attention+dropout dq (w.r.t. the unscaled q)")
    close(dkvb[:, 2 * D:3 * D].reshape(B, S, D), k.grad, rtol=gt, what="attention+dropout dk")
    close(dkvb[:, 4 * D:].reshape(B, S, D), v.grad, rtol=gt, what="attention+dropout dv")
    # nothing outside the addressed column blocks was touched
    assert bool((dqb[:, D:] == 5.0).all()) and bool((dkvb[:, :2 * D] == 5.0).all()) and bool((dkvb[:, 3 * D:4 * D] == 5.0).all())


@pytest.mark.parametrize("M,K,a16,p,use_res,use_add,use_y16", [
    (800, 256, False, 0.1, True, True, False),       # decoder out-projection + norm1 (+ query_pos twin)
    (8400, 256, False, 0.1, True, False, True),      # encoder out-projection + norm1 (+ bf16 twin)
    (8400, 2048, True, 0.1, True, True, False),      # encoder FFN linear2 + norm2 (bf16 hidden activation)
    (37, 2048, True, 0.0, False, False, True),       # ragged tile, no dropout / residual
    (800, 512, False, 0.25, True, True, True),
])
def test_gemm_with_fused_layernorm_equals_the_two_launches(hip, M, K, a16, p, use_res, use_add, use_y16):
    """detr_gemm_desc.ln_* (round 4): out = drop((x W^T + b)) + residual and LayerNorm(out) from ONE launch of the row-complete
    32 x 256 tile kernel against the GEMM on the tile engine followed by detr_hip_layernorm_fwd: same C (the LayerNorm input the
    backward needs), same y / y2 / bf16 twin / mean / rstd.  The K loop feeds every accumulator the same k in the same order and
    the fused epilogue is the LayerNorm kernel's arithmetic, so everything must be IDENTICAL bits."""
    torch.manual_seed(M + K)
    N = 256
    x = g(torch.randn(M, K))
    if a16:
        x = x.to(torch.bfloat16)
    W = g(torch.randn(N, K) / K ** 0.5).to(torch.bfloat16)
    bias, res = g(torch.randn(N)), g(torch.randn(M, N))
    gam, bet = g(torch.rand(N) + 0.5), g(torch.randn(N))
    add = g(torch.randn(100 if M % 100 == 0 else M, N))
    step = torch.tensor([0x2468ACE, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=DEV)
    kw = dict(residual=res if use_res else None, dropout_p=p, dropout_seed=9, dropout_step=step)
    outs = []
    for fused in (False, True):
        C = torch.full((M, N), 3.0, device=DEV)
        y, y2 = torch.full((M, N), 5.0, device=DEV), torch.full((M, N), 6.0, device=DEV)
        y16 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
        hip.COMPUTE_BF16 = 1
        try:
            if fused:
                hip.linear_fwd(x, W, bias, C, ln=dict(gamma=gam, beta=bet, y=y, mean=mean, rstd=rstd, eps=1e-5, add=add if use_add else None,
                                                      y2=y2 if use_add else None, y16=y16 if use_y16 else None), **kw)
            else:
                hip.linear_fwd(x, W, bias, C, **kw)
                hip.layernorm_fwd(C, gam, bet, y, mean, rstd, 1e-5, add=add if use_add else None, y2=y2 if use_add else None,
                                  y16=y16 if use_y16 else None)
        finally:
            hip.COMPUTE_BF16 = 0
        torch.cuda.synchronize()
        outs.append((C, y, y2, y16, mean, rstd))
    for a, b, what in zip(outs[0], outs[1], ("C", "y", "y2", "y16", "mean", "rstd")):
        assert torch.equal(a, b), what
    C, y = outs[1][0], outs[1][1]
    ref = F.layer_norm(C.double().cpu(), (N,), gam.double().cpu(), bet.double().cpu(), 1e-5)
    close(y, ref, rtol=2e-5, what="fused layernorm vs fp64 on the same C")
    if p > 0.0:
        lin = (x.float() @ W.float().t() + bias).cpu()
        dropped = ((C.cpu() - (res.cpu() if use_res else 0.0)).abs() < 1e-12) & (lin.abs() > 1e-3)
        assert abs(float(dropped.float().mean()) - p) < 0.02
    # what the fused form does not cover is rejected loudly: fp32 compute, N != 256, a grouped launch
    spec = dict(gamma=gam, beta=bet, y=outs[1][1], mean=outs[1][4], rstd=outs[1][5], eps=1e-5)
    with pytest.raises(RuntimeError):
        hip.linear_fwd(x.float(), W.float(), bias, outs[1][0], ln=spec)                       # compute = 0
    hip.COMPUTE_BF16 = 1
    try:
        with pytest.raises(RuntimeError):
            hip.linear_fwd(x, W[:128], bias[:128], torch.zeros(M, 128, device=DEV), ln=spec)  # N = 128
        with pytest.raises(RuntimeError):
            hip.gemm_group([hip.linear_fwd_call(x, W, bias, outs[1][0], ln=spec), hip.linear_fwd_call(x, W, bias, outs[0][0])])
    finally:
        hip.COMPUTE_BF16 = 0


@pytest.mark.parametrize("rows,C,period", [(8400, 256, 1050), (800, 256, 100), (37, 64, 37)])
def test_layernorm_fused_outputs(hip, rows, C, period):
    """Fused side outputs of the LayerNorm launches: forward y2 = y + add[r % period] (the `+ pos` operand of the next
    attention block); backward dx += dx_add and dx_drop = dropout_bwd(dx) with the GEMM-epilogue mask of (site, step)."""
    from oracle import dropout_ref as DR
    torch.manual_seed(rows)
    x, gam, bet = torch.randn(rows, C) * 2 + 0.3, torch.rand(C) + 0.5, torch.randn(C)
    add = torch.randn(period, C)
    xd, gd, bd, addd = g(x), g(gam), g(bet), g(add)
    yd, y2d = torch.zeros(rows, C, device=DEV), torch.zeros(rows, C, device=DEV)
    mean, rstd = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
    y16 = torch.zeros(rows, C, device=DEV, dtype=torch.bfloat16)
    hip.layernorm_fwd(xd, gd, bd, yd, mean, rstd, 1e-5, add=addd, y2=y2d, y16=y16)
    yref = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-5)
    close(yd, yref, rtol=1e-5, what="layernorm fwd")
    assert torch.equal(y2d, yd + addd.repeat(rows // period, 1))
    assert torch.equal(y16, yd.to(torch.bfloat16))                      # RNE twin of y
    dy, extra = torch.randn(rows, C), torch.randn(rows, C)
    p, site, step = 0.1, 1234, 0x0BADF00D
    stepd = torch.tensor([step] + [0] * 7, dtype=torch.int32, device=DEV)
    dx0, dx1, dxdrop = (torch.zeros(rows, C, device=DEV) for _ in range(3))
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    hip.layernorm_bwd(g(dy), xd, gd, mean, rstd, dx0, dg, db)
    hip.layernorm_bwd(g(dy), xd, gd, mean, rstd, dx1, dg, db, dx_add=g(extra), dx_drop=dxdrop, dropout_p=p, dropout_site=site,
                      dropout_step=stepd)
    close(dx1, dx0.cpu().double() + extra.double(), rtol=1e-6, what="layernorm dx + dx_add")
    keep = torch.from_numpy(DR.keep_mask(DR.drop_key(site, step), np.arange(rows * C).reshape(rows, C), p)).to(DEV)
    assert bool((dxdrop[~keep] == 0).all())
    d16, dx2 = torch.zeros(rows, C, device=DEV, dtype=torch.bfloat16), torch.zeros(rows, C, device=DEV)
    hip.layernorm_bwd(g(dy), xd, gd, mean, rstd, dx2, dg, db, dx_add=g(extra), dx_drop16=d16, dropout_p=p, dropout_site=site,
                      dropout_step=stepd)
    assert torch.equal(dx2, dx1) and torch.equal(d16, dxdrop.to(torch.bfloat16))
    hip.layernorm_bwd(g(dy), xd, gd, mean, rstd, dx2, dg, db, dx_drop16=d16, dropout_p=0.0)     # p = 0: a plain bf16 copy of dx
    assert torch.equal(d16, dx0.to(torch.bfloat16))
    close(dxdrop, torch.where(keep, dx1 / (1.0 - p), torch.zeros_like(dx1)).cpu().double(), rtol=1e-6, what="layernorm dx_drop")


def test_multi_copy_and_set_u32(hip):
    torch.manual_seed(3)
    a, b = torch.randn(256, 256, device=DEV), torch.randn(512, device=DEV)
    a16 = torch.randn(64, 256, device=DEV).to(torch.bfloat16)
    da, db, da16 = torch.zeros_like(a), torch.ones_like(b), torch.zeros_like(a16)
    table = hip.copy_table([(a, da, 0), (b, db, 1), (a16, da16, 0)], DEV)
    hip.multi_copy(table)
    assert torch.equal(da, a) and torch.equal(db, b + 1.0) and torch.equal(da16, a16)
    w = torch.zeros(8, dtype=torch.int32, device=DEV)
    hip.call("detr_hip_set_u32x8", w.data_ptr(), 0xDEADBEEF, 1, 2, 3, 4, 5, 6, 7)
    assert [int(v) & 0xFFFFFFFF for v in w.cpu()] == [0xDEADBEEF, 1, 2, 3, 4, 5, 6, 7]


# ------------------------------------------------------------------------------------------
# bf16-compute mode (fp32 storage, bf16 MFMA, fp32 accumulate): exact vs a reference whose OPERANDS are rounded
# to bf16 (the only rounding the kernel adds), so layout / indexing bugs cannot hide behind a loose tolerance
# ------------------------------------------------------------------------------------------
def _bf(t):
    return t.float().bfloat16().double()


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 92, 256), (1000, 256, 147), (77, 40, 33), (8400, 256, 2048), (130, 512, 100)])
@pytest.mark.parametrize("ak,bk", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_bf16_compute_layouts(hip, M, N, K, ak, bk):
    torch.manual_seed(M + N + K + ak * 2 + bk)
    A, B = torch.randn(M, K), torch.randn(K, N)
    ref = _bf(A) @ _bf(B)
    lda, ldb = (K if ak else M) + 4, (K if bk else N) + 4
    Am = torch.zeros((M, lda) if ak else (K, lda))
    Bm = torch.zeros((N, ldb) if bk else (K, ldb))
    if ak:
        Am[:, :K] = A
    else:
        Am[:, :M] = A.t()
    if bk:
        Bm[:, :K] = B.t()
    else:
        Bm[:, :N] = B
    Ad, Bd = g(Am), g(Bm)
    bias, R = torch.randn(N), torch.randn(M, N)
    bd, Rd = g(bias), g(R)
    C = torch.full((M, N + 4), 7.0, device=DEV)
    hip.gemm(M, N, K, Ad, lda, ak, Bd, ldb, bk, C, N + 4, bias=bd, residual=Rd, ldr=N, act=1, compute=1)
    close(C[:, :N], (ref + bias.double() + R.double()).clamp_min(0), rtol=2e-5, what=f"bf16c gemm {M}x{N}x{K} ak={ak} bk={bk}")
    assert float((C[:, N:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K,bk", [(8400, 256, 256, 1), (8400, 2048, 256, 0), (300, 92, 256, 1), (1000, 256, 1024, 0),
                                       (33600, 256, 1024, 1), (77, 64, 64, 0)])
def test_gemm_bf16_weight_shadow_operand(hip, M, N, K, bk):
    """B operand already bf16 in memory (detr_gemm_desc.b_dtype = 1, the per-step weight shadow): bit-identical to the
    fp32-B path on the same (bf16-representable) values, fused epilogue included."""
    torch.manual_seed(M + N + K + bk)
    A = torch.randn(M, K)
    Bm = _bf(torch.randn(N, K) if bk else torch.randn(K, N)).float()
    bias, res = torch.randn(N), torch.randn(M, N)
    Ad, Bd, B16 = g(A), g(Bm), g(Bm).to(torch.bfloat16)
    outs = []
    for Bop in (Bd, B16):
        C = torch.zeros(M, N, device=DEV)
        hip.gemm(M, N, K, Ad, K, 1, Bop, K if bk else N, bk, C, N, bias=g(bias), residual=g(res), ldr=N, act=1, compute=1)
        outs.append(C)
    ref = torch.relu(_bf(A) @ (Bm.double().t() if bk else Bm.double()) + bias.double() + res.double())
    close(outs[0], ref, rtol=5e-5, what="bf16c gemm (fp32 B)")
    assert torch.equal(outs[0], outs[1]), "bf16-B path must match the fp32-B path bit for bit"


@pytest.mark.parametrize("N,H,W,Ci,Co,stride", [(2, 13, 17, 64, 64, 1), (1, 20, 27, 128, 128, 2), (2, 25, 42, 256, 256, 1)])
def test_conv3x3_bf16_weight_shadow_operand(hip, N, H, W, Ci, Co, stride):
    """conv3x3 forward / dgrad with the kernel tensor already bf16 in memory (w_dtype = 1): bit-identical to the fp32-w path."""
    torch.manual_seed(N + H + W + Ci + stride)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x, dy = g(torch.randn(N, H, W, Ci)), g(torch.randn(N, Ho, Wo, Co))
    w = _bf(torch.randn(3, 3, Ci, Co) / (3 * Ci ** 0.5)).float()
    wd, w16 = g(w), g(w).to(torch.bfloat16)
    shift = g(torch.randn(Co))
    ys, dxs = [], []
    for wop in (wd, w16):
        y = torch.zeros(N, Ho, Wo, Co, device=DEV)
        hip.conv3x3(0, x, wop, y, N, H, W, Ci, Ho, Wo, Co, stride, bias=shift, act=1, compute=1)
        dx = torch.zeros(N, H, W, Ci, device=DEV)
        hip.conv3x3(1, dy, wop, dx, N, H, W, Ci, Ho, Wo, Co, stride, compute=1)
        ys.append(y); dxs.append(dx)
    assert torch.equal(ys[0], ys[1]) and torch.equal(dxs[0], dxs[1])
    assert float(ys[0].abs().max()) > 0 and float(dxs[0].abs().max()) > 0


@pytest.mark.parametrize("M,N,K,bk", [(8400, 256, 256, 1), (33600, 1024, 256, 0), (1000, 64, 256, 1), (531, 256, 64, 0)])
def test_gemm_bf16_activation_storage(hip, M, N, K, bk):
    """bf16 STORAGE of A / C / residual / mask (a_dtype .. m_dtype = 1): same accumulators as the fp32-storage call on
    the same values, the result rounded once to bf16 -- compared bit for bit with bf16(fp32-storage result)."""
    torch.manual_seed(M + N + K + bk + 5)
    A = _bf(torch.randn(M, K)).float()
    Bm = _bf(torch.randn(N, K) if bk else torch.randn(K, N)).float()
    res, msk, bias = _bf(torch.randn(M, N)).float(), _bf(torch.randn(M, N)).float(), torch.randn(N)
    Ad, Bd, rd, md, bd = g(A), g(Bm).to(torch.bfloat16), g(res), g(msk), g(bias)
    C32 = torch.zeros(M, N, device=DEV)
    hip.gemm(M, N, K, Ad, K, 1, Bd, K if bk else N, bk, C32, N, bias=bd, residual=rd, ldr=N, mask=md, ldmask=N, act=1, compute=1)
    C16 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm(M, N, K, Ad.to(torch.bfloat16), K, 1, Bd, K if bk else N, bk, C16, N, bias=bd, residual=rd.to(torch.bfloat16), ldr=N,
             mask=md.to(torch.bfloat16), ldmask=N, act=1, compute=1)
    assert torch.equal(C16, C32.to(torch.bfloat16)), "bf16-storage GEMM differs from bf16(fp32-storage GEMM)"
    assert float(C32.abs().max()) > 0
    # weight-gradient form: both operands MN-contiguous bf16 activations, fp32 split-K output
    dy, x = _bf(torch.randn(M, 64)).float(), A[:, :64].contiguous()
    ws = torch.empty(8 * 1024 * 1024, device=DEV)
    outs = []
    for cast in (lambda t: t, lambda t: t.to(torch.bfloat16)):
        dw = torch.zeros(64, 64, device=DEV)
        hip.gemm(64, 64, M, cast(g(dy)), 64, 0, cast(g(x)), 64, 0, dw, 64, split_k=16, workspace=ws, compute=1)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1]), "bf16-storage wgrad differs from the fp32-storage wgrad"
    close(outs[0], dy.double().t() @ x.double(), rtol=5e-5, what="bf16-storage wgrad")


@pytest.mark.parametrize("N,H,W,Ci,Co,stride", [(2, 13, 17, 64, 64, 1), (1, 20, 27, 128, 128, 2), (2, 25, 42, 256, 256, 1), (2, 21, 67, 64, 128, 2), (1, 9, 131, 256, 64, 2)])
def test_conv3x3_bf16_activation_storage(hip, monkeypatch, N, H, W, Ci, Co, stride):
    """conv3x3 forward / dgrad / wgrad and the stem max pooling on bf16-STORED tensors: bit-identical to bf16(result of the
    fp32-storage kernels) on the same (bf16-representable) values; weight gradients (fp32) identical."""
    torch.manual_seed(N + H + W + Ci + stride + 9)
    hip.ensure_workspace(DEV)
    # (the stride-1 all-bf16 calls would take the halo-staged kernel, which sums the taps in another order: it has its own
    #  test below; this one pins the storage-type identity of the tile kernel)
    hip.set_tuning("DETR_HIP_CONV_HALO", "2")      # (reset by the autouse fixture _reset_library_tuning)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    b16 = lambda t: t.to(torch.bfloat16)
    x, dy = g(_bf(torch.randn(N, H, W, Ci)).float()), g(_bf(torch.randn(N, Ho, Wo, Co)).float())
    w16 = g(_bf(torch.randn(3, 3, Ci, Co) / (3 * Ci ** 0.5)).float()).to(torch.bfloat16)
    shift, res, msk = g(torch.randn(Co)), g(_bf(torch.randn(N, Ho, Wo, Co)).float()), g(_bf(torch.randn(N, H, W, Ci)).float())
    y32, y16 = torch.zeros(N, Ho, Wo, Co, device=DEV), torch.zeros(N, Ho, Wo, Co, device=DEV, dtype=torch.bfloat16)
    hip.conv3x3(0, x, w16, y32, N, H, W, Ci, Ho, Wo, Co, stride, bias=shift, residual=res, act=1, compute=1)
    hip.conv3x3(0, b16(x), w16, y16, N, H, W, Ci, Ho, Wo, Co, stride, bias=shift, residual=b16(res), act=1, compute=1)
    assert torch.equal(y16, b16(y32)) and float(y32.abs().max()) > 0
    dx32, dx16 = torch.zeros(N, H, W, Ci, device=DEV), torch.zeros(N, H, W, Ci, device=DEV, dtype=torch.bfloat16)
    hip.conv3x3(1, dy, w16, dx32, N, H, W, Ci, Ho, Wo, Co, stride, mask=msk, compute=1)
    hip.conv3x3(1, b16(dy), w16, dx16, N, H, W, Ci, Ho, Wo, Co, stride, mask=b16(msk), compute=1)
    assert torch.equal(dx16, b16(dx32)) and float(dx32.abs().max()) > 0
    dws = []
    if stride == 2:
        hip.set_tuning("DETR_HIP_WGRAD_FUSED", "4")      # the nine-tap kernel's stride-2 form whatever the channel count
    for cast in (lambda t: t, b16):
        dw = torch.zeros(3, 3, Ci, Co, device=DEV)
        hip.conv3x3(2, cast(x), cast(dy), dw, N, H, W, Ci, Ho, Wo, Co, stride, split=5, compute=1)
        dws.append(dw)
    # stride 1: both storage types take the nine-tap kernel (same sums, same order: identical bits).  Stride 2 (round 5): bf16-stored tensors take the
    # nine-tap kernel's stride-2 form, fp32-stored ones the per-tap kernel -- the same products summed in another order; both against fp64
    ref = torch.nn.grad.conv2d_weight(x.double().cpu().permute(0, 3, 1, 2), (Co, Ci, 3, 3), dy.double().cpu().permute(0, 3, 1, 2), stride=stride, padding=1)
    ref = ref.permute(2, 3, 1, 0)                    # (Co, Ci, kh, kw) -> (kh, kw, Ci, Co)
    scale = float(ref.abs().max())
    for dw in dws:
        assert float((dw.double().cpu() - ref).abs().max()) <= 2e-5 * scale, float((dw.double().cpu() - ref).abs().max()) / scale
    if stride == 1:
        assert torch.equal(dws[0], dws[1])
    assert float(dws[0].abs().max()) > 0
    if stride == 1 and Ci == 64:          # the stem pooling pair on the same tensors
        C = Ci
        H2, W2 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        p32, p16 = torch.zeros(N, H2, W2, C, device=DEV), torch.zeros(N, H2, W2, C, device=DEV, dtype=torch.bfloat16)
        a32, a16 = (torch.zeros(N, H2, W2, C, device=DEV, dtype=torch.uint8) for _ in range(2))
        hip.call("detr_hip_maxpool3x3s2_fwd_f32", x.data_ptr(), p32.data_ptr(), a32.data_ptr(), N, H, W, C, H2, W2)
        hip.call("detr_hip_maxpool3x3s2_fwd_bf16", b16(x).data_ptr(), p16.data_ptr(), a16.data_ptr(), N, H, W, C, H2, W2)
        assert torch.equal(p16, b16(p32)) and torch.equal(a16, a32)
        gp = g(_bf(torch.randn(N, H2, W2, C)).float())
        d32, d16 = torch.zeros(N, H, W, C, device=DEV), torch.zeros(N, H, W, C, device=DEV, dtype=torch.bfloat16)
        hip.call("detr_hip_maxpool3x3s2_bwd_f32", gp.data_ptr(), a32.data_ptr(), x.data_ptr(), d32.data_ptr(), N, H, W, C, H2, W2)
        xb, gb = b16(x), b16(gp)
        hip.call("detr_hip_maxpool3x3s2_bwd_bf16", gb.data_ptr(), a16.data_ptr(), xb.data_ptr(), d16.data_ptr(), N, H, W, C, H2, W2)
        assert torch.equal(d16, b16(d32))


@pytest.mark.parametrize("rows", ["3", "4"])
@pytest.mark.parametrize("N,H,W,C", [(2, 19, 45, 64), (1, 8, 32, 128), (2, 25, 70, 256), (1, 40, 33, 64), (3, 9, 31, 128), (1, 13, 42, 512)])
def test_conv3x3_halo_staged_kernel(hip, monkeypatch, N, H, W, C, rows):
    """The halo-staged stride-1 kernel (csrc/conv_halo.h: bf16 x / w / y, 4 x 32 (DETR_HIP_CONV_HALO=3) and 8 x 32 (=4) pixel
    tiles, one staged input patch per 32-channel chunk) -- forward (+ BN shift, ReLU) and input gradient (+ ReLU mask) against
    fp64 on the same bf16 operands (one bf16 rounding) and against the tile kernel on the same call (DETR_HIP_CONV_HALO=2): the two
    sum the (tap, chunk) products in different orders, i.e. may differ by one bf16 ulp on a small fraction of the outputs.  The
    shapes cover ragged tile rows / columns, several tiles per image, several images and the 512-channel maps of layer4."""
    torch.manual_seed(N + H + W + C)
    b16 = lambda t: g(t.float()).to(torch.bfloat16)
    x, dy = _bf(torch.randn(N, H, W, C)), _bf(torch.randn(N, H, W, C))
    w = _bf(torch.randn(3, 3, C, C) / (3 * C ** 0.5))
    shift, msk = torch.randn(C).double(), _bf(torch.randn(N, H, W, C))
    w_oihw = w.permute(3, 2, 0, 1).contiguous()
    ref_y = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w_oihw, padding=1).permute(0, 2, 3, 1) + shift)
    ref_dx = torch.nn.functional.conv_transpose2d(dy.permute(0, 3, 1, 2), w_oihw, padding=1).permute(0, 2, 3, 1)
    ref_dx = torch.where(msk > 0, ref_dx, torch.zeros_like(ref_dx))
    xd, dyd, wd, md, sd = b16(x), b16(dy), b16(w), b16(msk), g(shift.float())
    outs = {}
    for mode in ("0", "2"):
        hip.set_tuning("DETR_HIP_CONV_HALO", rows if mode == "0" else mode)
        y = torch.full((N, H, W, C), 7.0, device=DEV, dtype=torch.bfloat16)
        dx = torch.full((N, H, W, C), 7.0, device=DEV, dtype=torch.bfloat16)
        hip.conv3x3(0, xd, wd, y, N, H, W, C, H, W, C, 1, bias=sd, act=1, compute=1)
        hip.conv3x3(1, dyd, wd, dx, N, H, W, C, H, W, C, 1, mask=md, compute=1)
        torch.cuda.synchronize()
        outs[mode] = (y.float().cpu().double(), dx.float().cpu().double())
    hip.set_tuning("DETR_HIP_CONV_HALO", None)
    for what, halo, tile, ref in (("forward", outs["0"][0], outs["2"][0], ref_y), ("dgrad", outs["0"][1], outs["2"][1], ref_dx)):
        scale = float(ref.abs().max())
        assert scale > 0
        err = (halo - ref).abs()
        assert float((err / (ref.abs() + 1e-2 * scale)).max()) < 2.0 ** -8 * 1.1, f"halo conv {what}: more than one bf16 rounding from fp64"
        diff = (halo - tile).abs()
        assert float((diff > 0).double().mean()) < 5e-3, (what, float((diff > 0).double().mean()))
        assert float((diff / (tile.abs() + 1e-2 * scale)).max()) <= 2.0 ** -7, what


@pytest.mark.parametrize("M,use_mask,pad", [(32, True, 0), (31, True, 0), (33, False, 0), (4096, True, 0), (4100, True, 8), (66800, True, 0), (66811, False, 16), (8, True, 0)])
def test_conv1x1_backward_fused_reads_dy_once(hip, M, use_mask, pad):
    """csrc/bwd_fused.hip (round 5): input gradient and weight gradient of a 64 -> 256 channel 1x1 convolution in one pass over dY.  The input
    gradient must equal the streaming / tile GEMM on the same operands bit for bit (one fp32 accumulator per output, k ascending, one bf16
    rounding); the weight gradient -- summed over per-workgroup slabs in a fixed order, scaled per output channel and ACCUMULATED into dW --
    against fp64 on the same bf16 operands, against the split-K GEMM, and twice in a row (bit-identical: no atomics).  Ragged row counts
    (the last strip is partial: out-of-range rows are zero operands and are not stored), padded leading dimensions, fewer strips than CUs."""
    torch.manual_seed(M + pad)
    d1, d2 = 64, 256
    b16 = lambda t: g(t.float()).to(torch.bfloat16)
    def padded(rows, cols):
        full = b16(torch.randn(rows, cols + pad))
        return full[:, :cols]
    dy, a, w = padded(M, d2), padded(M, d1), padded(d1, d2)
    scale = g(torch.rand(d2) + 0.5)
    dw0 = g(torch.randn(d1, d2))
    scratch = torch.empty(hip.conv1x1_bwd_fused_scratch_floats(M), device=DEV)
    outs = []
    for _ in range(2):
        da = torch.full((M + 2, d1), 7.0, device=DEV, dtype=torch.bfloat16)            # two guard rows behind the tensor
        dw = dw0.clone()
        hip.conv1x1_bwd_fused(dy, a, w, da[:M], dw, scratch, scale=scale, use_mask=use_mask, alpha=0.5)
        torch.cuda.synchronize()
        assert float(da[M:].float().min()) == 7.0 and float(da[M:].float().max()) == 7.0, "rows past M were written"
        outs.append((da[:M].clone(), dw.clone()))
    nda = int((outs[0][0].view(torch.int16) != outs[1][0].view(torch.int16)).sum())
    ndw = int((outs[0][1] != outs[1][1]).sum())
    assert nda == 0 and ndw == 0, f"run-to-run: {nda} input-gradient and {ndw} weight-gradient entries differ"
    # the two-launch form
    da_ref = torch.empty(M, d1, device=DEV, dtype=torch.bfloat16)
    hip.gemm(M, d1, d2, dy, dy.stride(0), 1, w, w.stride(0), 1, da_ref, d1, mask=(a if use_mask else None), ldmask=(a.stride(0) if use_mask else 0), compute=1)
    dw_ref = dw0.clone()
    sk = hip.pick_split_k(d1, d2, M)
    if sk > 1:
        hip.gemm(d1, d2, M, a, a.stride(0), 0, dy, dy.stride(0), 0, dw_ref, d2, alpha=0.5, scale=scale, split_k=sk, compute=1)
    else:
        hip.gemm(d1, d2, M, a, a.stride(0), 0, dy, dy.stride(0), 0, dw_ref, d2, alpha=0.5, scale=scale, residual=dw_ref, ldr=d2, compute=1)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0].view(torch.int16), da_ref.view(torch.int16)), "input gradient differs from the GEMM's"
    a64, dy64 = a.double().cpu(), dy.double().cpu()
    want = dw0.double().cpu() + 0.5 * scale.double().cpu()[None, :] * (a64.t() @ dy64)
    tol = 2e-6 * float((a64.abs().t() @ dy64.abs()).max()) + 1e-6
    assert float((outs[0][1].double().cpu() - want).abs().max()) <= tol
    assert float((dw_ref.double().cpu() - want).abs().max()) <= tol


@pytest.mark.parametrize("N,H,W,Ci,Co", [(1, 8, 32, 128, 128), (2, 25, 70, 256, 256), (3, 9, 31, 128, 256), (1, 13, 42, 512, 512), (2, 50, 84, 256, 128),
                                         (1, 4, 33, 128, 128), (1, 1, 1, 128, 128), (2, 100, 167, 128, 128), (1, 5, 300, 64, 128), (1, 6, 40, 32, 384)])
def test_conv3x3_halo_dma_is_bit_identical_to_the_register_staged_kernel(hip, N, H, W, Ci, Co):
    """The LDS-DMA form of the halo kernel (csrc/conv_halo_dma.h, round 5: patch and kernel tiles requested straight into LDS) against
    the register-staged kernel (DETR_HIP_CONV_DMA=2) on the same call: same tiles, same wave grid, same MFMA order per accumulator --
    every output bit equal; forward (+ shift, ReLU) and input gradient (+ ReLU mask), ragged tile rows / columns, a one-pixel map, maps wider
    than eight tiles, channel counts that differ between the two sides, and twice in a row (the pipeline's trailing requests land in LDS
    the epilogue reuses: a missing wait shows as a run-to-run difference)."""
    torch.manual_seed(N * 7 + H + W + Ci + Co)
    b16 = lambda t: g(t.float()).to(torch.bfloat16)
    x, dy = b16(torch.randn(N, H, W, Ci)), b16(torch.randn(N, H, W, Co))
    w = b16(torch.randn(3, 3, Ci, Co) / (3 * Ci ** 0.5))
    shift, msk = g(torch.randn(Co)), b16(torch.randn(N, H, W, Ci))
    outs = {}
    for mode in ("0", "2", "0b", "3"):                   # 3: the requests in front of the fragment reads (A/B form kept in the library)
        hip.set_tuning("DETR_HIP_CONV_DMA", None if mode in ("0", "0b") else mode)
        y = torch.full((N, H, W, Co), 7.0, device=DEV, dtype=torch.bfloat16)
        dx = torch.full((N, H, W, Ci), 7.0, device=DEV, dtype=torch.bfloat16)
        hip.conv3x3(0, x, w, y, N, H, W, Ci, H, W, Co, 1, bias=shift, act=1, compute=1)
        hip.conv3x3(1, dy, w, dx, N, H, W, Ci, H, W, Co, 1, mask=msk, compute=1)
        torch.cuda.synchronize()
        outs[mode] = (y.clone(), dx.clone())
    hip.set_tuning("DETR_HIP_CONV_DMA", None)
    assert float(outs["2"][0].float().abs().max()) > 0 and float(outs["2"][1].float().abs().max()) > 0
    for k in ("0", "0b", "3"):
        assert torch.equal(outs[k][0].view(torch.int16), outs["2"][0].view(torch.int16)), ("forward", k)
        assert torch.equal(outs[k][1].view(torch.int16), outs["2"][1].view(torch.int16)), ("dgrad", k)


@pytest.mark.parametrize("N,H,W,C", [(2, 19, 45, 64), (1, 8, 64, 128), (2, 25, 70, 256), (1, 41, 33, 64), (1, 14, 42, 512), (3, 9, 31, 128)])
def test_conv3x3_stride2_dgrad_halo_classes(hip, N, H, W, C):
    """Stride-2 input gradient on bf16 tensors: the pixel-parity classes on the halo-staged kernel (csrc/conv_halo.h, class form:
    four taps per chunk over the dy patch, missing taps skipped, output pixels (2 h2 + ph, 2 w2 + pw)) against fp64 on the same bf16
    operands (one bf16 rounding) and against the tile kernel's class launches (DETR_HIP_DGRAD_S2_CLASSES=3; other summation order:
    one bf16 ulp on a small fraction).  Odd and even input sizes (classes of different sizes, a last dy row / column without a
    successor), ragged tiles, several images, 64 ... 512 channels, with the ReLU mask."""
    torch.manual_seed(N + H + W + C)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    b16 = lambda t: g(t.float()).to(torch.bfloat16)
    dy = _bf(torch.randn(N, Ho, Wo, C))
    w = _bf(torch.randn(3, 3, C, C) / (3 * C ** 0.5))
    msk = _bf(torch.randn(N, H, W, C))
    w_oihw = w.permute(3, 2, 0, 1).contiguous()
    ref = torch.nn.functional.conv_transpose2d(dy.permute(0, 3, 1, 2), w_oihw, stride=2, padding=1,
                                               output_padding=(H + 2 - 3 - 2 * (Ho - 1), W + 2 - 3 - 2 * (Wo - 1))).permute(0, 2, 3, 1)
    assert ref.shape == (N, H, W, C)
    ref = torch.where(msk > 0, ref, torch.zeros_like(ref))
    dyd, wd, md = b16(dy), b16(w), b16(msk)
    outs = {}
    for mode in (None, "3"):
        hip.set_tuning("DETR_HIP_DGRAD_S2_CLASSES", mode)
        try:
            dx = torch.full((N, H, W, C), 7.0, device=DEV, dtype=torch.bfloat16)
            hip.conv3x3(1, dyd, wd, dx, N, H, W, C, Ho, Wo, C, 2, mask=md, compute=1)
            torch.cuda.synchronize()
        finally:
            hip.set_tuning("DETR_HIP_DGRAD_S2_CLASSES", None)
        outs[mode] = dx.float().cpu().double()
    halo, tile = outs[None], outs["3"]
    # the grid order of the class workgroups (round 5: round-robin; DETR_HIP_DGRAD_S2_CLASSES=4: one class after the other) changes nothing but time
    hip.set_tuning("DETR_HIP_DGRAD_S2_CLASSES", "4")
    try:
        dx = torch.full((N, H, W, C), 7.0, device=DEV, dtype=torch.bfloat16)
        hip.conv3x3(1, dyd, wd, dx, N, H, W, C, Ho, Wo, C, 2, mask=md, compute=1)
        torch.cuda.synchronize()
    finally:
        hip.set_tuning("DETR_HIP_DGRAD_S2_CLASSES", None)
    assert torch.equal(dx.float().cpu().double(), halo), "class grid order changed the result"
    scale = float(ref.abs().max())
    assert scale > 0
    assert float(((tile - ref).abs() / (ref.abs() + 1e-2 * scale)).max()) < 2.0 ** -8 * 1.1
    assert float(((halo - ref).abs() / (ref.abs() + 1e-2 * scale)).max()) < 2.0 ** -8 * 1.1, "more than one bf16 rounding from fp64"
    diff = (halo - tile).abs()
    assert float((diff > 0).double().mean()) < 5e-3
    assert float((diff / (tile.abs() + 1e-2 * scale)).max()) <= 2.0 ** -7


@pytest.mark.parametrize("compute", [0, 1])
def test_gemm_group_matches_individual_launches(hip, compute):
    """detr_hip_gemm_group_f32: members of one kernel variant (Q / K / V projections, their dgrads, their split-K weight
    gradients with fused bias gradients) in ONE launch -- bit-identical to the individual launches; a mixed list falls
    back to sequential launches."""
    torch.manual_seed(50 + compute)
    hip.ensure_workspace(DEV)
    M, D = 8400, 256
    xs = [g(torch.randn(M, D)) for _ in range(3)]
    W = g(torch.randn(3 * D, D) / 16)
    Wop = W.to(torch.bfloat16) if compute else W
    bias = g(torch.randn(3 * D))
    def fwd_calls(outs):
        return [hip.linear_fwd_call(xs[i], Wop[i * D:(i + 1) * D], bias[i * D:(i + 1) * D], outs[i], alpha=(0.5 if i == 0 else 1.0))
                for i in range(3)]
    ref, grp = [torch.zeros(M, D, device=DEV) for _ in range(3)], [torch.zeros(M, D, device=DEV) for _ in range(3)]
    old = hip.COMPUTE_BF16
    hip.COMPUTE_BF16 = compute
    try:
        for a, kw in fwd_calls(ref):
            hip.gemm(*a, **kw)
        hip.gemm_group(fwd_calls(grp))
        for i in range(3):
            assert torch.equal(ref[i], grp[i]), f"grouped forward member {i} differs"
        assert float(ref[0].abs().max()) > 0
        # weight gradients (split-K + fused bias gradient) of different shapes in one group
        dys = [g(torch.randn(M, D)) for _ in range(3)]
        def wg_calls(dws, dbs):
            return [hip.linear_wgrad_call(dys[i], xs[i], dws[i], alpha=(0.5 if i == 0 else 1.0), bias_grad=dbs[i]) for i in range(3)]
        dw_r, db_r = [torch.zeros(D, D, device=DEV) for _ in range(3)], [torch.zeros(D, device=DEV) for _ in range(3)]
        dw_g, db_g = [torch.zeros(D, D, device=DEV) for _ in range(3)], [torch.zeros(D, device=DEV) for _ in range(3)]
        for a, kw in wg_calls(dw_r, db_r):
            hip.gemm(*a, **kw)
        hip.gemm_group(wg_calls(dw_g, db_g))
        for i in range(3):
            assert torch.equal(dw_r[i], dw_g[i]) and torch.equal(db_r[i], db_g[i]), f"grouped wgrad member {i} differs"
        assert float(dw_r[1].abs().max()) > 0 and float(db_r[1].abs().max()) > 0
        # mixed variants (K-contiguous and MN-contiguous B) -> sequential fallback, same results
        o1, o2 = torch.zeros(M, D, device=DEV), torch.zeros(M, D, device=DEV)
        hip.gemm_group([hip.linear_fwd_call(xs[0], Wop[0:D], bias[0:D], o1), hip.linear_dgrad_call(xs[1], Wop[0:D], o2)])
        r1, r2 = torch.zeros(M, D, device=DEV), torch.zeros(M, D, device=DEV)
        hip.linear_fwd(xs[0], Wop[0:D], bias[0:D], r1)
        hip.linear_dgrad(xs[1], Wop[0:D], r2)
        assert torch.equal(o1, r1) and torch.equal(o2, r2)
    finally:
        hip.COMPUTE_BF16 = old


@pytest.mark.parametrize("compute", [0, 1])
def test_deferred_split_k_reductions_match_immediate(hip, compute):
    """detr_gemm_desc.defer_out + detr_hip_splitk_reduce_many: 20 weight-gradient GEMMs (plain and grouped, with fused bias
    gradients, small and large outputs, one output hit twice) whose reductions are queued and run 16 per launch give
    bit-identical results to the immediate reductions (same reduction body, same summation order)."""
    torch.manual_seed(60 + compute)
    hip.ensure_workspace(DEV)
    M = 8400
    shapes = [(256, 256)] * 6 + [(256, 2048), (2048, 256), (256, 256)] + [(128, 64), (92, 256), (256, 2048), (2048, 256)] * 3
    xs = [g(torch.randn(M, k)) for (n, k) in shapes]
    dys = [g(torch.randn(M, n)) for (n, k) in shapes]
    def run(deferred):
        dws = [torch.zeros(n, k, device=DEV) for (n, k) in shapes]
        dbs = [torch.zeros(n, device=DEV) for (n, k) in shapes]
        dws[8], dbs[8] = dws[0], dbs[0]          # two reductions into one gradient must not share a launch: the queue is flushed
        calls = [hip.linear_wgrad_call(dys[i], xs[i], dws[i], alpha=(0.5 if i % 3 == 0 else 1.0), bias_grad=dbs[i])
                 for i in range(len(shapes))]
        if deferred:
            hip.begin_deferred_reduces(DEV)
        try:
            hip.gemm_group(calls[0:3])
            hip.gemm_group(calls[3:6])
            for a, kw in calls[6:]:
                hip.gemm(*a, **kw)
            if deferred:
                assert hip.DEFER is not None and len(hip.DEFER) >= 1
                assert float(dws[7].abs().max()) > 0.0, "the repeated output did not flush the queue"
                assert float(dws[12].abs().max()) == 0.0 and float(dws[20].abs().max()) == 0.0, "a deferred reduction ran early"
        finally:
            hip.flush_reduces(end=True)
        assert hip.DEFER is None
        torch.cuda.synchronize()
        return dws, dbs
    old = hip.COMPUTE_BF16
    hip.COMPUTE_BF16 = compute
    try:
        (w_i, b_i), (w_d, b_d) = run(False), run(True)
    finally:
        hip.COMPUTE_BF16 = old
    for i in range(len(shapes)):
        assert torch.equal(w_i[i], w_d[i]) and torch.equal(b_i[i], b_d[i]), f"deferred wgrad {i} {shapes[i]} differs"
        assert float(w_i[i].abs().max()) > 0
    ref = dys[1].double().t() @ xs[1].double()
    assert (w_d[1].double() - ref).abs().max() < (2.0 if compute else 2e-2)


def test_gemm_bf16_compute_split_k(hip):
    torch.manual_seed(31)
    M, N, K = 256, 512, 20000
    A, B = torch.randn(K, M), torch.randn(K, N)
    scale = torch.rand(N) + 0.5
    Ad, Bd, sd = g(A), g(B), g(scale)
    ws = torch.empty(16 * 1024 * 1024, device=DEV)
    C = torch.zeros(M, N, device=DEV)
    hip.gemm(M, N, K, Ad, M, 0, Bd, N, 0, C, N, scale=sd, split_k=24, workspace=ws, compute=1)
    close(C, (_bf(A).t() @ _bf(B)) * scale.double(), rtol=5e-5, what="bf16c split-k")


@pytest.mark.parametrize("N,H,W,Ci,Co,stride", [(2, 13, 17, 64, 64, 1), (1, 20, 27, 128, 128, 2), (2, 50, 84, 256, 256, 1),
                                                (1, 25, 42, 512, 512, 2), (2, 9, 11, 32, 64, 1)])
def test_conv3x3_bf16_compute_all_modes(hip, N, H, W, Ci, Co, stride):
    torch.manual_seed(N + H + W + Ci + stride + 100)
    x = _bf(torch.randn(N, H, W, Ci)).requires_grad_(True)                  # operands already on the bf16 grid:
    w = _bf(torch.randn(3, 3, Ci, Co) / (3 * Ci ** 0.5)).requires_grad_(True)  # the kernel's rounding is then exact
    scale, shift = torch.rand(Co, dtype=torch.float64) + 0.5, torch.randn(Co, dtype=torch.float64)
    z = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), None, stride=stride, padding=1).permute(0, 2, 3, 1)
    Ho, Wo = z.shape[1], z.shape[2]
    y = torch.relu(z * scale + shift)
    xd, wd = g(x.detach().float()), g(w.detach().float())
    sd, hd_ = g(scale.float()), g(shift.float())
    yd = torch.zeros(N, Ho, Wo, Co, device=DEV)
    hip.conv3x3(0, xd, wd, yd, N, H, W, Ci, Ho, Wo, Co, stride, scale=sd, bias=hd_, act=1, compute=1)
    close(yd, y, rtol=3e-5, what="bf16c conv3x3 fwd")
    dz = _bf(torch.randn(N, Ho, Wo, Co))
    z.backward(dz)
    dzd = g(dz.float())
    dxd = torch.zeros(N, H, W, Ci, device=DEV)
    hip.conv3x3(1, dzd, wd, dxd, N, H, W, Ci, Ho, Wo, Co, stride, compute=1)
    close(dxd, x.grad, rtol=3e-5, what="bf16c conv3x3 dgrad")
    ws = torch.empty(32 * 1024 * 1024, device=DEV)
    old = hip.WORKSPACE
    try:
        hip.WORKSPACE = ws
        dwd = torch.zeros(3, 3, Ci, Co, device=DEV)
        hip.conv3x3(2, xd, dzd, dwd, N, H, W, Ci, Ho, Wo, Co, stride, scale=sd, compute=1)
        close(dwd, w.grad * scale, rtol=5e-5, what="bf16c conv3x3 wgrad")
    finally:
        hip.WORKSPACE = old


def test_scale_cols_bf16_group_matches_single_launches(hip):
    """The one-launch frozen-BN fold over a device table == the per-kernel fold, bit for bit (custom_layers.py:21-24)."""
    torch.manual_seed(77)
    shapes = [(64, 64), (9 * 64, 64), (256, 1024), (9 * 512, 512), (1, 4), (37, 12)]
    ws = [torch.randn(r, c, device=DEV) for r, c in shapes]
    scs = [torch.rand(c, device=DEV) + 0.5 for _, c in shapes]
    one = [torch.zeros(r, c, device=DEV, dtype=torch.bfloat16) for r, c in shapes]
    grp = [torch.zeros(r, c, device=DEV, dtype=torch.bfloat16) for r, c in shapes]
    for w, sc, o in zip(ws, scs, one):
        hip.call("detr_hip_scale_cols_bf16", w.data_ptr(), sc.data_ptr(), o.data_ptr(), w.shape[0], w.shape[1])
    table = torch.tensor([[w.data_ptr(), sc.data_ptr(), o.data_ptr(), w.numel() // 4, w.shape[1] // 4]
                          for w, sc, o in zip(ws, scs, grp)], dtype=torch.int64).to(DEV)
    hip.call("detr_hip_scale_cols_bf16_group", table.data_ptr(), len(shapes))
    torch.cuda.synchronize()
    for w, sc, a, b in zip(ws, scs, one, grp):
        assert torch.equal(a, b)
        assert torch.equal(a, (w * sc).to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K,bk,use_res,use_mask,act", [
    (20000, 256, 64, 0, True, False, 1),      # layer1 conv3 forward: bias + residual + ReLU
    (16411, 256, 64, 1, True, True, 0),       # layer1 conv1 input gradient: residual + ReLU mask, ragged last strip
    (16384, 64, 64, 0, False, False, 1),      # one column slice
    (33600, 512, 128, 0, True, False, 1),     # layer2 conv3 forward
    (17000, 512, 128, 1, True, True, 0),      # layer2 conv1 input gradient
    (16390, 128, 128, 1, False, True, 0),
    (16500, 64, 256, 0, False, False, 1),     # layer1 conv1 forward (two K chunks per strip)
    (33600, 1024, 256, 1, True, True, 0),     # layer3 conv1 input gradient
    (16700, 1024, 256, 0, True, False, 1),    # layer3 conv3 forward ([k][n] weights, ragged last strip)
    (16384, 128, 256, 1, False, True, 0),     # one 128-column group
])
@pytest.mark.parametrize("slices_per_wave", ["1", "0"])      # DETR_HIP_STREAM_NW: one column slice per wave / the rule (two where the 8-wave form exists)
def test_gemm_stream_bf16_short_k(hip, M, N, K, bk, use_res, use_mask, act, slices_per_wave):
    """The streaming short-K kernel (csrc/gemm_stream.h; bf16 A / B / C / residual / mask, K in {64, 128}, M >= 16384) against
    fp64 on the same bf16 operands (one bf16 rounding of the result) and against the generic tile engine on the same call
    (DETR_HIP_GEMM_STREAM=2): the two may differ by the order of the fp32 products inside an MFMA, i.e. by one bf16 ulp on
    a small fraction of the outputs."""
    import os
    if slices_per_wave == "0" and not ((K == 256 and N % 128 == 0) or (K == 128 and N % 256 == 0)):
        pytest.skip("two slices per wave exist for K = 256 (N % 128 == 0) and K = 128 (N % 256 == 0)")
    hip.set_tuning("DETR_HIP_STREAM_NW", slices_per_wave)
    torch.manual_seed(M + N + K + bk)
    A = _bf(torch.randn(M, K))
    Bm = _bf(torch.randn(N, K) / K ** 0.5 if bk else torch.randn(K, N) / K ** 0.5)
    res, msk, bias = _bf(torch.randn(M, N)), _bf(torch.randn(M, N)), torch.randn(N).double()
    ref = A @ (Bm.t() if bk else Bm) + bias
    if use_res:
        ref = ref + res
    if act:
        ref = torch.relu(ref)
    if use_mask:
        ref = torch.where(msk > 0, ref, torch.zeros_like(ref))
    b16 = lambda t: g(t.float()).to(torch.bfloat16)
    Ad, Bd, rd, md, bd = b16(A), b16(Bm), b16(res), b16(msk), g(bias.float())
    outs = []
    for mode in ("3", "2"):          # 3: the streaming kernel keeps every shape it can take (the ring kernel has first call on some K = 256 shapes)
        hip.set_tuning("DETR_HIP_GEMM_STREAM", mode)
        try:
            C = torch.full((M, N), 7.0, device=DEV, dtype=torch.bfloat16)
            hip.gemm(M, N, K, Ad, K, 1, Bd, K if bk else N, bk, C, N, bias=bd, residual=rd if use_res else None, ldr=N if use_res else 0,
                     mask=md if use_mask else None, ldmask=N if use_mask else 0, act=act, compute=1)
            torch.cuda.synchronize()
        finally:
            hip.set_tuning("DETR_HIP_GEMM_STREAM", None)
        outs.append(C.float().cpu().double())
    hip.set_tuning("DETR_HIP_STREAM_NW", None)
    stream, generic = outs
    scale = float(ref.abs().max())
    err = (stream - ref).abs()
    assert float((err / (ref.abs() + 1e-2 * scale)).max()) < 2.0 ** -8 * 1.05, "stream GEMM: more than one bf16 rounding from fp64"
    diff = (stream - generic).abs()
    assert float((diff > 0).double().mean()) < 2e-3, float((diff > 0).double().mean())
    assert float((diff / (generic.abs() + 1e-2 * scale)).max()) <= 2.0 ** -7




@pytest.mark.parametrize("M,N,bk,use_bias,act,use_mask,alpha,p", [
    (8400, 2048, 1, True, 1, False, 1.0, 0.1),        # encoder FFN linear1 forward: bias + ReLU + keyed dropout
    (8400, 2048, 0, False, 0, True, 1.0 / 0.9, 0.0),  # input gradient of linear2: alpha = 1/(1-p), ReLU / dropout mask
    (4100, 1024, 1, True, 0, False, 0.5, 0.25),       # ragged last strip, alpha and dropout together
])
@pytest.mark.parametrize("slices_per_wave", ["1", "0"])
def test_gemm_stream_extended_epilogue_equals_tile_engine(hip, M, N, bk, use_bias, act, use_mask, alpha, p, slices_per_wave):
    """The K = 256 streaming kernel with the extended epilogue ((acc + bias) * alpha, keyed dropout after the activation) that
    takes the transformer's FFN GEMMs at M = B*L: same call on the generic tile engine (DETR_HIP_GEMM_STREAM=2).  The dropout
    masks are the SAME counter hash, so the zero patterns must coincide exactly; kept values may differ by one bf16 ulp
    (order of the fp32 products inside an MFMA)."""
    K = 256
    hip.set_tuning("DETR_HIP_STREAM_NW", slices_per_wave)
    torch.manual_seed(M + N + bk)
    A = _bf(torch.randn(M, K))
    Bm = _bf(torch.randn(N, K) / K ** 0.5 if bk else torch.randn(K, N) / K ** 0.5)
    msk, bias = _bf(torch.randn(M, N)), torch.randn(N)
    b16 = lambda t: g(t.float()).to(torch.bfloat16)
    Ad, Bd, md, bd = b16(A), b16(Bm), b16(msk), g(bias.float())
    step = torch.tensor([0x1234567, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=DEV)
    outs = []
    for mode in ("3", "2"):          # 3: the streaming kernel keeps every shape it can take (the ring kernel has first call on some K = 256 shapes)
        hip.set_tuning("DETR_HIP_GEMM_STREAM", mode)
        try:
            C = torch.full((M, N), 7.0, device=DEV, dtype=torch.bfloat16)
            hip.gemm(M, N, K, Ad, K, 1, Bd, K if bk else N, bk, C, N, bias=bd if use_bias else None, alpha=alpha,
                     mask=md if use_mask else None, ldmask=N if use_mask else 0, act=act, compute=1, dropout_p=p, dropout_seed=77,
                     dropout_step=step)
            torch.cuda.synchronize()
        finally:
            hip.set_tuning("DETR_HIP_GEMM_STREAM", None)
        outs.append(C.float().cpu().double())
    hip.set_tuning("DETR_HIP_STREAM_NW", None)
    stream, generic = outs
    ref = A @ (Bm.t() if bk else Bm) + (bias.double() if use_bias else 0.0)
    ref = ref * alpha
    if act:
        ref = torch.relu(ref)
    if p > 0.0:
        kept = stream != 0
        frac = float((stream == 0).double().mean())
        base = float((ref == 0).double().mean()) if act else 0.0
        assert abs(frac - (base + (1 - base) * p)) < 0.01, (frac, base)            # the drop rate is p
        assert torch.equal(stream == 0, generic == 0), "stream / tile engine dropout masks differ"
        ref = torch.where(kept, ref / (1.0 - p), torch.zeros_like(ref))
    if use_mask:
        ref = torch.where(msk > 0, ref, torch.zeros_like(ref))
    scale = float(ref.abs().max())
    err = (stream - ref).abs()
    assert float((err / (ref.abs() + 1e-2 * scale)).max()) < 2.0 ** -8 * 1.05, "stream GEMM (extended epilogue): more than one bf16 rounding from fp64"
    diff = (stream - generic).abs()
    assert float((diff > 0).double().mean()) < 2e-3, float((diff > 0).double().mean())
    assert float((diff / (generic.abs() + 1e-2 * scale)).max()) <= 2.0 ** -7


@pytest.mark.parametrize("tile", [3, 1])
@pytest.mark.parametrize("ak,bk", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K,split", [(200, 136, 2048, 1), (72, 264, 1000, 1), (136, 72, 8 * 331, 5), (264, 200, 8 * 1001, 7)])
def test_gemm_bf16_deep_k_tiles_are_bit_identical(hip, tile, ak, bk, M, N, K, split):
    """The all-bf16 tile GEMM with 64-deep K tiles (DETR_HIP_GEMM_K64=1: every eligible call) against the 32-deep variant (=2):
    each wave feeds its MFMAs the same k in the same order, so outputs must be IDENTICAL bits -- both operand layouts, 64x64 and
    128x128 tiles, K and split ranges that are not multiples of 64 (half-empty last tiles), ragged M / N, the fused bias
    gradient (rowsum_a) of an M-contiguous A, and the deterministic split-K slabs."""
    hip.ensure_workspace(DEV)                        # split-K through the slab workspace (the atomic fallback has no fixed order)
    torch.manual_seed(M + N + K)
    A = g(torch.randn(M, K) if ak else torch.randn(K, M)).to(torch.bfloat16)
    Bm = g((torch.randn(N, K) if bk else torch.randn(K, N)) / K ** 0.5).to(torch.bfloat16)
    bias = g(torch.randn(N))
    outs = []
    for mode in ("1", "2"):
        hip.set_tuning("DETR_HIP_GEMM_K64", mode)
        hip.set_tuning("DETR_HIP_GEMM_TILE", str(tile))
        try:
            use_rs = (not ak) and split > 1
            C = torch.full((M, N), 0.5, device=DEV) if split > 1 else torch.full((M, N), 7.0, device=DEV, dtype=torch.bfloat16)
            rs = torch.zeros(M, device=DEV) if use_rs else None
            kw = dict(split_k=split, compute=1)
            if split == 1:
                kw.update(bias=bias, act=1)
            if use_rs:
                kw.update(rowsum_a=rs)
            hip.gemm(M, N, K, A, K if ak else M, ak, Bm, K if bk else N, bk, C, N, **kw)
            torch.cuda.synchronize()
        finally:
            hip.set_tuning("DETR_HIP_GEMM_K64", None)
            hip.set_tuning("DETR_HIP_GEMM_TILE", None)
        outs.append((C.float().cpu(), None if rs is None else rs.cpu()))
    (deep, rs_deep), (ref32, rs32) = outs
    A64 = A.double().cpu() if ak else A.double().cpu().t()
    B64 = Bm.double().cpu().t() if bk else Bm.double().cpu()
    want = A64 @ B64
    want = torch.relu(want + bias.double().cpu()) if split == 1 else want + 0.5
    assert float((ref32.double() - want).abs().max()) < (2.0 ** -7 if split == 1 else 1e-3) * float(want.abs().max())
    assert torch.equal(deep, ref32), float((deep - ref32).abs().max())
    if rs_deep is not None:
        assert torch.equal(rs_deep, rs32)
        assert float((rs_deep.double() - A64.sum(1)).abs().max()) < 1e-3 * float(A64.sum(1).abs().max() + 1)


@pytest.mark.parametrize("tile", [3, 1, 2, 5])
@pytest.mark.parametrize("M,N,K,epi", [(200, 136, 256, "bias_relu"), (75, 264, 96, "res_mask"), (1000, 256, 64, "drop_res"),
                                        (333, 72, 520, "scale_bias_res_mask_relu"), (130, 8, 32, "plain")])
def test_gemm_wide_bf16_epilogue_is_bit_identical(hip, tile, M, N, K, epi):
    """The all-bf16 epilogue (gemm_core.h epilogue_wide16: 8 columns per lane, every residual / mask row of a strip requested up
    front) against the 4-columns-per-lane form (DETR_HIP_EPI_WIDE=2): the per-element arithmetic is the same function, so the
    outputs must be IDENTICAL bits -- every tile shape of the bf16 engine, ragged M and N (N % 8 == 0), all epilogue items."""
    torch.manual_seed(M + N + K)
    b16 = torch.bfloat16
    A = g(torch.randn(M, K)).to(b16)
    Bm = g(torch.randn(N, K) / K ** 0.5).to(b16)
    bias, scale = g(torch.randn(N)), g(torch.rand(N) + 0.5)
    res, msk = g(torch.randn(M, N)).to(b16), g(torch.randn(M, N)).to(b16)
    step = torch.tensor([0x2468ace, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=DEV)
    kw = dict(compute=1)
    if "bias" in epi:
        kw.update(bias=bias)
    if "scale" in epi:
        kw.update(scale=scale, alpha=0.75)
    if "relu" in epi:
        kw.update(act=1)
    if "res" in epi:
        kw.update(residual=res, ldr=N)
    if "mask" in epi:
        kw.update(mask=msk, ldmask=N)
    if "drop" in epi:
        kw.update(dropout_p=0.1, dropout_seed=9, dropout_step=step)
    outs = []
    for mode in (None, "2"):
        hip.set_tuning("DETR_HIP_EPI_WIDE", mode)
        hip.set_tuning("DETR_HIP_GEMM_TILE", str(tile))
        hip.set_tuning("DETR_HIP_GEMM_STREAM", "2")
        try:
            C = torch.full((M, N), 7.0, device=DEV, dtype=b16)
            hip.gemm(M, N, K, A, K, 1, Bm, K, 1, C, N, **kw)
            torch.cuda.synchronize()
        finally:
            for k in ("DETR_HIP_EPI_WIDE", "DETR_HIP_GEMM_TILE", "DETR_HIP_GEMM_STREAM"):
                hip.set_tuning(k, None)
        outs.append(C.float().cpu())
    wide, narrow = outs
    want = A.double().cpu() @ Bm.double().cpu().t()
    if "scale" in epi:
        want = want * scale.double().cpu()
    if "bias" in epi:
        want = want + bias.double().cpu()
    if "scale" in epi:
        want = want * 0.75
    if "drop" not in epi:
        if "res" in epi:
            want = want + res.double().cpu()
        if "relu" in epi:
            want = torch.relu(want)
        if "mask" in epi:
            want = torch.where(msk.double().cpu() > 0, want, torch.zeros_like(want))
        assert float((narrow.double() - want).abs().max()) < 2.0 ** -7 * float(want.abs().max())
    assert float(narrow.abs().max()) > 0
    assert torch.equal(wide, narrow), float((wide - narrow).abs().max())


@pytest.mark.parametrize("split", [1, 2, 4])
@pytest.mark.parametrize("B,T,S,p", [(2, 200, 333, 0.1), (1, 70, 40, 0.0), (1, 129, 1050, 0.1)])
def test_fused_attention_bf16_in_workgroup_split(hip, split, B, T, S, p):
    """The bf16 attention kernels with the streamed dimension cut into 1 / 2 / 4 runs per workgroup (DETR_HIP_ATTN_SPLIT): forward merge of the partial softmaxes (incl. runs that own no keys:
    S = 40 is two tiles for four runs), partial dQ / dK / dV sums, ragged last tiles -- all against the fp64 reference with the
    oracle's dropout masks.  The split must not change what is computed."""
    from oracle import dropout_ref as DR
    torch.manual_seed(B * 13 + T + S)
    H, hd, seed = 8, 32, 4242
    D = H * hd
    q = (torch.randn(B, T, D, dtype=torch.float64) * 0.6).requires_grad_(True)
    k = torch.randn(B, S, D, dtype=torch.float64).requires_grad_(True)
    v = torch.randn(B, S, D, dtype=torch.float64).requires_grad_(True)
    qh, kh, vh = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    sc = qh @ kh.transpose(-1, -2)
    w = torch.softmax(sc, dim=-1)
    if p > 0.0:
        keep = torch.from_numpy(DR.keep_mask(DR.drop_key(seed, 0), DR.attn_index(B * H, T, S).reshape(B, H, T, S), p))
        w = torch.where(keep, w / (1.0 - p), torch.zeros_like(w))
    o = (w @ vh).transpose(1, 2).reshape(B, T, D)
    do = torch.randn(B, T, D, dtype=torch.float64)
    o.backward(do)
    qd, kd, vd, dod = (t.view(-1, D) for t in (g(q.detach().float()), g(k.detach().float()), g(v.detach().float()), g(do.float())))
    hip.set_tuning("DETR_HIP_ATTN_SPLIT", split)
    try:
        od, lse = torch.full((B * T, D), 7.0, device=DEV), torch.zeros(B * H, T, device=DEV)
        hip.attention(qd, kd, vd, od, lse, B, H, T, S, compute=1, dropout_p=p, dropout_site=seed)
        dq, dk, dv = (torch.full_like(t, 3.0).view(B, -1, D) for t in (qd, kd, vd))
        delta = torch.zeros(B * H, T, device=DEV)
        hip.attention(qd, kd, vd, od, lse, B, H, T, S, compute=1, dropout_p=p, dropout_site=seed, d_o=dod, dq=dq.view(-1, D),
                      dk=dk.view(-1, D), dv=dv.view(-1, D), delta=delta)
        torch.cuda.synchronize()
    finally:
        hip.set_tuning("DETR_HIP_ATTN_SPLIT", None)
    close(od.view(B, T, D), o, rtol=1.5e-2, what=f"bf16 attention fwd (split {split})")
    close(lse.view(B, H, T), torch.logsumexp(sc, dim=-1), rtol=3e-3, what=f"bf16 attention lse (split {split})")
    close(dq, q.grad, rtol=2.5e-2, what=f"bf16 attention dq (split {split})")
    close(dk, k.grad, rtol=2.5e-2, what=f"bf16 attention dk (split {split})")
    close(dv, v.grad, rtol=2.5e-2, what=f"bf16 attention dv (split {split})")


def _pack_bits(t):
    """[M, N] bool -> [M, N / 8] uint8, bit (n & 7) of byte n / 8 (the layout of detr_gemm_desc.maskbits_out)."""
    M, N = t.shape
    w = (2 ** torch.arange(8, dtype=torch.int32)).view(1, 1, 8)
    return (t.view(M, N // 8, 8).to(torch.int32) * w).sum(-1).to(torch.uint8)


@pytest.mark.parametrize("M,N,K,bk,use_res", [
    (20000, 256, 64, 0, True),       # streaming kernel, 4 slices per workgroup (layer1 conv3 forward / conv1 input gradient)
    (17000, 512, 128, 1, True),      # streaming kernel, K = 128, ragged last strip
    (16500, 1024, 256, 1, True),     # streaming kernel, K = 256
    (8400, 2048, 256, 1, False),     # streaming kernel: input gradient of input_proj (mask only)
    (8400, 2048, 512, 1, True),      # tile engine (128x128, wide epilogue): layer4
    (300, 128, 96, 0, True),         # tile engine 64x64, ragged rows
])
@pytest.mark.parametrize("slices_per_wave", ["1", "0"])      # DETR_HIP_STREAM_NW: both forms of the K = 128 / 256 streaming kernel
def test_gemm_bitpacked_relu_masks(hip, M, N, K, bk, use_res, slices_per_wave):
    """Round 4: the ReLU masks of the bottleneck block outputs as BITS.  Producer: a GEMM with a ReLU epilogue also writes one byte
    per 8 outputs, bit = (stored bf16 output > 0), and its C is unchanged by that.  Consumer: the same masked GEMM once with the
    bf16 activation as `mask` and once with the bytes (m_dtype 2) -- bit-identical results on the streaming kernel and on the tile
    engine (the wide all-bf16 epilogue)."""
    if slices_per_wave == "0" and K == 64:
        pytest.skip("two slices per wave exist for K = 128 / 256")
    hip.set_tuning("DETR_HIP_STREAM_NW", slices_per_wave)
    torch.manual_seed(M + N + K)
    b16 = lambda t: g(t.float()).to(torch.bfloat16)
    A = b16(_bf(torch.randn(M, K)))
    Bm = b16(_bf(torch.randn(N, K) / K ** 0.5 if bk else torch.randn(K, N) / K ** 0.5))
    res = b16(_bf(torch.randn(M, N)))
    bias = g(torch.randn(N))
    ldb = K if bk else N
    # ---- producer
    C0 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    C1 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    bits = torch.full((M, N // 8), 0xAA, device=DEV, dtype=torch.uint8)
    kw = dict(bias=bias, residual=res if use_res else None, ldr=N if use_res else 0, act=1, compute=1)
    hip.gemm(M, N, K, A, K, 1, Bm, ldb, bk, C0, N, **kw)
    hip.gemm(M, N, K, A, K, 1, Bm, ldb, bk, C1, N, maskbits_out=bits, **kw)
    torch.cuda.synchronize()
    assert torch.equal(C0, C1)
    assert torch.equal(bits.cpu(), _pack_bits((C1.float() > 0).cpu()))
    frac = float((C1.float() > 0).float().mean())
    assert 0.2 < frac < 0.8
    # ---- consumer: same call, bf16 mask vs its bits
    D0 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    D1 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    kw = dict(residual=res if use_res else None, ldr=N if use_res else 0, compute=1)
    hip.gemm(M, N, K, A, K, 1, Bm, ldb, bk, D0, N, mask=C1, ldmask=N, **kw)
    hip.gemm(M, N, K, A, K, 1, Bm, ldb, bk, D1, N, mask=bits, ldmask=bits.stride(0), **kw)
    torch.cuda.synchronize()
    assert torch.equal(D0, D1) and float(D0.float().abs().max()) > 0
    assert float((D0.float() == 0).float().mean()) > 0.15          # the mask did something
    # fp32 C / split-K cannot carry bits: rejected loudly
    Cf = torch.zeros(M, N, device=DEV)
    with pytest.raises(RuntimeError):
        hip.gemm(M, N, K, A, K, 1, Bm, ldb, bk, Cf, N, mask=bits, ldmask=bits.stride(0), compute=1)
    hip.set_tuning("DETR_HIP_STREAM_NW", None)


@pytest.mark.parametrize("N,H,W,C,stride", [(2, 19, 45, 64, 1), (1, 25, 70, 128, 1), (2, 24, 40, 128, 2), (1, 13, 42, 512, 1)])
def test_conv3x3_bitpacked_relu_masks(hip, N, H, W, C, stride):
    """The 3x3 convolution with bit-packed ReLU masks (round 4): the forward (+ BN shift, ReLU) also writes (y > 0) as one byte per 8
    channels without changing y; the input gradient takes those bytes instead of the bf16 activation -- bit-identical dx.  Covers the
    halo-staged kernel (8- and 4-row tiles), the stride-2 class form and the implicit-GEMM tile kernel (stride-2 forward)."""
    torch.manual_seed(N + H + W + C + stride)
    b16 = lambda t: g(t.float()).to(torch.bfloat16)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = b16(_bf(torch.randn(N, H, W, C)))
    w = b16(_bf(torch.randn(3, 3, C, C) / (3 * C ** 0.5)))
    shift = g(torch.randn(C) * 0.1)
    y0 = torch.zeros(N, Ho, Wo, C, device=DEV, dtype=torch.bfloat16)
    y1 = torch.zeros_like(y0)
    bits = torch.full((N * Ho * Wo, C // 8), 0x55, device=DEV, dtype=torch.uint8)
    hip.conv3x3(0, x, w, y0, N, H, W, C, Ho, Wo, C, stride, bias=shift, act=1, compute=1)
    hip.conv3x3(0, x, w, y1, N, H, W, C, Ho, Wo, C, stride, bias=shift, act=1, compute=1, maskbits_out=bits)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.equal(bits.cpu(), _pack_bits((y1.float() > 0).view(-1, C).cpu()))
    # input gradient masked by the INPUT activation x (as conv2's dgrad is masked by y1 in the engine): bf16 tensor vs its bits
    xbits = _pack_bits((x.float() > 0).view(-1, C).cpu()).to(DEV)
    dy = b16(_bf(torch.randn(N, Ho, Wo, C)))
    dx0 = torch.zeros(N, H, W, C, device=DEV, dtype=torch.bfloat16)
    dx1 = torch.zeros_like(dx0)
    hip.conv3x3(1, dy, w, dx0, N, H, W, C, Ho, Wo, C, stride, mask=x, compute=1)
    hip.conv3x3(1, dy, w, dx1, N, H, W, C, Ho, Wo, C, stride, mask=xbits, compute=1)
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1) and float(dx0.float().abs().max()) > 0


@pytest.mark.parametrize("compute", [0, 1])
@pytest.mark.parametrize("M,N,K,split,rowsum", [(256, 1024, 4096, 8, False), (100, 92, 2048, 5, True), (200, 136, 33 * 64, 16, True), (64, 256, 8192, 12, False)])
def test_tile_ordered_split_k_slabs_equal_row_major_slabs(hip, compute, M, N, K, split, rowsum):
    """Round 4: split-K partials stored in MFMA register order (tile-ordered slabs, padded to whole tiles) and un-permuted by the reduce
    launch must give the SAME BITS as row-major slabs (DETR_HIP_SLAB_TS=2) -- fp32 and bf16 compute, ragged outputs, the fused bias
    gradient riding behind the slabs, 64x64 / 128x128 / 64x128 tiles -- and the 3x3 weight-gradient kernels likewise."""
    hip.ensure_workspace(DEV)
    torch.manual_seed(M + N + K + compute)
    dt = torch.bfloat16 if (compute and M % 8 == 0 and N % 8 == 0) else torch.float32      # (bf16 storage needs 16-byte rows)
    A = g(_bf(torch.randn(K, M)).float()).to(dt)     # weight-gradient layout: both operands reduction-major
    Bm = g(_bf(torch.randn(K, N) / K ** 0.5).float()).to(dt)
    scale = g(torch.rand(N) + 0.5)
    outs = []
    for mode in (None, 2):
        hip.set_tuning("DETR_HIP_SLAB_TS", mode)
        try:
            C = torch.full((M, N), 0.25, device=DEV)
            rs = torch.zeros(M, device=DEV) if rowsum else None
            hip.gemm(M, N, K, A, M, 0, Bm, N, 0, C, N, alpha=0.5, scale=scale, split_k=split, compute=compute, rowsum_a=rs)
            torch.cuda.synchronize()
        finally:
            hip.set_tuning("DETR_HIP_SLAB_TS", None)
        outs.append((C.cpu(), None if rs is None else rs.cpu()))
    (c_ts, r_ts), (c_rm, r_rm) = outs
    assert torch.equal(c_ts, c_rm), float((c_ts - c_rm).abs().max())
    if rowsum:
        assert torch.equal(r_ts, r_rm)
    want = 0.25 + 0.5 * scale.cpu().double() * (A.double().cpu().t() @ Bm.double().cpu())
    assert float((c_ts.double() - want).abs().max()) < (2e-2 if compute else 1e-4) * float(want.abs().max())


@pytest.mark.parametrize("C,stride", [(64, 1), (128, 1), (128, 2)])
def test_conv3x3_wgrad_tile_ordered_slabs_equal_row_major(hip, C, stride):
    hip.ensure_workspace(DEV)
    torch.manual_seed(C + stride)
    N, H, W = 2, 23, 40
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = g(torch.randn(N, H, W, C)).to(torch.bfloat16)
    dy = g(torch.randn(N, Ho, Wo, C)).to(torch.bfloat16)
    scale = g(torch.rand(C) + 0.5)
    outs = []
    for mode in (None, 2):
        hip.set_tuning("DETR_HIP_SLAB_TS", mode)
        try:
            dw = torch.zeros(3, 3, C, C, device=DEV)
            hip.conv3x3(2, x, dy, dw, N, H, W, C, Ho, Wo, C, stride, scale=scale, split=6, compute=1)
            torch.cuda.synchronize()
        finally:
            hip.set_tuning("DETR_HIP_SLAB_TS", None)
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0


RING_CASES = [
    # M, N, K, b_kcontig, epilogue, C / residual fp32?
    (600, 256, 128, 1, "res_mask", False),
    (600, 256, 128, 0, "bias_relu", False),
    (1333, 128, 256, 1, "mask", False),
    (1333, 128, 256, 0, "plain", False),
    (2100, 512, 192, 0, "bias_res_relu", False),
    (2100, 384, 192, 1, "scale_bias_res_mask_relu", False),
    (840, 256, 2048, 1, "bias_res", True),
    (840, 256, 2048, 0, "drop_res", True),
    (4200, 1024, 512, 1, "res_mask", False),
    (33, 256, 64 * 3, 1, "bias_relu", False),
    (8, 136, 128, 0, "bias", False),
]


@pytest.mark.parametrize("variant", ["auto", "ns2", "bn128", "wgs64", "wgs512"])
@pytest.mark.parametrize("M,N,K,bk,epi,f32", RING_CASES)
def test_gemm_ring_is_bit_identical_to_the_tile_engine(hip, variant, M, N, K, bk, epi, f32):
    """The 8-wave LDS-DMA ring kernel (csrc/gemm_ring.h, round 5) against the 4-wave tile engine (DETR_HIP_GEMM_RING=2): same MFMA
    instruction, every accumulator sees its k-steps in ascending order, same epilogue function -> IDENTICAL bits.  Covers both B
    layouts ([n][k]: XOR-swizzled K-contiguous image; [k][n]: transpose-read image), ragged M (last tile / last 32-row block / a row
    wave without rows), N that is not a multiple of the column panel, 2- and 3-stage rings, both column panels, forced workgroup
    counts (row pitches from 8 to 256), every epilogue item incl. fp32 C / residual and the keyed dropout.  The tile engine itself
    is pinned against fp64 here as well."""
    torch.manual_seed(M + N + K + bk)
    b16 = torch.bfloat16
    A = g(torch.randn(M, K)).to(b16)
    Bm = (g(torch.randn(N, K) / K ** 0.5) if bk else g(torch.randn(K, N) / K ** 0.5)).to(b16)
    bias, scale = g(torch.randn(N)), g(torch.rand(N) + 0.5)
    rdt = torch.float32 if f32 else b16
    res, msk = g(torch.randn(M, N)).to(rdt), g(torch.randn(M, N)).to(b16)
    step = torch.tensor([0x13579bd, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=DEV)
    kw = dict(compute=1)
    if "bias" in epi:
        kw.update(bias=bias)
    if "scale" in epi:
        kw.update(scale=scale, alpha=0.75)
    if "relu" in epi:
        kw.update(act=1)
    if "res" in epi:
        kw.update(residual=res, ldr=N)
    if "mask" in epi:
        kw.update(mask=msk, ldmask=N)
    if "drop" in epi:
        kw.update(dropout_p=0.1, dropout_seed=11, dropout_step=step)
    forced = {"auto": {}, "ns2": {"DETR_HIP_RING_NS": "2"}, "bn128": {"DETR_HIP_RING_BN": "128"}, "wgs64": {"DETR_HIP_RING_WGS": "64"},
              "wgs512": {"DETR_HIP_RING_WGS": "512"}}[variant]
    plan = (ctypes.c_int32 * 8)()
    outs = []
    for ring in ("1", "2"):
        env = {"DETR_HIP_GEMM_RING": ring, "DETR_HIP_GEMM_STREAM": "2"}
        env.update(forced)
        for k, v in env.items():
            hip.set_tuning(k, v)
        try:
            has_plan = hip.load().detr_hip_gemm_ring_plan(M, N, K, plan)
            C = torch.full((M, N), 7.0, device=DEV, dtype=rdt)
            hip.gemm(M, N, K, A, K, 1, Bm, Bm.stride(0), bk, C, N, **kw)
            torch.cuda.synchronize()
        finally:
            for k in env:
                hip.set_tuning(k, None)
        outs.append(C.float().cpu())
    ring_out, tile_out = outs
    if not has_plan:
        pytest.skip(f"no ring plan for this shape under {variant}")
    want = A.double().cpu() @ (Bm.double().cpu().t() if bk else Bm.double().cpu())
    if "scale" in epi:
        want = want * scale.double().cpu()
    if "bias" in epi:
        want = want + bias.double().cpu()
    if "scale" in epi:
        want = want * 0.75
    if "drop" not in epi:
        if "res" in epi:
            want = want + res.double().cpu()
        if "relu" in epi:
            want = torch.relu(want)
        if "mask" in epi:
            want = torch.where(msk.double().cpu() > 0, want, torch.zeros_like(want))
        tol = (2.0 ** -16 if f32 else 2.0 ** -7) * float(want.abs().max())
        assert float((tile_out.double() - want).abs().max()) < tol
    assert float(tile_out.abs().max()) > 0
    assert torch.equal(ring_out, tile_out), (list(plan), float((ring_out - tile_out).abs().max()),
                                             int((ring_out != tile_out).sum()), ring_out.numel())


def test_gemm_ring_takes_the_step_shapes_and_repeats_exactly(hip):
    """At the step's own sizes (M = 33600 / 8400 rows) the default dispatch takes the ring kernel (plan exists, result equals the tile
    engine's bit for bit) and twenty back-to-back launches return the same bits every time -- the screen for a staged buffer read
    before its DMA has landed (guide: 'place reads by the vmcnt / barrier count, never by clean runs'; a race shows up as rare
    wrong tiles that come and go)."""
    b16 = torch.bfloat16
    for M, N, K, bk, kwx in [(33600, 256, 1024, 1, "mask"), (33600, 256, 1024, 0, "bias_relu"), (8400, 256, 2048, 1, "res32"),
                             (8400, 2048, 512, 0, "bias_res_relu"), (33600, 1024, 512, 1, "res_mask")]:
        torch.manual_seed(M + N + K)
        A = g(torch.randn(M, K)).to(b16)
        Bm = (g(torch.randn(N, K) / K ** 0.5) if bk else g(torch.randn(K, N) / K ** 0.5)).to(b16)
        f32 = kwx == "res32"
        rdt = torch.float32 if f32 else b16
        kw = dict(compute=1)
        if "bias" in kwx:
            kw.update(bias=g(torch.randn(N)))
        if "relu" in kwx:
            kw.update(act=1)
        if "res" in kwx:
            kw.update(residual=g(torch.randn(M, N)).to(rdt), ldr=N)
        if "mask" in kwx:
            kw.update(mask=g(torch.randn(M, N)).to(b16), ldmask=N)
        plan = (ctypes.c_int32 * 8)()
        assert hip.load().detr_hip_gemm_ring_plan(M, N, K, plan) == 1
        outs = []
        for ring in (None, "2"):
            hip.set_tuning("DETR_HIP_GEMM_RING", ring)
            try:
                C = torch.full((M, N), 7.0, device=DEV, dtype=rdt)
                hip.gemm(M, N, K, A, K, 1, Bm, Bm.stride(0), bk, C, N, **kw)
                torch.cuda.synchronize()
                outs.append(C.clone())
                if ring is None:
                    for _ in range(20):
                        C.fill_(3.0)
                        hip.gemm(M, N, K, A, K, 1, Bm, Bm.stride(0), bk, C, N, **kw)
                        assert torch.equal(C, outs[0])
            finally:
                hip.set_tuning("DETR_HIP_GEMM_RING", None)
        assert torch.equal(outs[0], outs[1]), (M, N, K, bk, list(plan), int((outs[0] != outs[1]).sum()))


@pytest.mark.parametrize("ns", [None, "3"])
@pytest.mark.parametrize("M,N,K,split", [(256, 1024, 8400, 8), (1024, 256, 4200, 5), (128, 512, 33600, 64), (264, 136, 2100, 3),
                                         (512, 2048, 1050, 2), (120, 128, 1000, 4)])
def test_gemm_ring_split_k_weight_gradient_is_bit_identical(hip, ns, M, N, K, split):
    """The split-K weight gradients of the backbone's 1x1 convolutions (dW = x^T dy, both operands bf16 and M / N-contiguous) on the
    8-wave ring kernel (gemm_ring.h: gemm_ring_wgrad_kernel) against the 4-wave engine (DETR_HIP_GEMM_RING=3): same 128 x 128 tiles,
    same split ranges (incl. K that is no multiple of 64 or 32: the last stage is zero-filled by the descriptor bound), same MFMA
    order, tile-ordered slabs in the same unit order -> the reduced gradient must be IDENTICAL bits; ragged M / N, 3- and 4-stage
    rings.  Pinned against fp64 as well."""
    torch.manual_seed(M + N + K)
    hip.ensure_workspace(DEV)
    b16 = torch.bfloat16
    A = g(torch.randn(K, M)).to(b16)
    Bm = g(torch.randn(K, N) / K ** 0.5).to(b16)
    outs = []
    for ring in (None, "3"):
        env = {"DETR_HIP_GEMM_RING": ring, "DETR_HIP_GEMM_TILE": "1", "DETR_HIP_RING_NS": ns}
        for k, v in env.items():
            hip.set_tuning(k, v)
        try:
            C = torch.full((M, N), 0.5, device=DEV)
            hip.gemm(M, N, K, A, M, 0, Bm, N, 0, C, N, compute=1, split_k=split, alpha=0.75)
            torch.cuda.synchronize()
        finally:
            for k in env:
                hip.set_tuning(k, None)
        outs.append(C.cpu())
    want = 0.5 + 0.75 * (A.double().cpu().t() @ Bm.double().cpu())
    assert float((outs[1].double() - want).abs().max()) < 2e-3 * max(1.0, float(want.abs().max()))
    assert torch.equal(outs[0], outs[1]), (float((outs[0] - outs[1]).abs().max()), int((outs[0] != outs[1]).sum()))


@pytest.mark.parametrize("rows", [None, "64", "96", "160"])
@pytest.mark.parametrize("M,N,K,bk,epi", [(600, 256, 128, 1, "res_mask"), (600, 256, 128, 0, "bias_relu"), (1333, 128, 64, 0, "plain"),
                                          (2100, 384, 192, 1, "scale_bias_res_mask_relu"), (840, 256, 2048, 0, "drop_res"),
                                          (4200, 1024, 512, 1, "bias_res"), (33, 136, 96, 0, "bias")])
def test_gemm_ring_fp32_is_bit_identical_to_the_tile_engine(hip, rows, M, N, K, bk, epi):
    """The fp32 form of the ring kernel (csrc/gemm_ring.h gemm_ring_f32_kernel: exact-f32 MFMA, TM x TN independent accumulators per
    wave) against the 4-wave fp32 engine: every accumulator sees k in the same ascending pairs and the MFMA is an fmaf chain, so the
    outputs must be IDENTICAL bits -- both weight layouts ([n][k]: 16-byte chunk + lane-half select; [k][n]: one dword per MFMA
    step), ragged M / N, every epilogue item, several row pitches.  The engine itself is pinned against fp64."""
    torch.manual_seed(M + N + K + bk)
    A = g(torch.randn(M, K))
    Bm = g(torch.randn(N, K) / K ** 0.5) if bk else g(torch.randn(K, N) / K ** 0.5)
    bias, scale = g(torch.randn(N)), g(torch.rand(N) + 0.5)
    res, msk = g(torch.randn(M, N)), g(torch.randn(M, N))
    step = torch.tensor([0x13579bd, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=DEV)
    kw = dict(compute=0)
    if "bias" in epi:
        kw.update(bias=bias)
    if "scale" in epi:
        kw.update(scale=scale, alpha=0.75)
    if "relu" in epi:
        kw.update(act=1)
    if "res" in epi:
        kw.update(residual=res, ldr=N)
    if "mask" in epi:
        kw.update(mask=msk, ldmask=N)
    if "drop" in epi:
        kw.update(dropout_p=0.1, dropout_seed=11, dropout_step=step)
    outs = []
    for ring in ("1", "2"):
        env = {"DETR_HIP_GEMM_RING": ring, "DETR_HIP_RING_ROWS": rows}
        for k, v in env.items():
            hip.set_tuning(k, v)
        try:
            C = torch.full((M, N), 7.0, device=DEV)
            hip.gemm(M, N, K, A, K, 1, Bm, Bm.stride(0), bk, C, N, **kw)
            torch.cuda.synchronize()
        finally:
            for k in env:
                hip.set_tuning(k, None)
        outs.append(C.cpu())
    ring_out, tile_out = outs
    want = A.double().cpu() @ (Bm.double().cpu().t() if bk else Bm.double().cpu())
    if "scale" in epi:
        want = want * scale.double().cpu()
    if "bias" in epi:
        want = want + bias.double().cpu()
    if "scale" in epi:
        want = want * 0.75
    if "drop" not in epi:
        if "res" in epi:
            want = want + res.double().cpu()
        if "relu" in epi:
            want = torch.relu(want)
        if "mask" in epi:
            want = torch.where(msk.double().cpu() > 0, want, torch.zeros_like(want))
        assert float((tile_out.double() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    assert float(tile_out.abs().max()) > 0
    assert torch.equal(ring_out, tile_out), (float((ring_out - tile_out).abs().max()), int((ring_out != tile_out).sum()), ring_out.numel())
