"""GPU parity tests of the f32x3 compute mode (detr_gemm_desc.compute = 2; csrc/gemm_core.h: mma_ktile_split3): fp32 storage and fp32
ACCURACY on the bf16 matrix pipe -- every operand value is split exactly into three bf16 values and a product is the sum of the six
largest bf16 partial products, accumulated in fp32.  The bar is the exact-fp32 kernels' own: the same tolerances as tests/test_gpu_kernels.py
(2e-5 of the output scale against fp64), and, measured directly, an error against fp64 no larger than the exact fp32 MFMA kernel's.
Reference lines the kernels stand in for: resnet_backbone.py:116-137 (convolutions), custom_layers.py Linear, transformer.py:285-356."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def g(t):
    return t.to(DEV).contiguous()


@pytest.fixture(autouse=True)
def _split_every_shape(hip):
    """The product dispatch keeps the exact kernel for the shapes where the split is slower (gemm_f32.hip: gemm_split3_shape);
    the tests force the split kernel onto every 64x64 / 128x128 launch unless they say otherwise."""
    hip.set_tuning("DETR_HIP_SPLIT3_ALL", 1)
    yield
    for k in ("DETR_HIP_SPLIT3_ALL", "DETR_HIP_SPLIT3_T128", "DETR_HIP_GEMM_TILE", "DETR_HIP_CONV_TILE", "DETR_HIP_WGRAD_TILE", "DETR_HIP_X3_T192"):
        hip.set_tuning(k, None)


def close(a, b, rtol=2e-5, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert err <= rtol * scale, f"{what}: max abs err {err:.3e} > {rtol * scale:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("M,N,K,bk,epi", [(33600, 256, 1024, 1, False), (1000, 300, 147, 0, True), (33600, 256, 128, 0, True), (190, 128, 64, 1, False)])
def test_gemm_f32x3_192_row_tiles_equal_the_128_row_tiles(hip, M, N, K, bk, epi):
    """The 192 x 128 tile of the f32x3 GEMM (round 6: fewer rounds where 128 x 128 tiles overflow the chip's 512 workgroup slots by a few; K-contiguous A,
    unsplit) sums the same products in the same order: bit-identical to the 128 x 128 tiles, with ragged edges and a residual + ReLU epilogue.  The rule
    picks it for M33600 N256 (526 tiles of 128 x 128) by itself."""
    torch.manual_seed(M + N + K)
    A, B = g(torch.randn(M, K)), g(torch.randn(N, K) if bk else torch.randn(K, N))
    R = g(torch.randn(M, N)) if epi else None
    bias = g(torch.randn(N)) if epi else None
    outs = {}
    for mode in (2, 1, 0):                # never / wherever eligible / the rule
        hip.set_tuning("DETR_HIP_X3_T192", mode if mode else None)
        C = torch.full((M, N), 3.0, device=DEV)
        hip.gemm(M, N, K, A, K, 1, B, K if bk else N, bk, C, N, compute=2, **(dict(residual=R, ldr=N, bias=bias, act=1) if epi else {}))
        torch.cuda.synchronize()
        outs[mode] = C
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[0], outs[2])
    ref = A.double() @ (B.double().t() if bk else B.double())
    if epi:
        ref = torch.relu(ref + bias.double() + R.double())
    close(outs[1], ref, what=f"f32x3 gemm 192-row tiles {M}x{N}x{K}")


@pytest.mark.parametrize("tile", [None, 1, 2, 3])       # the dispatch's choice, 128x128 / 128x64 / 64x64 forced
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 92, 256), (1000, 256, 147), (77, 40, 33), (8400, 64, 64), (520, 2048, 256), (130, 32, 100)])
@pytest.mark.parametrize("ak,bk", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_layouts_f32x3(hip, tile, M, N, K, ak, bk):
    """Every operand layout, ragged shapes, padded leading dimensions: 2e-5 of the output scale against fp64 (the exact kernels' bound)."""
    if tile is not None:
        hip.set_tuning("DETR_HIP_GEMM_TILE", tile)
    torch.manual_seed(M * 7 + N * 3 + K + ak * 2 + bk)
    A, B = torch.randn(M, K), torch.randn(K, N)
    ref = A.double() @ B.double()
    lda, ldb = (K if ak else M) + 4, (K if bk else N) + 4
    Am, Bm = torch.zeros((M, lda) if ak else (K, lda)), torch.zeros((N, ldb) if bk else (K, ldb))
    if ak:
        Am[:, :K] = A
    else:
        Am[:, :M] = A.t()
    if bk:
        Bm[:, :K] = B.t()
    else:
        Bm[:, :N] = B
    C = torch.full((M, N + 4), 7.0, device=DEV)
    hip.gemm(M, N, K, g(Am), lda, ak, g(Bm), ldb, bk, C, N + 4, compute=2)
    torch.cuda.synchronize()
    close(C[:, :N], ref, what=f"f32x3 gemm {M}x{N}x{K} ak={ak} bk={bk}")
    assert float((C[:, N:] - 7.0).abs().max()) == 0.0, "gemm wrote outside its columns"


def test_gemm_epilogue_and_split_k_f32x3(hip):
    torch.manual_seed(2)
    M, N, K = 333, 200, 96
    A, W = torch.randn(M, K), torch.randn(N, K)
    scale, bias = torch.rand(N) + 0.5, torch.randn(N)
    R, Mk = torch.randn(M, N), torch.randn(M, N)
    acc = A.double() @ W.double().t()
    for act in (0, 1, 2):
        ref = (acc * scale.double() + bias.double()) * 0.25 + R.double()
        ref = ref.clamp_min(0) if act == 1 else (torch.sigmoid(ref) if act == 2 else ref)
        ref = torch.where(Mk.double() > 0, ref, torch.zeros_like(ref))
        C = torch.zeros(M, N, device=DEV)
        hip.gemm(M, N, K, g(A), K, 1, g(W), K, 1, C, N, alpha=0.25, scale=g(scale), bias=g(bias), residual=g(R), ldr=N, mask=g(Mk), ldmask=N,
                 act=act, compute=2)
        close(C, ref, what=f"f32x3 gemm epilogue act={act}")
    # split-K through the workspace (tile-ordered slabs + reduce launch): deterministic, and a fused bias gradient (row sums of A)
    hip.ensure_workspace(DEV)
    M, N, K = 256, 1024, 8400
    A, B = torch.randn(K, M), torch.randn(K, N)
    C0 = torch.randn(M, N)
    outs = []
    for rep in range(2):
        C, rs = g(C0.clone()), torch.zeros(M, device=DEV)
        hip.gemm(M, N, K, g(A), M, 0, g(B), N, 0, C, N, alpha=0.5, split_k=8, rowsum_a=rs, rowsum_alpha=0.5, compute=2)
        outs.append((C.clone(), rs.clone()))
    close(outs[0][0], C0.double() + 0.5 * (A.double().t() @ B.double()), rtol=5e-5, what="f32x3 split-K")
    close(outs[0][1], 0.5 * A.double().sum(0), rtol=5e-5, what="f32x3 split-K row sums")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "workspace split-K must be deterministic"


@pytest.mark.parametrize("M,N,K,ak,bk", [(8400, 256, 2048, 1, 1), (4096, 1024, 256, 1, 0), (256, 1024, 8400, 0, 0), (800, 256, 256, 1, 1)])
def test_f32x3_is_as_accurate_as_the_exact_fp32_mfma(hip, M, N, K, ak, bk):
    """The claim itself: against an fp64 product of the same fp32 operands, the rms and the maximum error of compute = 2 are no larger than
    1.25x those of compute = 0 (measured: 0.8-1.0x, scripts/micro_split3.py) -- and three orders of magnitude below bf16 MFMA inputs."""
    torch.manual_seed(M + N + K)
    A = g(torch.randn(M, K) if ak else torch.randn(K, M))
    B = g((torch.randn(N, K) if bk else torch.randn(K, N)) / K ** 0.5)
    ref = (A if ak else A.t()).double() @ (B.t() if bk else B).double()
    err = {}
    for mode in (0, 2, 1):
        C = torch.zeros(M, N, device=DEV)
        hip.gemm(M, N, K, A, A.stride(0), ak, B, B.stride(0), bk, C, N, compute=mode)
        d = C.double() - ref
        err[mode] = (float(d.pow(2).mean().sqrt()), float(d.abs().max()))
    assert err[2][0] <= 1.25 * err[0][0] and err[2][1] <= 1.5 * err[0][1], err
    assert err[2][0] <= 1e-3 * err[1][0], err


def test_gemm_group_f32x3_matches_individual_launches(hip):
    torch.manual_seed(52)
    hip.ensure_workspace(DEV)
    M, D = 8400, 256
    xs = [g(torch.randn(M, D)) for _ in range(3)]
    W, bias = g(torch.randn(3 * D, D) / 16), g(torch.randn(3 * D))
    old = hip.COMPUTE_BF16
    hip.COMPUTE_BF16 = 2
    try:
        calls = lambda outs: [hip.linear_fwd_call(xs[i], W[i * D:(i + 1) * D], bias[i * D:(i + 1) * D], outs[i], alpha=(0.5 if i == 0 else 1.0)) for i in range(3)]
        ref, grp = [torch.zeros(M, D, device=DEV) for _ in range(3)], [torch.zeros(M, D, device=DEV) for _ in range(3)]
        for a, kw in calls(ref):
            hip.gemm(*a, **kw)
        hip.gemm_group(calls(grp))
        for i in range(3):
            assert torch.equal(ref[i], grp[i]), f"grouped forward member {i} differs"
            close(ref[i], (xs[i].double() @ W[i * D:(i + 1) * D].double().t() + bias[i * D:(i + 1) * D].double()) * (0.5 if i == 0 else 1.0), what="group fwd")
        dys = [g(torch.randn(M, D)) for _ in range(3)]
        wg = lambda dws, dbs: [hip.linear_wgrad_call(dys[i], xs[i], dws[i], bias_grad=dbs[i]) for i in range(3)]
        dw_r, db_r = [torch.zeros(D, D, device=DEV) for _ in range(3)], [torch.zeros(D, device=DEV) for _ in range(3)]
        dw_g, db_g = [torch.zeros(D, D, device=DEV) for _ in range(3)], [torch.zeros(D, device=DEV) for _ in range(3)]
        for a, kw in wg(dw_r, db_r):
            hip.gemm(*a, **kw)
        hip.gemm_group(wg(dw_g, db_g))
        for i in range(3):
            assert torch.equal(dw_r[i], dw_g[i]) and torch.equal(db_r[i], db_g[i]), f"grouped wgrad member {i} differs"
            close(dw_r[i], dys[i].double().t() @ xs[i].double(), rtol=5e-5, what="group wgrad")
    finally:
        hip.COMPUTE_BF16 = old


@pytest.mark.parametrize("tile", [None, 1, 3])
@pytest.mark.parametrize("N,H,W,Ci,Co,stride", [(2, 13, 17, 64, 64, 1), (1, 20, 27, 128, 128, 2), (2, 9, 11, 16, 32, 1), (1, 50, 67, 64, 64, 1), (1, 25, 42, 256, 256, 2),
                                                (2, 50, 84, 256, 256, 1)])
def test_conv3x3_all_modes_f32x3(hip, tile, N, H, W, Ci, Co, stride):
    """3x3 convolution forward (+ folded BN + ReLU), input gradient (+ mask) and weight gradient (+ scale) at the exact kernels' tolerances."""
    if tile is not None:
        hip.set_tuning("DETR_HIP_CONV_TILE", tile)
        hip.set_tuning("DETR_HIP_WGRAD_TILE", tile)
    hip.ensure_workspace(DEV)
    torch.manual_seed(N + H + W + Ci + stride)
    x = torch.randn(N, H, W, Ci, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(3, 3, Ci, Co, dtype=torch.float64) / (3 * Ci ** 0.5)).requires_grad_(True)
    scale, shift = torch.rand(Co, dtype=torch.float64) + 0.5, torch.randn(Co, dtype=torch.float64)
    z = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), None, stride=stride, padding=1).permute(0, 2, 3, 1)
    Ho, Wo = z.shape[1], z.shape[2]
    y = torch.relu(z * scale + shift)
    xd, wd = g(x.detach().float()), g(w.detach().float())
    yd = torch.zeros(N, Ho, Wo, Co, device=DEV)
    hip.conv3x3(0, xd, wd, yd, N, H, W, Ci, Ho, Wo, Co, stride, scale=g(scale.float()), bias=g(shift.float()), act=1, compute=2)
    close(yd, y, what="f32x3 conv3x3 fwd")
    dz = torch.randn_like(z)
    z.backward(dz)
    dzd = g(dz.float())
    mask = torch.randn(N, H, W, Ci)
    dxd = torch.zeros(N, H, W, Ci, device=DEV)
    hip.conv3x3(1, dzd, wd, dxd, N, H, W, Ci, Ho, Wo, Co, stride, mask=g(mask), compute=2)
    close(dxd, x.grad * (mask.double() > 0), what="f32x3 conv3x3 dgrad(+mask)")
    dws = []
    for rep in range(2):
        dwd = torch.zeros(3, 3, Ci, Co, device=DEV)
        hip.conv3x3(2, xd, dzd, dwd, N, H, W, Ci, Ho, Wo, Co, stride, scale=g(scale.float()), compute=2)
        dws.append(dwd)
    close(dws[0], w.grad * scale, rtol=5e-5, what="f32x3 conv3x3 wgrad(+scale)")
    assert torch.equal(dws[0], dws[1])


@pytest.mark.parametrize("N,H,W,C", [(8, 50, 84, 256), (2, 21, 37, 128)])
def test_conv3x3_f32x3_192_row_tiles_equal_the_128_row_tiles(hip, N, H, W, C):
    """The 192 x 128 and 128 x 64 tiles of the f32x3 3x3 convolution (forward and input gradient; the rule takes 192 rows at 8 x 50 x 84 x 256, where
    128 x 128 tiles are 526 workgroups on 512 slots, and 128 x 64 for a single 128-channel panel): same products in the same order, bit-identical
    to the 128 x 128 tiles."""
    torch.manual_seed(N + H + W + C)
    x, w = g(torch.randn(N, H, W, C)), g(torch.randn(3, 3, C, C) / (3 * C ** 0.5))
    shift, mask = g(torch.randn(C)), g(torch.randn(N, H, W, C))
    hip.set_tuning("DETR_HIP_CONV_TILE", None)
    outs = {}
    for mode in (2, 1, 3, 0):             # 128 x 128 / 192 x 128 / 128 x 64 tiles / the rule
        hip.set_tuning("DETR_HIP_X3_T192", mode if mode else None)
        hip.set_tuning("DETR_HIP_SPLIT3_T128", 1)          # (128-row tiles from one tile on, so that the small case compares the two tile heights too)
        y, dx = torch.zeros(N, H, W, C, device=DEV), torch.zeros(N, H, W, C, device=DEV)
        hip.conv3x3(0, x, w, y, N, H, W, C, H, W, C, 1, bias=shift, act=1, compute=2)
        hip.conv3x3(1, x, w, dx, N, H, W, C, H, W, C, 1, mask=mask, compute=2)
        torch.cuda.synchronize()
        outs[mode] = (y, dx)
    for k in (0, 1):
        assert torch.equal(outs[1][k], outs[2][k]) and torch.equal(outs[0][k], outs[2][k]) and torch.equal(outs[3][k], outs[2][k])
    ref = torch.relu(F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=1).permute(0, 2, 3, 1) + shift.double())
    close(outs[1][0], ref, what="f32x3 conv3x3 fwd, 192-row tiles")


@pytest.mark.parametrize("N,H,W", [(2, 37, 53), (1, 128, 160)])
def test_stem_conv_f32x3(hip, N, H, W):
    torch.manual_seed(N + H + W)
    hip.ensure_workspace(DEV)
    img = torch.randn(N, H, W, 3, dtype=torch.float64)
    w = (torch.randn(7, 7, 3, 64, dtype=torch.float64) / 12.0).requires_grad_(True)
    scale, shift = torch.rand(64, dtype=torch.float64) + 0.5, torch.randn(64, dtype=torch.float64)
    z = F.conv2d(F.pad(img.permute(0, 3, 1, 2), (3, 3, 3, 3)), w.permute(3, 2, 0, 1), None, stride=2).permute(0, 2, 3, 1)
    Ho, Wo = z.shape[1], z.shape[2]
    y = torch.relu(z * scale + shift)
    imgd, sd = g(img.float()), g(scale.float())
    ws = g((w.detach() * scale).float().reshape(147, 64))
    yd = torch.zeros(N, Ho, Wo, 64, device=DEV)
    hip.stem_conv(0, imgd, ws, yd, N, H, W, Ho, Wo, bias=g(shift.float()), act=1, compute=2)
    close(yd, y, what="f32x3 stem conv fwd")
    dz = torch.randn(N, Ho, Wo, 64, dtype=torch.float64)
    (z * dz).sum().backward()
    res = []
    for rep in range(2):
        dw = torch.zeros(147, 64, device=DEV)
        hip.stem_conv(2, imgd, g(dz.float()), dw, N, H, W, Ho, Wo, scale=sd, split=7, compute=2)
        res.append(dw)
    close(res[0].view(7, 7, 3, 64), w.grad * scale, rtol=5e-5, what="f32x3 stem conv wgrad")
    assert torch.equal(res[0], res[1])


def test_product_dispatch_keeps_the_exact_kernel_where_the_split_is_slower(hip):
    """Without DETR_HIP_SPLIT3_ALL the narrow / short-K / 64-row shapes run the exact kernel: bit-identical to compute = 0."""
    hip.set_tuning("DETR_HIP_SPLIT3_ALL", None)
    torch.manual_seed(9)
    for M, N, K, same in ((4096, 64, 256, True), (4096, 256, 64, True), (64, 256, 4096, True), (4096, 256, 256, False)):
        A, B = g(torch.randn(M, K)), g(torch.randn(N, K))
        C0, C2 = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
        hip.gemm(M, N, K, A, K, 1, B, K, 1, C0, N, compute=0)
        hip.gemm(M, N, K, A, K, 1, B, K, 1, C2, N, compute=2)
        assert torch.equal(C0, C2) == same, (M, N, K)
        close(C2, A.double() @ B.double().t(), what=f"{M}x{N}x{K}")
