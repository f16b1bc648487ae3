"""CPU tests (-m "not gpu"): the oracle against SciPy and analytic identities, the host logic,
and that the C-ABI library loads and exports every symbol include/detr_hip.h declares."""
import ctypes
import math
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lsap_lib():
    so = os.path.join(ROOT, "oracle", "_build", "liblsap_oracle.so")
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "oracle", "lsap.c"), "-o", so, "-lm"])
    return ctypes.CDLL(so)


def _solve(lib, C):
    nr, nc = C.shape
    C = np.ascontiguousarray(C, np.float64)
    a = np.zeros(min(nr, nc), np.int64)
    b = np.zeros(min(nr, nc), np.int64)
    rc = lib.lsap_oracle_solve(nr, nc, C.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p),
                               b.ctypes.data_as(ctypes.c_void_p))
    return rc, a, b


def test_c_lsap_restatement_pinned_to_scipy(lsap_lib):
    """oracle/lsap.c == scipy.optimize.linear_sum_assignment (the reference's own dependency),
    bit-exact indices incl. tie cases, ragged shapes and the DETR shapes 100 x n."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(0)
    shapes = [(100, n) for n in (1, 2, 7, 20, 50, 99)] + [(300, 99), (5, 9), (9, 5), (1, 1), (64, 64)]
    shapes += [(int(rng.integers(1, 130)), int(rng.integers(1, 130))) for _ in range(150)]
    for t, (nr, nc) in enumerate(shapes):
        C = rng.normal(size=(nr, nc)).astype(np.float32).astype(np.float64)
        if t % 4 == 0:
            C = np.round(C * 2) / 2
        rc, a, b = _solve(lsap_lib, C)
        r, c = linear_sum_assignment(C)
        assert rc == 0 and np.array_equal(a, r) and np.array_equal(b, c), (nr, nc)


def test_c_lsap_invalid_inputs(lsap_lib):
    from scipy.optimize import linear_sum_assignment
    C = np.ones((4, 3))
    C[1, 1] = np.nan
    assert _solve(lsap_lib, C)[0] == -2
    with pytest.raises(ValueError):
        linear_sum_assignment(C)
    C = np.ones((4, 3))
    C[:, 2] = np.inf
    assert _solve(lsap_lib, C)[0] == -1
    with pytest.raises(ValueError):
        linear_sum_assignment(C)


def test_box_identities():
    from oracle import set_loss_ref as L
    b = torch.tensor([[0.5, 0.5, 0.2, 0.4], [0.1, 0.9, 0.4, 0.4]])
    xy = L.xcycwh_to_xy_min_xy_max(b)
    assert torch.allclose(xy[0], torch.tensor([0.4, 0.3, 0.6, 0.7]))
    assert torch.allclose(xy[1], torch.tensor([0.0, 0.7, 0.3, 1.0]))          # clipped (bbox.py:182)
    g = L.giou_matrix(xy, xy)
    assert torch.allclose(torch.diagonal(g), torch.ones(2), atol=1e-6)          # GIoU(x,x) = 1
    assert float(g.min()) >= -1.0 and float(g.max()) <= 1.0 + 1e-6
    yx = L.xcycwh_to_yx_min_yx_max(b)
    assert torch.allclose(yx[:, [1, 0, 3, 2]], xy)


def test_loss_identities():
    """CE of uniform logits = ln C; total = sum of 1/2/5-weighted parts; permutation invariance."""
    from oracle import set_loss_ref as L
    B, Q, C = 2, 100, 92
    tb, tc = L.make_targets(B, seed=3, force_full=False)
    torch.manual_seed(0)
    boxes = torch.rand(B, Q, 4) * 0.5 + 0.1
    out = {"pred_logits": torch.zeros(B, Q, C), "pred_boxes": boxes}
    total, losses = L.get_losses(out, torch.tensor(tb), torch.tensor(tc), 91)
    assert abs(float(losses["label_cost"]) - math.log(C)) < 1e-5
    assert abs(float(total) - float(losses["label_cost"] + 2 * losses["giou_loss"] + 5 * losses["l1_loss"])) < 1e-5
    logits = torch.randn(B, Q, C)
    out = {"pred_logits": logits, "pred_boxes": boxes}
    t1, _ = L.get_losses(out, torch.tensor(tb), torch.tensor(tc), 91)
    perm = torch.randperm(Q)
    out2 = {"pred_logits": logits[:, perm], "pred_boxes": boxes[:, perm]}
    t2, _ = L.get_losses(out2, torch.tensor(tb), torch.tensor(tc), 91)
    assert abs(float(t1) - float(t2)) < 1e-4 * abs(float(t1))
    # aux handling: 6 levels -> 36 entries, suffixes _0.._4 (loss.py:27-29,172-179)
    out3 = dict(out, aux=[out] * 5)
    t3, l3 = L.get_losses(out3, torch.tensor(tb), torch.tensor(tc), 91)
    assert len(l3) == 36 and "l1_loss_4" in l3 and abs(float(t3) - 6 * float(t1)) < 1e-3


def test_matching_double_swap_semantics():
    """after the two name swaps t_indices index targets, p_indices predictions."""
    from oracle import set_loss_ref as L
    tb, tc = L.make_targets(1, seed=9, force_full=False)
    n = int(tb[0, 0, 0])
    boxes = torch.rand(100, 4)
    boxes[:n] = torch.tensor(tb[0, 1:1 + n])        # prediction i sits exactly on target i
    logits = torch.zeros(100, 92)
    ti, pi, sel, tbs, tcs = L.hungarian_matching(torch.tensor(tb[0]), torch.tensor(tc[0]), boxes, logits)
    assert torch.equal(ti, torch.arange(n)) and torch.equal(pi, torch.arange(n))
    assert sel.shape[0] == 100 and int(sel.sum()) == n and tbs.shape == (n, 4)


def test_oracle_forward_shapes_and_groups():
    from oracle import detr_ref as R, optim_ref as O
    P = R.to_torch(R.make_params(0, num_enc=1, num_dec=2))
    out = R.detr_forward(torch.randn(1, 64, 96, 3), P, num_enc=1, num_dec=2)
    assert out["pred_logits"].shape == (1, 100, 92) and out["pred_boxes"].shape == (1, 100, 4) and len(out["aux"]) == 1
    assert float(out["pred_boxes"].min()) > 0 and float(out["pred_boxes"].max()) < 1
    names = list(R.param_shapes())
    n_train = sum(int(np.prod(s)) for k, s in R.param_shapes().items() if R.trainable(k))
    assert 41.0e6 < n_train < 41.7e6                               # ~41.5 M trainable scalars
    assert O.variable_group("input_proj/kernel") == "backbone" and O.variable_group("query_embed/kernel") == "backbone"
    assert O.variable_group("class_embed/bias") == "transformers"
    assert O.variable_group("transformer/decoder/norm/gamma") == "transformers"
    assert O.variable_group("cls_layer/kernel") == "nlayers"
    assert not R.trainable("backbone/layer1/0/bn1/weight") and R.trainable("backbone/layer1/0/conv1/kernel")
    assert len(names) == len(set(names))


def test_position_embedding_formula():
    """y=(i+1)/(H+1e-6)*2pi, dim_t=10000^(2*floor(k/2)/128), sin even / cos odd, [pos_y, pos_x]."""
    from oracle import detr_ref as R
    H, W = 5, 7
    pos = R.position_embedding_sine(1, H, W)[0]
    i, j, k = 3, 2, 10
    dim = 10000.0 ** (2 * (k // 2) / 128)
    y = (i + 1) / (H + 1e-6) * 2 * math.pi
    x = (j + 1) / (W + 1e-6) * 2 * math.pi
    assert abs(float(pos[i, j, k]) - math.sin(y / dim)) < 1e-5
    assert abs(float(pos[i, j, k + 1]) - math.cos(y / dim)) < 1e-5
    assert abs(float(pos[i, j, 128 + k]) - math.sin(x / dim)) < 1e-5


def test_adam_clipnorm_oracle():
    from oracle import optim_ref as O
    g = np.full((4,), 3.0, np.float32)                       # ||g|| = 6 > 0.1
    c, n = O.clip_by_norm(g, 0.1)
    assert abs(n - 6.0) < 1e-6 and abs(float(np.linalg.norm(c)) - 0.1) < 1e-6
    c2, _ = O.clip_by_norm(g * 1e-3, 0.1)
    assert np.allclose(c2, g * 1e-3)
    p = {"w": np.zeros(4, np.float32)}
    opt = O.Adam(1e-2, clipnorm=None)
    opt.apply({"w": np.ones(4, np.float32)}, p)
    assert np.allclose(p["w"], -1e-2, rtol=1e-5)              # first Adam step = -lr*sign(g)
    st = {}
    p = {"w": np.zeros(1, np.float32)}
    opt = O.Adam(1.0, clipnorm=None)
    for step in range(4):                                     # agg=2: apply at steps 1 and 3 only
        O.aggregate_and_apply(st, "backbone", opt, {"w": np.ones(1, np.float32)}, p, step, 2, True)
        assert opt.t == (step + 1) // 2


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads (no GPU needed) and exports every function of include/detr_hip.h;
    the ctypes table of the host package covers exactly the same set."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
    from detr_tf import _hip
    hdr = open(os.path.join(ROOT, "include", "detr_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(detr_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = _hip.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in detr_hip.h but not exported"
    assert sorted(_hip.EXPORTED_SYMBOLS) == declared
    assert lib.detr_hip_abi_version() == _hip.ABI_VERSION == 9


def _hip_lib():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
    from detr_tf import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _hip, _hip.load()


def test_ctypes_mirrors_match_the_library_struct_layouts():
    """Boundary hygiene: every descriptor struct mirrored by hand in detr_tf/_hip.py has the size and the
    field offsets the library was compiled with (detr_hip_struct_layout); a deliberately drifted mirror is caught."""
    import ctypes
    _hip, lib = _hip_lib()
    for which, S in enumerate(_hip.LAYOUT_STRUCTS):
        buf = (ctypes.c_int32 * 96)()
        n = lib.detr_hip_struct_layout(which, buf, 96)
        assert n == 1 + len(S._fields_), (S.__name__, n)
        assert list(buf[:n]) == _hip.struct_layout_mirror(S), S.__name__
    assert lib.detr_hip_struct_layout(99, (ctypes.c_int32 * 4)(), 4) < 0

    class Drifted(ctypes.Structure):          # detr_reduce_desc with `rows` widened: every later offset moves
        _fields_ = [(n, (ctypes.c_int64 if n == "rows" else t)) for n, t in _hip.ReduceDesc._fields_]
    buf = (ctypes.c_int32 * 96)()
    n = lib.detr_hip_struct_layout(0, buf, 96)
    assert list(buf[:n]) != _hip.struct_layout_mirror(Drifted)


def test_workspace_bytes_queries_and_tuning_reload():
    """detr_hip_workspace_bytes_* (SURVEY 8b): pure host arithmetic on the descriptors (no GPU): a split-K GEMM needs
    effective_splits * (M*N [+ M with a fused bias gradient]) floats, forward convs none, the nine-tap weight gradient
    splits * 9*Ci*Co floats; the tuning variables are re-read only by detr_hip_reload_tuning()."""
    from ctypes import byref
    _hip, lib = _hip_lib()
    d = _hip.GemmDesc()
    d.M, d.N, d.K, d.batch, d.split_k, d.compute = 256, 1024, 33600, 1, 32, 1
    assert lib.detr_hip_workspace_bytes_gemm(byref(d)) == 32 * 256 * 1024 * 4
    d.rowsum_a = 16                                  # (any non-null pointer: never dereferenced)
    assert lib.detr_hip_workspace_bytes_gemm(byref(d)) == 32 * (256 * 1024 + 256) * 4
    d.rowsum_a, d.split_k = None, 1
    assert lib.detr_hip_workspace_bytes_gemm(byref(d)) == 0
    d.split_k, d.K = 1000, 64 * 32                   # more splits than K tiles: the library runs 64
    assert lib.detr_hip_workspace_bytes_gemm(byref(d)) == 64 * 256 * 1024 * 4
    # round 4: the partial slabs are tile-ordered images of the MFMA accumulators -- whole 64x64 / 128x128 tiles, so a ragged output
    # needs a little more than split*M*N floats (DETR_HIP_SLAB_TS=2 keeps row-major slabs: the documented minimum)
    d.M, d.N, d.K, d.split_k = 100, 92, 64 * 32, 8
    assert lib.detr_hip_workspace_bytes_gemm(byref(d)) == 8 * (2 * 2 * 64 * 64) * 4
    _hip.set_tuning("DETR_HIP_SLAB_TS", 2)
    assert lib.detr_hip_workspace_bytes_gemm(byref(d)) == 8 * 100 * 92 * 4
    _hip.set_tuning("DETR_HIP_SLAB_TS", None)
    d.M = 0
    assert lib.detr_hip_workspace_bytes_gemm(byref(d)) < 0
    c = _hip.Conv3x3Desc()
    c.N, c.Hi, c.Wi, c.Ci, c.Ho, c.Wo, c.Co, c.stride, c.pad, c.compute = 8, 50, 84, 256, 50, 84, 256, 1, 1, 1
    assert lib.detr_hip_workspace_bytes_conv3x3(byref(c), 0) == 0 and lib.detr_hip_workspace_bytes_conv3x3(byref(c), 1) == 0
    fused = lib.detr_hip_workspace_bytes_conv3x3(byref(c), 2)
    assert fused == 32 * 9 * 256 * 256 * 4           # fused nine-tap kernel: 512 workgroups over 16 (ci, co) tiles
    os.environ["DETR_HIP_WGRAD_FUSED"] = "2"
    try:
        assert lib.detr_hip_workspace_bytes_conv3x3(byref(c), 2) == fused        # not re-read on the call path ...
        assert lib.detr_hip_reload_tuning() == 0
        per_tap = lib.detr_hip_workspace_bytes_conv3x3(byref(c), 2)              # ... only on request
        assert per_tap != fused and per_tap % (9 * 256 * 256 * 4) == 0
    finally:
        os.environ.pop("DETR_HIP_WGRAD_FUSED", None)
        lib.detr_hip_reload_tuning()
    # round 5: the nine-tap kernel's stride-2 form is taken for bf16-STORED tensors up to 128 channels (the query follows the launch's choice);
    # DETR_HIP_WGRAD_FUSED = 3 / 4 force the per-tap / the nine-tap kernel for every stride-2 weight gradient
    s2 = _hip.Conv3x3Desc()
    s2.N, s2.Hi, s2.Wi, s2.Ci, s2.Ho, s2.Wo, s2.Co, s2.stride, s2.pad, s2.compute = 8, 200, 334, 128, 100, 167, 128, 2, 1, 1
    s2.x_dtype = s2.w_dtype = 1                      # (mode 2: `w` carries dy)
    nine_tap = 127 * 9 * 128 * 128 * 4               # 4800 units of 32 output pixels over 512 / 4 (ci, co) tiles = 128 splits of 38 units: 127 non-empty
    assert lib.detr_hip_workspace_bytes_conv3x3(byref(s2), 2) == nine_tap
    s2.x_dtype = s2.w_dtype = 0                      # fp32-stored: per-tap kernel
    per_tap_128 = lib.detr_hip_workspace_bytes_conv3x3(byref(s2), 2)
    assert per_tap_128 != nine_tap and per_tap_128 % (9 * 128 * 128 * 4) == 0
    s2.x_dtype = s2.w_dtype = 1
    s2.Ci = s2.Co = 256
    s2.Hi, s2.Wi, s2.Ho, s2.Wo = 100, 167, 50, 84
    per_tap_256 = lib.detr_hip_workspace_bytes_conv3x3(byref(s2), 2)
    _hip.set_tuning("DETR_HIP_WGRAD_FUSED", 4)
    forced = lib.detr_hip_workspace_bytes_conv3x3(byref(s2), 2)
    _hip.set_tuning("DETR_HIP_WGRAD_FUSED", None)
    assert forced == 32 * 9 * 256 * 256 * 4 and per_tap_256 != forced
    s2.Ci = s2.Co = 128
    s2.Hi, s2.Wi, s2.Ho, s2.Wo = 200, 334, 100, 167
    _hip.set_tuning("DETR_HIP_WGRAD_FUSED", 3)
    assert lib.detr_hip_workspace_bytes_conv3x3(byref(s2), 2) == per_tap_128
    _hip.set_tuning("DETR_HIP_WGRAD_FUSED", None)
    st = _hip.StemDesc()
    st.N, st.H, st.W, st.Ho, st.Wo, st.compute, st.w_dtype, st.split = 8, 800, 1333, 400, 667, 1, 1, 512
    assert lib.detr_hip_workspace_bytes_stem(byref(st), 0) == 0
    assert lib.detr_hip_workspace_bytes_stem(byref(st), 2) == 768 * 147 * 64 * 4
    ln = _hip.LayerNormDesc()
    ln.rows, ln.C = 8400, 256
    assert lib.detr_hip_workspace_bytes_layernorm(byref(ln)) == 512 * 2 * 256 * 4


def test_fused_1x1_backward_sizing_and_rejections():
    """detr_hip_conv1x1_bwd_fused_bf16 (ABI 7): the workspace query is host arithmetic (one 64 x 256 slab per workgroup, at most 256), and every
    argument the kernel is not built for is REJECTED before anything is launched (no GPU needed: the pointers below are never dereferenced) --
    the caller keeps the two-launch form for those."""
    import ctypes
    _hip, lib = _hip_lib()
    q = lib.detr_hip_conv1x1_bwd_fused_workspace_floats
    assert q(534400) == 256 * 64 * 256 and q(32) == 64 * 256 and q(33) == 2 * 64 * 256 and q(8191) == 256 * 64 * 256
    f = lib.detr_hip_conv1x1_bwd_fused_bf16
    P = 1 << 20                                      # "pointers": 16-byte aligned, never touched
    ok = dict(dy=P, ldg=256, a=P, lda=64, w=P, ldw=256, da=P, ldda=64, use_mask=1, dw=P, lddw=256, scale=None, alpha=1.0, M=4096, d1=64, d2=256,
              ws=P, wsf=256 * 64 * 256, stream=None)

    def call(**kw):
        v = dict(ok, **kw)
        return f(v["dy"], v["ldg"], v["a"], v["lda"], v["w"], v["ldw"], v["da"], v["ldda"], v["use_mask"], v["dw"], v["lddw"], v["scale"],
                 ctypes.c_float(v["alpha"]), v["M"], v["d1"], v["d2"], v["ws"], v["wsf"], v["stream"])
    for bad, what in ((dict(d1=128), "built for"), (dict(d2=512), "built for"), (dict(ldg=255), "leading"), (dict(lda=60), "leading"),
                      (dict(dy=P + 2), "alignment"), (dict(ws=P + 4), "alignment"), (dict(wsf=64 * 256 * 127), "workspace"), (dict(M=0), "bad operands"),
                      (dict(dw=None), "bad operands"), (dict(M=1 << 24, wsf=1 << 40), "descriptor")):
        assert call(**bad) != 0, bad
        msg = lib.detr_hip_last_error().decode()
        assert what in msg, (bad, msg)


def test_host_side_of_the_c_abi_under_address_sanitizer():
    """SURVEY section 5 stance: the host side of the boundary (descriptor validation, planning, scratch-size queries, layout
    self-check) built with -fsanitize=address and driven by tests/host_abi_check.c (plain C, no GPU): every rejection path
    returns a negative code without touching memory it was not given; ASan aborts on any stack / heap error."""
    import subprocess
    pkg = os.path.join(ROOT, "detr-tensorflow_amd")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    r = subprocess.run(["make", "-C", pkg, "-j", str(os.cpu_count() or 2), "asan-check"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "host_abi_check: 0 failed expectation(s)" in r.stdout


def test_profile_plumbing_of_the_bench_roofline():
    """bench.py's `roofline.traffic` comes from profiles/r03_traffic_bf16.json, which scripts/pmc_summary.py writes by folding rocprofv3
    kernel names into the family names detr_tf/_hip.py reports.  A kernel body that gets a new name (as the 64-deep K-tile variant
    did) must still fold into its family, or the field silently turns null: every GEMM kernel of the committed kernel-stats profile
    folds, the traffic file has the bench's dominant family, and the per-launch traffic is of the order of the algorithmic bytes."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    try:
        import pmc_summary
    finally:
        sys.path.pop(0)
    names = []
    with open(os.path.join(root, "profiles", "r03_rocprofv3_kernel_stats_bf16.txt")) as f:
        for line in f:
            m = re.search(r"(void )?detr::\w+.*", line)
            if m and "detr::gemm_" in line:
                names.append(m.group(0))
    assert len(names) >= 10
    assert all(pmc_summary.fold(n) in ("gemm_bf16c_kernel", "gemm_f32_kernel", "gemm_stream_bf16_kernel") for n in names), \
        [n for n in names if pmc_summary.fold(n) is None]
    assert pmc_summary.fold("detr::layernorm_fwd_kernel(float const*)") is None
    traffic = json.load(open(os.path.join(root, "profiles", "r03_traffic_bf16.json")))
    line = json.load(open(os.path.join(root, "profiles", "r03_bench_line.json")))
    fam = line["roofline"]["kernel"].split("detr::")[1].split(" ")[0]
    assert fam in traffic["per_symbol"], (fam, list(traffic["per_symbol"]))
    t = traffic["per_symbol"][fam]["traffic_bytes_per_launch"]
    assert 0.5 < t / line["roofline"]["algorithmic_bytes_per_launch"] < 3.0
    with open(os.path.join(root, "bench.py")) as f:
        assert "r03_traffic_bf16.json" in f.read()
