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
    assert lib.detr_hip_abi_version() == 3
