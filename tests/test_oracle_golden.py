"""CPU tests: the oracle against the golden vectors written by the reference's own code, and the
oracle's internal consistency (known answers, dense == CSR, torch twin == numpy, gradcheck).

tests/golden/preproc_*.npz were produced by oracle/make_fixtures.py, which imports the reference's
utils/process.py; they pin the integer / mask preprocessing bit-exactly.  The floating-point
attention math has no reference-side vectors (TensorFlow absent, no reference tests): for it these
tests pin the restatement against analytic answers only ("parity unpinned").
"""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import ref_numpy, ref_torch
from tests.util import layer_params, random_graph

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "preproc_*.npz")))
DENSE_GOLDEN = [g for g in GOLDEN if "random200" not in g]


def _bitwise_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def test_golden_files_present():
    assert len(GOLDEN) >= 8


@pytest.mark.parametrize("impl", ["oracle", "product"])
@pytest.mark.parametrize("path", DENSE_GOLDEN, ids=[os.path.basename(p) for p in DENSE_GOLDEN])
def test_preprocessing_matches_reference_bit_exactly(path, impl):
    if impl == "oracle":
        mod = ref_numpy
    else:
        from gat_amd import preprocess as mod
    z = np.load(path)
    a = z["adj_dense"]
    size = int(z["size"])
    idx, data, shape = mod.preprocess_adj_bias(sp.csr_matrix(a))
    assert _bitwise_equal(idx, z["pab_indices"]) and idx.dtype == np.int32
    assert _bitwise_equal(data, z["pab_data"]) and data.dtype == np.float32
    assert tuple(shape) == tuple(z["pab_shape"])
    for nhood, key in ((1, "bias_n1"), (2, "bias_n2")):
        b = mod.adj_to_bias(a[None].copy(), [size], nhood=nhood)
        assert _bitwise_equal(b, z[key]), "%s nhood=%d" % (path, nhood)   # includes -0.0 vs 0.0 and the border
    f = mod.preprocess_features(sp.csr_matrix(z["feat_dense"]))
    f = f[0] if isinstance(f, tuple) else f
    assert _bitwise_equal(np.asarray(f), z["feat_norm"])


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_random200_matches_reference(impl):
    if impl == "oracle":
        mod = ref_numpy
    else:
        from gat_amd import preprocess as mod
    z = np.load([g for g in GOLDEN if "random200" in g][0])
    adj = sp.coo_matrix((z["adj_data"], (z["adj_row"], z["adj_col"])), shape=(200, 200)).tocsr()
    idx, data, shape = mod.preprocess_adj_bias(adj)
    assert _bitwise_equal(idx, z["pab_indices"]) and _bitwise_equal(data, z["pab_data"])
    feats = sp.coo_matrix((z["feat_data"], (z["feat_row"], z["feat_col"])), shape=(200, 1000)).tocsr()
    f = mod.preprocess_features(feats)
    f = f[0] if isinstance(f, tuple) else f
    assert _bitwise_equal(np.asarray(f), z["feat_norm"])


def test_mask_values_are_exactly_zero_or_minus_1e9():
    z = np.load(DENSE_GOLDEN[0])
    b = ref_numpy.adj_to_bias(z["adj_dense"][None], [z["adj_dense"].shape[0]], 1)
    assert set(np.unique(b).tolist()) <= {-1e9, 0.0}
    assert np.signbit(b[b == 0.0]).all()          # the reference produces -0.0


# ---- floating point restatement: internal consistency ------------------------------------------
def _dense_mask(n, rowptr, col):
    m = np.full((1, n, n), -1e9)
    for i in range(n):
        m[0, i, col[rowptr[i]:rowptr[i + 1]]] = 0.0
    return m


def test_dense_and_csr_restatements_agree_in_float64():
    n, F, H, D = 60, 9, 4, 5
    dst, src = random_graph(n, 4.0, seed=3)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    p = layer_params(F, H, D, seed=1, res=True)
    x = np.random.default_rng(2).normal(0, 1, (n, F))
    for act, red, res in ((ref_numpy.elu, "concat", True), (ref_numpy.identity, "mean", False)):
        a = ref_numpy.gat_layer_csr(x, p, rowptr, col, act, red, res)
        b = ref_numpy.gat_layer_dense(x[None], p, _dense_mask(n, rowptr, col), act, red, res)[0]
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)


def test_float32_mask_underflows_to_exact_zero():
    """In fp32 exp(-1e9 + x - max) is exactly 0, so the dense masked path equals the sparse one."""
    n, F, D = 30, 5, 4
    dst, src = random_graph(n, 3.0, seed=9)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    p = {k: v[..., 0, :] if v.ndim == 3 else v[0] for k, v in layer_params(F, 1, D, seed=4).items()}
    x = np.random.default_rng(5).normal(0, 1, (1, n, F)).astype(np.float32)
    out, coefs, _ = ref_numpy.attn_head_dense(x, p, _dense_mask(n, rowptr, col).astype(np.float32), return_coefs=True)
    mask = _dense_mask(n, rowptr, col)[0] < 0
    assert (coefs[0][mask] == 0.0).all()
    np.testing.assert_allclose(coefs[0].sum(-1), 1.0, rtol=1e-6)


def test_known_answers():
    n, F, H, D = 25, 4, 2, 3
    dst, src = random_graph(n, 3.0, seed=1)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    p = layer_params(F, H, D, seed=0)
    x = np.random.default_rng(1).normal(0, 1, (n, F))
    # zero attention vectors and biases => uniform coefficients => neighbour mean + bias
    q = dict(p, a1=np.zeros((H, D)), a2=np.zeros((H, D)), b1=np.zeros(H), b2=np.zeros(H))
    out = ref_numpy.gat_layer_csr(x, q, rowptr, col, ref_numpy.identity, "concat")
    wh = x @ p["W"].reshape(F, H * D)
    ref = np.stack([wh[col[rowptr[i]:rowptr[i + 1]]].mean(0) for i in range(n)]) + p["bias"].reshape(-1)
    np.testing.assert_allclose(out, ref, rtol=1e-12)
    # the two scalar logit biases do not cancel (LeakyReLU sits before the softmax)
    o1 = ref_numpy.gat_layer_csr(x, p, rowptr, col, ref_numpy.identity, "concat")
    o2 = ref_numpy.gat_layer_csr(x, dict(p, b1=p["b1"] + 0.7), rowptr, col, ref_numpy.identity, "concat")
    assert np.abs(o1 - o2).max() > 1e-6
    # relabelling the nodes permutes the output rows
    perm = np.random.default_rng(3).permutation(n)
    inv = np.argsort(perm)
    rp2, col2 = ref_numpy.coo_to_csr(np.stack([inv[dst], inv[src]], 1), n)
    o3 = ref_numpy.gat_layer_csr(x[perm], p, rp2, col2, ref_numpy.identity, "concat")
    np.testing.assert_allclose(o3, o1[perm], rtol=1e-10, atol=1e-12)


def test_residual_is_a_no_op_when_widths_agree():
    """utils/layers.py:31-32 assigns the sum to a dead name."""
    n, F = 12, 6
    dst, src = random_graph(n, 2.0, seed=4)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    p = {k: v[..., 0, :] if v.ndim == 3 else v[0] for k, v in layer_params(F, 1, F, seed=4).items()}
    x = np.random.default_rng(5).normal(0, 1, (n, F))
    a = ref_numpy.sp_attn_head_csr(x, p, rowptr, col, residual=True)
    b = ref_numpy.sp_attn_head_csr(x, p, rowptr, col, residual=False)
    assert (a == b).all()


def test_torch_twin_matches_numpy_and_passes_gradcheck():
    n, F, H, D = 14, 5, 2, 3
    dst, src = random_graph(n, 2.5, seed=6)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    p = layer_params(F, H, D, seed=2, res=True)
    x = np.random.default_rng(7).normal(0, 1, (n, F))
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    trp, tcol = torch.tensor(rowptr), torch.tensor(col)
    for act, nact, red, res in (("elu", ref_numpy.elu, "concat", True), ("identity", ref_numpy.identity, "mean", False)):
        a = ref_torch.layer_csr(tx, tp, trp, tcol, act, red, res).detach().numpy()
        b = ref_numpy.gat_layer_csr(x, p, rowptr, col, nact, red, res)
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)
    keys = sorted(tp)

    def fn(xx, *ws):
        return ref_torch.layer_csr(xx, dict(zip(keys, ws)), trp, tcol, "elu", "concat", True)
    assert torch.autograd.gradcheck(fn, (tx, *[tp[k] for k in keys]), eps=1e-6, atol=1e-6)
    # dense twin
    m = torch.tensor(_dense_mask(n, rowptr, col))
    d = ref_torch.layer_dense(tx[None], tp, m, "elu", "concat", True)[0].detach().numpy()
    np.testing.assert_allclose(d, ref_numpy.gat_layer_csr(x, p, rowptr, col, ref_numpy.elu, "concat", True), rtol=1e-12, atol=1e-13)


def test_two_layer_composition():
    """models/sp_gat.py:7-34: concat hidden heads, mean of output heads."""
    n, F, C = 40, 11, 7
    dst, src = random_graph(n, 3.0, seed=8)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    l0 = layer_params(F, 8, 8, seed=1)
    l1 = layer_params(64, 1, C, seed=2)
    x = np.random.default_rng(3).normal(0, 1, (n, F))
    out = ref_numpy.gat_inference(x, [l0, l1], ("csr", rowptr, col))
    assert out.shape == (n, C)
    h = ref_numpy.gat_layer_csr(x, l0, rowptr, col, ref_numpy.elu, "concat")
    ref = ref_numpy.gat_layer_csr(h, l1, rowptr, col, ref_numpy.identity, "mean")
    np.testing.assert_allclose(out, ref)
    t = ref_torch.inference_csr(torch.tensor(x), [{k: torch.tensor(v) for k, v in l.items()} for l in (l0, l1)],
                                torch.tensor(rowptr), torch.tensor(col)).numpy()
    np.testing.assert_allclose(t, out, rtol=1e-11, atol=1e-12)


def test_base_gattn_losses_and_metrics():
    """The plain-arithmetic helpers of models/base_gattn.py against hand-computed values."""
    import torch
    from gat_amd.models import BaseGAttN as M
    logits = torch.tensor([[2.0, 0.0], [0.0, 1.0], [3.0, 1.0], [0.5, 0.7]])
    labels = torch.tensor([[1.0, 0.0], [0.0, 1.0], [0.0, 1.0], [1.0, 0.0]])
    ce = -(labels * torch.log_softmax(logits, -1)).sum(-1).mean()
    assert torch.allclose(M.loss(logits, labels), ce)
    a, ra = torch.tensor([[1.0, -2.0]]), torch.tensor([[0.5, -1.0]])
    d, rd = torch.tensor([[2.0, 0.0]]), torch.tensor([[1.0, 1.0]])
    assert torch.allclose(M.bi_loss(logits, labels, a, ra, d, rd, 0.5), ce + 0.5 * (0.625 + 0.1 * 1.0))
    assert torch.allclose(M.AE_classify_loss(logits, labels, a, ra, d, rd, 0.5), ce + 0.5 * (0.625 + 0.1 * 1.0))      # |d| weights: (1*2 + 1*0)/2 = 1
    assert torch.allclose(M.AE_regression_loss(logits, labels, a, ra, d, rd, 2.0),
                          ((labels - logits) ** 2).mean() + 2.0 * ((0.25 * 1 + 1.0 * 2) / 2 + 1.0))
    assert torch.allclose(M.syn_loss(logits, labels, a, ra, 2.0), ((labels - logits) ** 2).mean() + 2.0 * 0.625)
    x = torch.tensor([1.0, 2.0, 3.0, 4.0])
    assert abs(float(M.syn_accuracy(x, 2 * x + 1)) - 1.0) < 1e-5 and abs(float(M.syn_accuracy(x, -x)) + 1.0) < 1e-5
    assert float(M.accuracy(logits, labels)) == 0.5
    assert M.confmat(logits, labels.argmax(1)).tolist() == [[1, 1], [1, 1]]
    f, p, r = M.micro_f1(logits, labels)
    assert float(p) == 0.5 and float(r) == 0.5 and abs(float(f) - 0.5) < 1e-6
    assert float(M.MSE(torch.tensor([[1.0], [3.0]]), torch.tensor([0.0, 1.0]))) == 2.5
    assert float(M.MAE(torch.tensor([[1.0], [3.0]]), torch.tensor([0.0, 1.0]))) == 1.5
    lg, lb = M.preshape(torch.zeros(2, 3, 4), torch.zeros(2, 3), 4)
    assert tuple(lg.shape) == (6, 4) and tuple(lb.shape) == (6,)
