"""Sparse node features (the sparse tuple of preprocess_features, utils/process.py:155-162) as the
input of the first sparse attention layer: Wh = X.W over stored entries, dW = X^T.dWh over the
transposed CSR.  Checked against the float64 oracle run on the dense matrix (same tolerance rule
as the CSR layer tests) and against the dense-input kernels."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import ref_numpy, ref_torch
from tests.test_models_gpu import _export, _randomise_biases
from tests.util import assert_close, layer_params, random_graph

pytestmark = pytest.mark.gpu


def _features(n, F, density, seed, empty_rows=True):
    rng = np.random.RandomState(seed)
    x = sp.random(n, F, density=density, random_state=rng, format="lil", dtype=np.float32)
    if empty_rows:
        x[3, :] = 0            # a node without features: its row sum is 0 and stays 0
        x[n - 1, :] = 0
    return x.tocsr()


@pytest.mark.parametrize("n,F,H,D,activation,reduce", [(300, 143, 8, 8, "elu", "concat"), (257, 50, 1, 7, "identity", "mean"),
                                                        (120, 1433, 8, 8, "elu", "concat"), (90, 20, 4, 16, "elu", "concat")])
def test_layer_on_sparse_features_matches_the_oracle(n, F, H, D, activation, reduce):
    from gat_amd import CSRGraph, ops, preprocess
    from gat_amd.features import SparseFeatures
    dev = torch.device("cuda:0")
    xs = _features(n, F, 0.05, seed=n)
    x = xs.toarray().astype(np.float64)
    dst, src = random_graph(n, 5.0, seed=F)
    p = layer_params(F, H, D, seed=2)
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    graph = CSRGraph.from_coo(dst, src, n, dev)
    feats = SparseFeatures.from_tuple(preprocess.sparse_to_tuple(xs), dev)
    assert feats.shape == (n, F) and feats.nnz == xs.nnz
    assert_close(feats.to_dense().cpu().numpy(), x, rtol=0, atol_scale=0, what="to_dense")
    cfg = ops.make_cfg(H, D, activation, reduce)
    out = ops.sparse_gat_layer(feats, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph, cfg)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    rp = {}
    for k, v in p.items():
        if k in ("b1", "b2"):
            v = np.broadcast_to(v, (n, v.shape[0])).copy()
        rp[k] = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.layer_csr(torch.tensor(x), rp, torch.tensor(rowptr), torch.tensor(col), activation, reduce)
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what="sparse-feature forward")
    dout = np.random.default_rng(1).normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    ref.backward(torch.tensor(dout))
    for k in ("W", "a1", "a2", "bias"):
        assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), atol_scale=5e-6, what="sparse-feature d" + k)
    for k in ("b1", "b2"):
        terms = rp[k].grad.numpy()
        tol = 1e-5 * np.abs(terms.sum(0)) + 1e-6 * np.abs(terms).sum(0)
        assert (np.abs(tp[k].grad.cpu().numpy() - terms.sum(0)) <= tol).all(), k
    # the dense-input path computes the same layer
    dense = ops.sparse_gat_layer(torch.tensor(x, dtype=torch.float32, device=dev), tp["W"], tp["a1"], tp["a2"], tp["b1"],
                                 tp["b2"], tp["bias"], graph, cfg)
    assert_close(out.detach().cpu().numpy(), dense.detach().cpu().numpy(), rtol=2e-6, atol_scale=1e-6, what="sparse vs dense input")


def test_row_normalisation_on_the_device_matches_preprocess_features():
    from gat_amd import preprocess
    from gat_amd.features import SparseFeatures
    dev = torch.device("cuda:0")
    xs = _features(500, 300, 0.03, seed=4)
    dense_ref, tup_ref = preprocess.preprocess_features(xs)
    got = SparseFeatures.from_scipy(xs, dev).row_normalize()
    assert_close(got.to_dense().cpu().numpy(), np.asarray(dense_ref), rtol=1e-6, atol_scale=0, what="row-normalised features")
    assert torch.isfinite(got.val).all()
    rows = got.to_dense().sum(1).cpu().numpy()
    nonempty = np.asarray(xs.sum(1)).ravel() != 0
    assert np.allclose(rows[nonempty], 1.0, atol=1e-5) and (rows[~nonempty] == 0).all()


def test_spgat_inference_accepts_sparse_features():
    from gat_amd import layers, preprocess
    from gat_amd.features import SparseFeatures
    from gat_amd.models import SpGAT
    dev = torch.device("cuda:0")
    n, F, C = 400, 600, 7
    rng = np.random.RandomState(0)
    adj = sp.random(n, n, density=0.01, random_state=rng, format="csr")
    adj = adj + adj.T
    dense, tup = preprocess.preprocess_features(sp.random(n, F, density=0.02, random_state=rng, format="csr"))
    triple = preprocess.preprocess_adj_bias(adj)
    feats = SparseFeatures.from_tuple(tup, dev)
    store = layers.VariableStore()
    kw = dict(bias_mat=triple, hid_units=[8], n_heads=[8, 1], activation=layers.elu, residual=False)
    with layers.variable_scope(store):
        SpGAT.inference(feats, C, n, False, 0.0, 0.0, **kw)
        _randomise_biases(store, 1)
        logits = SpGAT.inference(feats, C, n, False, 0.0, 0.0, **kw)
        dense_logits = SpGAT.inference(torch.tensor(np.asarray(dense)[None], dtype=torch.float32, device=dev), C, n,
                                       False, 0.0, 0.0, **kw)
    assert tuple(logits.shape) == (1, n, C)
    rowptr, col = ref_numpy.coo_to_csr(triple[0], n)
    ref = ref_numpy.gat_inference(np.asarray(dense, dtype=np.float64), _export(store), ("csr", rowptr, col))
    assert_close(logits[0].detach().cpu().numpy(), ref, what="SpGAT.inference on sparse features")
    assert_close(logits.detach().cpu().numpy(), dense_logits.detach().cpu().numpy(), rtol=2e-6, atol_scale=1e-6, what="sparse vs dense features")
    logits.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in store.parameters())
    # training mode: dropout on the stored feature values keeps the pattern and the call runs
    with layers.variable_scope(store):
        tr = SpGAT.inference(feats, C, n, True, 0.6, 0.6, **kw)
    assert torch.isfinite(tr).all()
