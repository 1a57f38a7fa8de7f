"""GPU tests of the drop-in surface: attn_head / sp_attn_head, GAT.inference / SpGAT.inference
(models/gat.py:7-31, models/sp_gat.py:7-34) at the Cora hyper-parameters, and the device graph
builder against the host one."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import ref_numpy
from tests.util import assert_close, random_graph

pytestmark = pytest.mark.gpu


def _export(store):
    """Weights of a VariableStore as oracle layer dicts (float64)."""
    out = []
    for m in store.layers:
        d = {k: getattr(m, k).detach().cpu().numpy().astype(np.float64) for k in ("W", "a1", "a2", "b1", "b2", "bias")}
        if m.has_res:
            d["W_res"] = m.W_res.detach().cpu().numpy().astype(np.float64)
            d["b_res"] = m.b_res.detach().cpu().numpy().astype(np.float64)
        out.append(d)
    return out


def _randomise_biases(store, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in store.layers:
            for k in ("b1", "b2", "bias"):
                t = getattr(m, k)
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)


def test_spgat_inference_cora_shape():
    from gat_amd import layers, preprocess
    from gat_amd.models import SpGAT
    dev = torch.device("cuda:0")
    n, F, C = 400, 50, 7
    rng = np.random.RandomState(0)
    adj = sp.random(n, n, density=0.01, random_state=rng, format="csr")
    adj = adj + adj.T
    feats = preprocess.preprocess_features(sp.random(n, F, density=0.1, random_state=rng, format="csr"))[0]
    triple = preprocess.preprocess_adj_bias(adj)
    x = torch.tensor(np.asarray(feats)[None], dtype=torch.float32, device=dev)
    store = layers.VariableStore()
    with layers.variable_scope(store):
        logits = SpGAT.inference(x, C, n, False, 0.0, 0.0, bias_mat=triple, hid_units=[8], n_heads=[8, 1],
                                 activation=layers.elu, residual=False)
        _randomise_biases(store, 1)
        logits = SpGAT.inference(x, C, n, False, 0.0, 0.0, bias_mat=triple, hid_units=[8], n_heads=[8, 1],
                                 activation=layers.elu, residual=False)
    assert tuple(logits.shape) == (1, n, C) and len(store.layers) == 2
    rowptr, col = ref_numpy.coo_to_csr(triple[0], n)
    ref = ref_numpy.gat_inference(np.asarray(feats, dtype=np.float64), _export(store), ("csr", rowptr, col))
    assert_close(logits[0].detach().cpu().numpy(), ref, what="SpGAT.inference")
    # gradients flow to every variable
    logits.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in store.parameters())


def test_gat_inference_dense_three_layers_with_residual():
    from gat_amd import layers, preprocess
    from gat_amd.models import GAT
    dev = torch.device("cuda:0")
    B, n, F, C = 2, 90, 20, 5
    rng = np.random.default_rng(3)
    adj = (rng.random((B, n, n)) < 0.08).astype(np.float64)
    bias = preprocess.adj_to_bias(adj, [n] * B, nhood=1)
    x = rng.normal(0, 1, (B, n, F))
    store = layers.VariableStore()
    kw = dict(hid_units=[8, 16], n_heads=[4, 2, 2], activation=layers.elu, residual=True)
    with layers.variable_scope(store):
        GAT.inference(torch.tensor(x, dtype=torch.float32, device=dev), C, n, False, 0.0, 0.0,
                      bias_mat=torch.tensor(bias, dtype=torch.float32, device=dev), **kw)
        _randomise_biases(store, 2)
        out = GAT.inference(torch.tensor(x, dtype=torch.float32, device=dev), C, n, False, 0.0, 0.0,
                            bias_mat=torch.tensor(bias, dtype=torch.float32, device=dev), **kw)
    assert tuple(out.shape) == (B, n, C) and len(store.layers) == 3
    assert store.layers[1].has_res and not store.layers[0].has_res and not store.layers[2].has_res
    ref = ref_numpy.gat_inference(x, _export(store), ("dense", bias), residual=True)
    assert_close(out.detach().cpu().numpy(), ref, rtol=2e-5, atol_scale=4e-6, what="GAT.inference")


def test_attn_head_and_sp_attn_head_single_head_calls():
    from gat_amd import layers, preprocess
    dev = torch.device("cuda:0")
    n, F, D = 120, 14, 8
    rng = np.random.RandomState(5)
    adj = sp.random(n, n, density=0.03, random_state=rng, format="csr")
    triple = preprocess.preprocess_adj_bias(adj)
    x = np.random.default_rng(6).normal(0, 1, (1, n, F))
    xt = torch.tensor(x, dtype=torch.float32, device=dev)
    store = layers.VariableStore()
    with layers.variable_scope(store):
        a = layers.sp_attn_head(xt, D, triple, layers.elu, n)
        store.begin()
        # the dense head on the equivalent mask, sharing the same variable
        bm = np.full((1, n, n), -1e9, dtype=np.float32)
        bm[0, triple[0][:, 0], triple[0][:, 1]] = 0.0
        b = layers.attn_head(xt, D, torch.tensor(bm, device=dev), layers.elu)
    assert len(store.layers) == 1 and tuple(a.shape) == (1, n, D) == tuple(b.shape)
    rowptr, col = ref_numpy.coo_to_csr(triple[0], n)
    ref = ref_numpy.gat_layer_csr(x[0], _export(store)[0], rowptr, col, ref_numpy.elu, "concat")
    assert_close(a[0].detach().cpu().numpy(), ref, what="sp_attn_head")
    assert_close(b[0].detach().cpu().numpy(), ref, what="attn_head")


def test_device_graph_builder_matches_host_builder():
    from gat_amd import CSRGraph
    dev = torch.device("cuda:0")
    n = 5000
    dst, src = random_graph(n, 6.0, seed=7, hubs=4, hub_deg=1500)
    h = CSRGraph.from_coo(dst, src, n, dev, chunk=128)
    d = CSRGraph.from_device_coo(torch.tensor(dst, dtype=torch.int64, device=dev), torch.tensor(src, dtype=torch.int64, device=dev),
                                 n, n, chunk=128)
    assert (h.rowptr_host == d.rowptr_host).all() and (h.colptr_host == d.colptr_host).all()
    for name in ("col", "rowidx", "perm", "items_csr", "split_csr", "items_csc", "split_csc"):
        assert torch.equal(getattr(h, name), getattr(d, name)), name


def test_cpu_tensors_are_rejected_loudly():
    from gat_amd import CSRGraph, ops
    n = 10
    dst, src = random_graph(n, 2.0, seed=1)
    g = CSRGraph.from_coo(dst, src, n, torch.device("cuda:0"))
    z = torch.zeros
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.sparse_gat_layer(z(n, 4), z(4, 1, 4), z(1, 4), z(1, 4), z(1), z(1), z(1, 4), g, ops.make_cfg(1, 4))


def test_head_counts_outside_the_fused_shapes_run_in_groups():
    """6 heads (BNF_AE.py:18) is not a power of two: the layer runs as a 4-head and a 2-head sweep."""
    from gat_amd import CSRGraph
    from gat_amd.layers import GATHeads, _head_groups
    from tests.util import layer_params
    assert _head_groups(6, 8) == [(0, 4), (4, 2)] and _head_groups(8, 8) == [(0, 8)] and _head_groups(3, 64) == [(0, 2), (2, 1)]
    dev = torch.device("cuda:0")
    n, F, H, D = 150, 12, 6, 8
    dst, src = random_graph(n, 4.0, seed=3)
    p = layer_params(F, H, D, seed=9)
    m = GATHeads(F, H, D, device=dev)
    m.load(p)
    x = np.random.default_rng(1).normal(0, 1, (n, F))
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    g = CSRGraph.from_coo(dst, src, n, dev)
    out = m(xt, g, "elu", "concat")
    assert_close(out.detach().cpu().numpy(), ref_numpy.gat_layer_csr(x, p, rowptr, col, ref_numpy.elu, "concat"), what="6 heads concat")
    out.sum().backward()
    assert m.W.grad is not None and torch.isfinite(m.W.grad).all() and (m.W.grad.abs().sum((0, 2)) > 0).all()
    mean = m(xt, g, "identity", "mean")
    assert_close(mean.detach().cpu().numpy(), ref_numpy.gat_layer_csr(x, p, rowptr, col, ref_numpy.identity, "mean"), what="6 heads mean")
    bm = np.full((1, n, n), -1e9, dtype=np.float32)
    bm[0, dst, src] = 0.0
    dense = m(xt[None], torch.tensor(bm, device=dev), "elu", "concat")
    assert_close(dense[0].detach().cpu().numpy(), out.detach().cpu().numpy(), rtol=2e-6, atol_scale=1e-6, what="6 heads dense")


def test_training_step_replays_from_a_hip_graph():
    """The C ABI only enqueues on the caller's stream and allocates nothing, so a whole
    forward + backward can be captured once and replayed (small graphs are launch-bound)."""
    from gat_amd import CSRGraph
    from gat_amd.layers import GATHeads
    from tests.util import layer_params
    dev = torch.device("cuda:0")
    n, F, H, D = 300, 16, 8, 8
    dst, src = random_graph(n, 5.0, seed=4, hubs=1, hub_deg=280)
    g = CSRGraph.from_coo(dst, src, n, dev, chunk=64)
    m = GATHeads(F, H, D, device=dev)
    m.load(layer_params(F, H, D, seed=5))
    x = torch.randn(n, F, device=dev)
    dout = torch.randn(n, H * D, device=dev)
    static_out = torch.empty(n, H * D, device=dev)

    def step():
        for p in m.parameters():
            p.grad = None
        out = m(x, g, "elu", "concat")
        out.backward(dout)
        static_out.copy_(out.detach())
    step()
    eager = (static_out.clone(), m.W.grad.clone(), m.a1.grad.clone())
    from gat_amd import graphs
    replay, _ = graphs.capture(step, warmup=1)
    x.copy_(torch.randn(n, F, device=dev))        # new input, then back: replays must track the buffers
    replay()
    assert not torch.equal(static_out, eager[0])
    x.copy_(torch.randn(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(0)))
    replay()
    a = static_out.clone()
    step()
    torch.cuda.synchronize()
    assert torch.equal(a, static_out)


def test_gat_inference_accepts_a_prebuilt_graph_for_bias_mat():
    """GAT.inference on a batch of small graphs with bias_mat replaced by the block-diagonal graph
    built once from it (CSRGraph.from_bias_mat): same logits, mean over the output heads included."""
    from gat_amd import CSRGraph, layers, preprocess
    from gat_amd.models import GAT
    dev = torch.device("cuda:0")
    B, N, F, C = 6, 40, 11, 5
    rng = np.random.default_rng(0)
    adj = (rng.random((B, N, N)) < 0.1).astype(np.float64)
    bm = torch.tensor(preprocess.adj_to_bias(adj, [N] * B, 1), dtype=torch.float32, device=dev)
    x = torch.tensor(rng.normal(0, 1, (B, N, F)), dtype=torch.float32, device=dev)
    store = layers.VariableStore()
    kw = dict(hid_units=[8, 8], n_heads=[4, 2, 2], activation=layers.elu, residual=True)
    with layers.variable_scope(store):
        a = GAT.inference(x, C, N, False, 0.0, 0.0, bias_mat=bm, **kw)
        _randomise_biases(store, 2)
        a = GAT.inference(x, C, N, False, 0.0, 0.0, bias_mat=bm, **kw)
        b = GAT.inference(x, C, N, False, 0.0, 0.0, bias_mat=CSRGraph.from_bias_mat(bm), **kw)
    assert tuple(b.shape) == (B, N, C)
    assert_close(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=5e-6, atol_scale=2e-6, what="graph for bias_mat")
