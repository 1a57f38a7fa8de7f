"""GPU parity of the batched dense masked-attention layer (utils/layers.py:6-34 on [B,N,F] with
an additive [B,N,N] bias_mat) against the float64 oracle; same tolerance rule as the CSR tests."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy, ref_torch
from tests.util import assert_close, dense_dropout_masks, layer_params

pytestmark = pytest.mark.gpu


def _masks(B, N, density, seed, soft=False):
    """adj_to_bias-style masks (0 / -1e9, self-loops kept); soft=True adds arbitrary finite biases."""
    rng = np.random.default_rng(seed)
    adj = rng.random((B, N, N)) < density
    adj |= np.eye(N, dtype=bool)[None]
    m = np.where(adj, 0.0, -1e9)
    if soft:
        m = m + np.where(adj, rng.normal(0, 1.0, (B, N, N)), 0.0)
    return m


CASES = [
    # B, N, F, H, D, activation, soft-mask
    (3, 50, 12, 8, 8, "elu", False),
    (2, 264, 16, 8, 8, "elu", False),
    (2, 77, 9, 1, 7, "identity", True),
    (1, 130, 20, 4, 16, "elu", False),
    (2, 33, 6, 1, 64, "identity", False),
    (4, 19, 5, 2, 4, "elu", True),
    (1, 1000, 10, 8, 8, "elu", False),      # too large for the LDS-resident kernels: chunked path
    (1, 530, 8, 1, 64, "identity", False),  # same for a single wide head
    (3, 300, 10, 4, 8, "elu", True),        # 32-float rows: 8 lanes per row in the list kernels, two list passes
    (1, 70, 8, 4, 64, "elu", False),        # 256-float rows: one row per wave
    (2, 90, 7, 8, 8, "relu", False),
]


@pytest.mark.parametrize("path", ["dense", "lists"])
@pytest.mark.parametrize("B,N,F,H,D,activation,soft", CASES)
def test_dense_layer_forward_and_gradients(B, N, F, H, D, activation, soft, path):
    """path: the matrix-core kernels ("dense") or the compacted-list kernels for sparse masks ("lists")."""
    from gat_amd import ops
    if path == "lists" and H * ops.padded_width(D) < 32:
        pytest.skip("the list kernels take rows of at least 32 floats")
    if path == "dense" and H * ops.padded_width(D) > 64:
        path = "auto"          # rows wider than the matrix-core kernels take: auto routes them to the list kernels
    dev = torch.device("cuda:0")
    p = layer_params(F, H, D, seed=N, res=(N == 50))
    rng = np.random.default_rng(N + 1)
    x = rng.normal(0, 1, (B, N, F))
    bm = _masks(B, N, 0.15, N + 2, soft)
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    cfg = ops.make_cfg(H, D, activation=activation, mask_path=path)
    out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"],
                              torch.tensor(bm, dtype=torch.float32, device=dev), cfg, tp.get("W_res"), tp.get("b_res"))
    act = {"elu": ref_numpy.elu, "relu": ref_numpy.relu, "identity": ref_numpy.identity}[activation]
    ref = ref_numpy.gat_layer_dense(x, p, bm, act, "concat", residual="W_res" in p)
    assert_close(out.detach().cpu().numpy(), ref, what="dense forward")

    dout = rng.normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    rp = {}
    for k, v in p.items():
        if k in ("b1", "b2"):
            v = np.broadcast_to(v, (B, N, v.shape[0])).copy()
        rp[k] = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rout = ref_torch.layer_dense(rx, rp, torch.tensor(bm), activation, "concat", residual="W_res" in p)
    rout.backward(torch.tensor(dout))
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="dense dX")
    for k in p:
        ref_g = rp[k].grad.numpy()
        got = tp[k].grad.cpu().numpy().astype(np.float64)
        if k in ("b1", "b2"):
            terms = ref_g.reshape(-1, ref_g.shape[-1])
            tol = 1e-5 * np.abs(terms.sum(0)) + 1e-6 * np.abs(terms).sum(0)
            assert (np.abs(got - terms.sum(0)) <= tol).all(), "d%s: got %s ref %s" % (k, got, terms.sum(0))
        else:
            assert_close(got, ref_g, atol_scale=5e-6, what="dense d" + k)


def test_dense_equals_sparse_on_the_same_graph():
    """attn_head with the adj_to_bias mask and sp_attn_head with the COO triple compute the same
    layer (A1 == A2): exp(-1e9 + x - max) is exactly 0 in fp32 and every row keeps its self-loop."""
    from gat_amd import CSRGraph, ops
    from tests.util import random_graph
    dev = torch.device("cuda:0")
    n, F, H, D = 200, 24, 8, 8
    dst, src = random_graph(n, 5.0, seed=12)
    p = layer_params(F, H, D, seed=3)
    x = np.random.default_rng(4).normal(0, 1, (n, F)).astype(np.float32)
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in p.items()}
    cfg = ops.make_cfg(H, D, "elu")
    xs = torch.tensor(x, device=dev)
    a = ops.sparse_gat_layer(xs, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"],
                             CSRGraph.from_coo(dst, src, n, dev), cfg)
    bm = np.full((1, n, n), -1e9, dtype=np.float32)
    bm[0, dst, src] = 0.0
    b = ops.dense_gat_layer(xs[None], tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"],
                            torch.tensor(bm, device=dev), cfg)[0]
    assert_close(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-6, atol_scale=1e-6, what="dense vs sparse")


def test_rows_masked_everywhere_and_automatic_choice():
    """A row whose mask is -1e9 in every column attends uniformly in fp32 (the -1e9 swallows the
    logits): both kernel families give that.  'auto' measures the density once per mask tensor."""
    from gat_amd import ops
    dev = torch.device("cuda:0")
    B, N, F, H, D = 2, 40, 6, 8, 8
    rng = np.random.default_rng(0)
    bm = _masks(B, N, 0.1, 3)
    bm[0, 5, :] = -1e9
    bm[1, 0, :] = -1e9
    p = layer_params(F, H, D, seed=1)
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in p.items()}
    x = torch.tensor(rng.normal(0, 1, (B, N, F)), dtype=torch.float32, device=dev)
    bt = torch.tensor(bm, dtype=torch.float32, device=dev)
    outs = {}
    for path in ("dense", "lists", "auto"):
        outs[path] = ops.dense_gat_layer(x, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], bt,
                                         ops.make_cfg(H, D, "elu", mask_path=path)).cpu().numpy()
    assert_close(outs["lists"], outs["dense"], rtol=2e-6, atol_scale=1e-6, what="lists vs dense")
    assert np.array_equal(outs["auto"], outs["lists"])          # density 0.1: auto takes the list kernels
    wh = (x.cpu().numpy().astype(np.float64) @ p["W"].reshape(F, H * D))
    uniform = ref_numpy.elu(wh[0].mean(0) + p["bias"].reshape(-1))
    assert_close(outs["lists"][0, 5], uniform, rtol=1e-5, atol_scale=1e-5, what="uniform row")
    assert ops._mask_density(bt) < 0.2
    dense_mask = torch.zeros(B, N, N, device=dev)
    assert ops._mask_density(dense_mask) == 1.0
    assert not ops._use_mask_lists(ops.make_cfg(H, D), dense_mask)
    # a batch of many small graphs (the graph-resident list kernels): lists at any density of a hard mask, not for
    # dense soft masks, not for rows that are not 64 floats
    many = torch.zeros(256, N, N, device=dev)
    many[:, :, ::2] = -1e9
    many[:, torch.arange(N), torch.arange(N)] = 0.0
    assert ops._mask_facts(many) == (pytest.approx(0.5, abs=0.02), False)
    assert ops._use_mask_lists(ops.make_cfg(H, D), many)
    assert ops._use_mask_lists(ops.make_cfg(H, D), torch.zeros(256, N, N, device=dev)) is False     # density 1.0 > 0.9
    assert not ops._use_mask_lists(ops.make_cfg(4, 8), many)
    soft = many.clone()
    soft[:, 1, 1] = -0.5
    assert ops._mask_facts(soft)[1] and not ops._use_mask_lists(ops.make_cfg(H, D), soft)


@pytest.mark.parametrize("path", ["dense", "lists"])
def test_padded_batch_with_per_graph_sizes(path):
    """Graphs of different sizes padded to one [G,N,N] batch (process_ppi.py:183-215) and masked by
    adj_to_bias(adj, sizes): the binarised block covers each graph's own nodes, padded nodes keep
    the raw (I+I) diagonal, i.e. a +1e9 bias, and therefore attend to themselves only.  Real
    nodes' outputs equal those of the unpadded graph; everything is compared with the oracle."""
    from gat_amd import ops, preprocess
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(17)
    F, H, D = 10, 8, 8
    ns = [37, 50, 12, 50, 1]
    adjs, feats = [], []
    for n in ns:
        a = (rng.random((n, n)) < 0.15).astype(np.float64)
        adjs.append(np.maximum(a, a.T))
        feats.append(rng.normal(0, 1, (n, F)))
    adj, x, sizes, node_mask = preprocess.pad_graphs(adjs, feats)
    assert adj.shape == (5, 50, 50) and (node_mask.sum(1) == ns).all()
    bm = preprocess.adj_to_bias(adj, sizes, nhood=1)
    assert bm[2, 20, 20] == 1e9 and bm[2, 20, 21] == -1e9          # padded node: +1e9 on its diagonal
    p = layer_params(F, H, D, seed=5)
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    cfg = ops.make_cfg(H, D, activation="elu", mask_path=path)
    out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"],
                              torch.tensor(bm, dtype=torch.float32, device=dev), cfg)
    ref = ref_numpy.gat_layer_dense(x, p, bm, ref_numpy.elu, "concat")
    assert_close(out.detach().cpu().numpy(), ref, what="padded forward")
    # a real node's output does not depend on the padding
    alone = ref_numpy.gat_layer_dense(feats[2][None], p, preprocess.adj_to_bias(adjs[2][None], [12]), ref_numpy.elu, "concat")
    assert_close(out.detach().cpu().numpy()[2, :12], alone[0], what="padded vs unpadded")
    # a padded node sees only itself: out = elu(Wh_i + bias) with Wh_i = 0 (zero features)
    assert_close(out.detach().cpu().numpy()[2, 12:], np.broadcast_to(ref_numpy.elu(p["bias"].reshape(-1)), (38, H * D)),
                 what="padded rows")
    dout = rng.normal(0, 1, tuple(out.shape)) * node_mask[..., None]
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    rp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rout = ref_torch.layer_dense(rx, rp, torch.tensor(bm), "elu", "concat")
    rout.backward(torch.tensor(dout))
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="padded dX")
    for k in p:
        assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), atol_scale=5e-6, what="padded d" + k)


def test_hard_mask_as_block_diagonal_graph():
    """CSRGraph.from_bias_mat: a batch of adj_to_bias masks converted once into one graph over B*N
    nodes; GATHeads then runs the CSR kernels on [B,N,F] input and matches the dense route.  Soft
    masks are refused (their biases would be lost)."""
    from gat_amd import CSRGraph
    from gat_amd.layers import GATHeads
    dev = torch.device("cuda:0")
    B, N, F, H, D = 5, 70, 9, 8, 8
    bm = torch.tensor(_masks(B, N, 0.12, 21), dtype=torch.float32, device=dev)
    x = torch.tensor(np.random.default_rng(3).normal(0, 1, (B, N, F)), dtype=torch.float32, device=dev, requires_grad=True)
    m = GATHeads(F, H, D, device=dev)
    m.load(layer_params(F, H, D, seed=4))
    graph = CSRGraph.from_bias_mat(bm)
    assert graph.n_dst == B * N and graph.nnz == int((bm > -1e8).sum()) and len(graph.split_csr) == 0
    assert (graph.items_csr[:, 0].cpu().numpy() == np.arange(B * N)).all()           # rows in their own order
    a = m(x, graph, "elu")
    ga = torch.autograd.grad(a.square().sum(), [x, m.W, m.a1])
    b = m(x, bm, "elu")
    gb = torch.autograd.grad(b.square().sum(), [x, m.W, m.a1])
    assert tuple(a.shape) == (B, N, H * D)
    assert_close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol_scale=1e-6, what="graph vs mask")
    for u, v, name in zip(ga, gb, ("dX", "dW", "da1")):
        assert_close(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-5, atol_scale=5e-6, what="graph vs mask " + name)
    with pytest.raises(ValueError):
        CSRGraph.from_bias_mat(torch.tensor(_masks(1, 20, 0.2, 2, soft=True), dtype=torch.float32, device=dev))
    # a padded batch (+1e9 on the diagonal of padded nodes, utils/process.py:14-25 with sizes) is a hard mask too
    from gat_amd import preprocess
    rng = np.random.default_rng(1)
    adjs = [np.maximum(a, a.T) for a in ((rng.random((k, k)) < 0.2).astype(np.float64) for k in (9, 20, 14))]
    adj, feats, sizes, _ = preprocess.pad_graphs(adjs, [rng.normal(0, 1, (a.shape[0], F)) for a in adjs])
    pbm = torch.tensor(preprocess.adj_to_bias(adj, sizes), dtype=torch.float32, device=dev)
    px = torch.tensor(feats, dtype=torch.float32, device=dev)
    pa, pb = m(px, CSRGraph.from_bias_mat(pbm), "elu"), m(px, pbm, "elu")
    assert_close(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-6, atol_scale=1e-6, what="padded batch as a graph")
    with pytest.raises(ValueError):
        m(x[:, :10], graph, "elu")


def test_narrow_output_head_on_one_large_sparse_graph():
    """A single 7-wide head on a 700-node graph with a 1 % mask: 'auto' zero-pads the head to 32
    floats and takes the list kernels; results equal the matrix-core path and the oracle."""
    from gat_amd import ops
    dev = torch.device("cuda:0")
    B, N, F, H, D = 1, 700, 12, 1, 7
    bm = _masks(B, N, 0.01, 5)
    p = layer_params(F, H, D, seed=6)
    x = np.random.default_rng(7).normal(0, 1, (B, N, F))
    bt = torch.tensor(bm, dtype=torch.float32, device=dev)
    assert ops._widen_for_lists(ops.make_cfg(H, D, "identity"), bt).dp == 32
    assert ops._widen_for_lists(ops.make_cfg(H, D, "identity", mask_path="dense"), bt).dp == 8
    outs, grads = {}, {}
    for path in ("auto", "dense"):
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], bt, ops.make_cfg(H, D, "identity", mask_path=path))
        out.square().sum().backward()
        outs[path], grads[path] = out.detach().cpu().numpy(), [xt.grad.cpu().numpy()] + [tp[k].grad.cpu().numpy() for k in sorted(tp)]
    assert outs["auto"].shape == (B, N, H * D)
    assert_close(outs["auto"], ref_numpy.gat_layer_dense(x, p, bm, ref_numpy.identity, "concat"), what="widened head vs oracle")
    assert_close(outs["auto"], outs["dense"], rtol=2e-6, atol_scale=1e-6, what="widened vs matrix cores")
    for a, b, name in zip(grads["auto"], grads["dense"], ["dX"] + ["d" + k for k in sorted(p)]):
        assert_close(a, b, rtol=1e-5, atol_scale=5e-6, what="widened " + name)


def test_repeated_hard_mask_is_converted_once_and_tracks_changes():
    """GATHeads keeps a memo of masks it has met: the second unchanged sighting of a sparse hard mask
    builds the block-diagonal graph, later calls use it, an in-place change of the mask drops it."""
    from gat_amd import layers
    from gat_amd.layers import GATHeads
    dev = torch.device("cuda:0")
    B, N, F, H, D = 4, 60, 8, 8, 8
    bm = torch.tensor(_masks(B, N, 0.1, 9), dtype=torch.float32, device=dev)
    x = torch.tensor(np.random.default_rng(0).normal(0, 1, (B, N, F)), dtype=torch.float32, device=dev)
    m = GATHeads(F, H, D, device=dev)
    m.load(layer_params(F, H, D, seed=2))
    layers.mask_graphs._entries.clear()
    first = m(x, bm, "elu")
    assert layers.mask_graphs.lookup(bm) is not None            # second sighting converts
    third = m(x, bm, "elu")
    assert_close(third.detach().cpu().numpy(), first.detach().cpu().numpy(), rtol=2e-6, atol_scale=1e-6, what="graph vs kernels")
    bm[0, 3, :] = 0.0                                             # node 3 of graph 0 now attends to everybody
    assert layers.mask_graphs.lookup(bm) is None                # version changed: entry dropped, dense kernels again
    changed = m(x, bm, "elu")
    assert (changed[0, 3] - first[0, 3]).abs().max() > 1e-4 and torch.allclose(changed[1:], first[1:], atol=1e-6)
    again = m(x, bm, "elu")                                       # converted anew from the changed mask
    assert_close(again.detach().cpu().numpy(), changed.detach().cpu().numpy(), rtol=2e-6, atol_scale=1e-6, what="after the change")
    soft = torch.tensor(_masks(2, 30, 0.2, 4, soft=True), dtype=torch.float32, device=dev)
    xs = x[:2, :30]
    a, b2, c = m(xs, soft, "elu"), m(xs, soft, "elu"), m(xs, soft, "elu")      # soft masks stay with the dense kernels
    assert torch.equal(a, b2) and torch.equal(b2, c)
    layers.mask_graphs.enabled = False
    try:
        assert layers.mask_graphs.lookup(bm) is None
    finally:
        layers.mask_graphs.enabled = True


def test_row_masked_everywhere_stays_with_the_dense_kernels():
    """A row that is -1e9 in every column attends uniformly in fp32 (the bias swallows the logits,
    utils/layers.py:17).  The CSR kernels would compute a true softmax over the logits instead, so
    from_bias_mat refuses such a mask and the memo of repeated masks keeps it dense: the first, second
    and third call give the same (uniform) row, equal to the forced dense and list routes."""
    from gat_amd import CSRGraph, layers, ops
    from gat_amd.layers import GATHeads
    dev = torch.device("cuda:0")
    B, N, F, H, D = 3, 48, 7, 8, 8
    bm = _masks(B, N, 0.1, 12)
    bm[1, 7, :] = -1e9
    bt = torch.tensor(bm, dtype=torch.float32, device=dev)
    with pytest.raises(ValueError):
        CSRGraph.from_bias_mat(bt)
    shifted = torch.tensor(_masks(B, N, 0.1, 12), dtype=torch.float32, device=dev)
    shifted[0, 2] += 1.0e6                                        # a shared 1e6 on a row with several kept entries
    with pytest.raises(ValueError):
        CSRGraph.from_bias_mat(shifted)
    p = layer_params(F, H, D, seed=3)
    m = GATHeads(F, H, D, device=dev)
    m.load(p)
    x = torch.tensor(np.random.default_rng(4).normal(0, 1, (B, N, F)), dtype=torch.float32, device=dev)
    layers.mask_graphs._entries.clear()
    calls = [m(x, bt, "elu").detach().cpu().numpy() for _ in range(4)]
    assert layers.mask_graphs.lookup(bt) is None                 # decided: stays dense
    for c in calls[1:]:
        assert np.array_equal(c, calls[0])
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in p.items()}
    for path in ("dense", "lists"):
        o = ops.dense_gat_layer(x, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], bt,
                                ops.make_cfg(H, D, "elu", mask_path=path)).cpu().numpy()
        assert_close(calls[0], o, rtol=2e-6, atol_scale=1e-6, what="memo route vs " + path)
    wh = x.cpu().numpy().astype(np.float64) @ p["W"].reshape(F, H * D)
    assert_close(calls[0][1, 7], ref_numpy.elu(wh[1].mean(0) + p["bias"].reshape(-1)), rtol=1e-5, atol_scale=1e-5, what="uniform row")
    # the oracle evaluated in fp32 (as TensorFlow does) swallows the logits of that row too
    assert_close(calls[0], ref_numpy.gat_layer_dense(x.cpu().numpy(), p, bm.astype(np.float32), ref_numpy.elu, "concat"),
                 rtol=1e-5, atol_scale=1e-5, what="fp32 oracle")


@pytest.mark.parametrize("B,N,F,H,D,soft", [(256, 52, 6, 8, 8, False), (260, 264, 8, 8, 8, False), (256, 72, 5, 4, 8, True),
                                             (300, 132, 7, 1, 64, False), (256, 8, 3, 2, 16, True)])
def test_graph_resident_kernels_for_batches_of_many_small_graphs(B, N, F, H, D, soft):
    """Batches of at least 256 graphs whose rows are a multiple of 16 bytes take the graph-resident form
    of the list kernels (one block per graph, the graph's projected rows staged in LDS).  Graphs are
    independent, so the float64 oracle checks the first and the last graphs of the batch, forward and
    gradients of the inputs; the whole batch is compared with the matrix-core kernels."""
    from gat_amd import ops
    dev = torch.device("cuda:0")
    bm = _masks(B, N, 0.1, 40 + N, soft=soft).astype(np.float32)
    bm[5, N // 2, :] = -1e9                                   # a row masked everywhere: uniform attention in fp32 (checked against the matrix-core kernels)
    p = layer_params(F, H, D, seed=N)
    x = np.random.default_rng(N + 1).normal(0, 1, (B, N, F)).astype(np.float32)
    dout = np.random.default_rng(N + 2).normal(0, 1, (B, N, H * D)).astype(np.float32)
    res = {}
    for path in ("lists", "dense"):
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, device=dev, requires_grad=True)
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], torch.tensor(bm, device=dev),
                                  ops.make_cfg(H, D, "elu", mask_path=path))
        out.backward(torch.tensor(dout, device=dev))
        res[path] = (out.detach().cpu().numpy(), xt.grad.cpu().numpy(), tp["W"].grad.cpu().numpy(), tp["a1"].grad.cpu().numpy(),
                     tp["a2"].grad.cpu().numpy(), tp["bias"].grad.cpu().numpy(), tp["b1"].grad.cpu().numpy(), tp["b2"].grad.cpu().numpy())
    # the bias gradient of the graph-resident backward is summed per graph and then over the graphs in two stages
    for a, b, name in zip(res["lists"], res["dense"], ("out", "dX", "dW", "da1", "da2", "dbias", "db1", "db2")):
        assert_close(a, b, rtol=1e-5, atol_scale=5e-6, what="graph-resident vs matrix cores " + name)
    for sl in (slice(0, 3), slice(B - 2, B)):
        rp = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
        rx = torch.tensor(x[sl].astype(np.float64), requires_grad=True)
        ref = ref_torch.layer_dense(rx, rp, torch.tensor(bm[sl].astype(np.float64)), "elu", "concat")
        ref.backward(torch.tensor(dout[sl].astype(np.float64)))
        assert_close(res["lists"][0][sl], ref.detach().numpy(), what="graph-resident forward vs oracle")
        assert_close(res["lists"][1][sl], rx.grad.numpy(), atol_scale=5e-6, what="graph-resident dX vs oracle")


@pytest.mark.parametrize("scale", [30.0])
def test_graph_resident_rows_beside_a_node_with_towering_logits(scale):
    """Node 0 of every graph gets an f2 some 30 above everyone else's in every head, and only rows 0..3 visit it: their
    softmax collapses onto node 0, while every other row of the graph must come out as if node 0 were not there (a
    softmax that took its maximum over the graph instead of over the row's own list would lose those rows' digits).
    Mask route and prebuilt lists against the float64 oracle, and against each other bit for bit."""
    from gat_amd import ops
    from gat_amd.graph import MaskLists
    dev = torch.device("cuda:0")
    B, N, F, H, D = 256, 24, 5, 8, 8
    rng = np.random.default_rng(77)
    adj = (rng.random((B, N, N)) < 0.2) | np.eye(N, dtype=bool)[None]
    adj[:, :, 0] = False
    adj[:, :4, 0] = True                                     # only rows 0..3 visit node 0
    bm = np.where(adj, 0.0, -1e9).astype(np.float32)
    p = layer_params(F, H, D, seed=9)
    x = rng.normal(0, 1, (B, N, F)).astype(np.float32)
    # node 0's features along the direction that raises f2 in every head (f2 = <W x, a2>: the gradient of sum_h f2_h)
    direction = np.einsum("fhd,hd->f", p["W"].reshape(F, H, D), p["a2"].reshape(H, D))
    wa2 = np.einsum("fhd,hd->fh", p["W"].reshape(F, H, D), p["a2"].reshape(H, D))
    x[:, 0, :] = (scale * direction / np.abs(direction @ wa2).min()).astype(np.float32)
    f2_node0 = x[0, 0].astype(np.float64) @ wa2
    assert f2_node0.min() > 0.9 * scale                      # every head of node 0 towers over the rest
    dout = rng.normal(0, 1, (B, N, H * D)).astype(np.float32)
    rp = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    rx = torch.tensor(x[:4].astype(np.float64), requires_grad=True)
    ref = ref_torch.layer_dense(rx, rp, torch.tensor(bm[:4].astype(np.float64)), "elu", "concat")
    ref.backward(torch.tensor(dout[:4].astype(np.float64)))
    res = []
    bt = torch.tensor(bm, device=dev)
    for mask in (bt, MaskLists(bt)):
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, device=dev, requires_grad=True)
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], mask, ops.make_cfg(H, D, "elu", mask_path="lists"))
        out.backward(torch.tensor(dout, device=dev))
        assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(xt.grad).all())
        assert_close(out.detach().cpu().numpy()[:4], ref.detach().numpy(), what="forward, scale %g" % scale)
        # rows 4.. never see node 0: their outputs must not feel the bound at all (rows of exact zeros would show here)
        assert float(out.detach()[:, 4:].abs().min(dim=-1).values.max()) > 0
        assert_close(xt.grad.cpu().numpy()[:4], rx.grad.numpy(), atol_scale=5e-6, what="dX, scale %g" % scale)
        res.append((out.detach(), xt.grad, tp["W"].grad))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_graph_resident_parameter_gradients_against_the_oracle():
    """Every gradient of the graph-resident kernels — dX, dW, da1, da2, db1, db2, dbias — against the float64 oracle
    on a whole batch (256 graphs of 52 nodes: small enough for float64 on the host), hard and soft masks, without
    going through the matrix-core kernels."""
    from gat_amd import ops
    dev = torch.device("cuda:0")
    B, N, F, H, D = 256, 52, 6, 8, 8
    for soft in (False, True):
        bm = _masks(B, N, 0.12, 77, soft=soft).astype(np.float32)
        p = layer_params(F, H, D, seed=31)
        x = np.random.default_rng(32).normal(0, 1, (B, N, F)).astype(np.float32)
        dout = np.random.default_rng(33).normal(0, 1, (B, N, H * D)).astype(np.float32)
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, device=dev, requires_grad=True)
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], torch.tensor(bm, device=dev),
                                  ops.make_cfg(H, D, "elu", mask_path="lists"))
        out.backward(torch.tensor(dout, device=dev))
        rp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
        rx = torch.tensor(x.astype(np.float64), requires_grad=True)
        ref = ref_torch.layer_dense(rx, rp, torch.tensor(bm.astype(np.float64)), "elu", "concat")
        ref.backward(torch.tensor(dout.astype(np.float64)))
        assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), what="forward")
        assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="dX")
        for k in ("W", "a1", "a2", "bias"):
            assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), atol_scale=5e-6, what="d" + k)
        for k in ("b1", "b2"):          # sums of terms that cancel almost completely: tolerance relative to the sum of magnitudes
            got, want = tp[k].grad.cpu().numpy().astype(np.float64), rp[k].grad.numpy()
            assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-6 * B * N * np.abs(dout).mean() * H), (k, got, want)


@pytest.mark.parametrize("B,N,F,H,D,drops", [(256, 52, 6, 8, 8, (0.0, 0.0)), (260, 264, 8, 8, 8, (0.0, 0.0)), (300, 132, 7, 1, 64, (0.0, 0.0)),
                                              (256, 8, 3, 4, 16, (0.0, 0.0)), (256, 72, 5, 4, 16, (0.6, 0.6)), (256, 52, 6, 8, 8, (0.5, 0.0))])
def test_prebuilt_lists_equal_the_mask_reading_kernels_bit_for_bit(B, N, F, H, D, drops):
    """MaskLists (the lists of an unchanging mask, built once) through the graph-resident kernels: the lists hold the
    entries the mask-reading kernels visit, in the order in which those compact them, so forward, every gradient and
    the dropout masks must be bit-identical to the run that reads the mask (the sums over all nodes — the gradients
    of bias, a1, a2, b1, b2, whose partial sums follow the block shape — to rounding)."""
    from gat_amd import ops
    from gat_amd.graph import MaskLists
    dev = torch.device("cuda:0")
    bm = torch.tensor(_masks(B, N, 0.1, 11 + N).astype(np.float32), device=dev)
    ml = MaskLists(bm)
    assert ml.fits(H, ops.padded_width(D)) and ml.entries >= int((bm == 0).sum()) and ml.max_fwd % 4 == 0
    p = layer_params(F, H, D, seed=N)
    x = np.random.default_rng(N + 1).normal(0, 1, (B, N, F)).astype(np.float32)
    dout = torch.tensor(np.random.default_rng(N + 2).normal(0, 1, (B, N, H * D)).astype(np.float32), device=dev)
    res = []
    for mask in (bm, ml):
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, device=dev, requires_grad=True)
        cfg = ops.make_cfg(H, D, "elu", coef_drop=drops[0], in_drop=drops[1], training=True, seed=4242, mask_path="lists")
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], mask, cfg)
        out.backward(dout)
        res.append([out.detach(), xt.grad] + [tp[k].grad for k in ("W", "a1", "a2", "b1", "b2", "bias")])
    torch.cuda.synchronize()
    for a, b, name in zip(res[0], res[1], ("out", "dX", "dW", "da1", "da2", "db1", "db2", "dbias")):
        if name in ("dbias", "da1", "da2", "db1", "db2") and H >= 2:
            # sums over all nodes of the batch: the half-row blocks of the lists' backward kernel form them themselves
            # (the mask route leaves da, db to the projection's backward pass) and group their partial sums differently
            assert_close(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-5, atol_scale=5e-6, what="prebuilt lists vs mask: " + name)
            continue
        assert torch.equal(a, b), "prebuilt lists vs mask: %s differs (max %g)" % (name, float((a - b).abs().max()))
    assert float(res[1][0].abs().max()) > 0


@pytest.mark.parametrize("B,N,F,H,D,density,drops", [(256, 132, 6, 4, 16, 0.5, (0.0, 0.0)), (256, 100, 5, 2, 32, 0.7, (0.6, 0.5)),
                                                     (257, 320, 4, 8, 8, 0.3, (0.0, 0.0)), (256, 4, 3, 8, 8, 1.0, (0.3, 0.0))])
def test_prebuilt_lists_longer_than_one_fetch(B, N, F, H, D, density, drops):
    """Lists of more than 32 entries (the walk kernels fetch a row's or column's entries 32 at a time), every head
    layout of the half-row backward kernel, the largest and the smallest graph: mask route and prebuilt lists bit for
    bit, and the first graphs against the float64 oracle with the dropout masks replayed on the host."""
    from gat_amd import ops
    from gat_amd.graph import MaskLists
    dev = torch.device("cuda:0")
    bm = torch.tensor(_masks(B, N, density, 5 + N).astype(np.float32), device=dev)
    ml = MaskLists(bm)
    assert ml.fits(H, ops.padded_width(D)) and (N < 64 or ml.max_fwd > 32)
    p = layer_params(F, H, D, seed=N + 1)
    x = np.random.default_rng(N + 2).normal(0, 1, (B, N, F)).astype(np.float32)
    dout = np.random.default_rng(N + 3).normal(0, 1, (B, N, H * D)).astype(np.float32)
    res = []
    for mask in (bm, ml):
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, device=dev, requires_grad=True)
        cfg = ops.make_cfg(H, D, "elu", coef_drop=drops[0], in_drop=drops[1], training=True, seed=77, mask_path="lists")
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], mask, cfg)
        out.backward(torch.tensor(dout, device=dev))
        res.append([out.detach(), xt.grad] + [tp[k].grad for k in ("W", "a1", "a2", "b1", "b2")])
    from gat_amd import _lib
    # the largest graphs leave the mask route no room for its bits and lists beside the rows: its backward pass then
    # runs the row-block kernels, which add in another order
    same_walk = bool(_lib.lib.gatmi_dense_attn_bwd_fused_ok(B, N, H, ops.padded_width(D), 1.0 - drops[0], 1.0 - drops[1]))
    assert same_walk == (N < 320)
    for a, b, name in zip(res[0], res[1], ("out", "dX", "dW", "da1", "da2", "db1", "db2")):
        if name == "out" or (same_walk and name in ("dX", "dW")):
            assert torch.equal(a, b), "prebuilt lists vs mask: %s differs (max %g)" % (name, float((a - b).abs().max()))
        else:
            assert_close(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-5, atol_scale=5e-6, what="prebuilt lists vs mask " + name)
    g = 2
    dp = ops.padded_width(D)
    coef_mask, ft_mask = dense_dropout_masks(77, B, N, H, D, dp, drops[0], drops[1])
    rp = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    rx = torch.tensor(x[:g].astype(np.float64), requires_grad=True)
    ref = ref_torch.layer_dense(rx, rp, bm[:g].cpu().double(), "elu", "concat",
                                coef_mask=None if coef_mask is None else torch.tensor(coef_mask[:g]),
                                ft_mask=None if ft_mask is None else torch.tensor(ft_mask[:g]),
                                coef_keep=1.0 - drops[0], ft_keep=1.0 - drops[1])
    ref.backward(torch.tensor(dout[:g].astype(np.float64)))
    assert_close(res[1][0][:g].cpu().numpy(), ref.detach().numpy(), what="forward vs oracle")
    assert_close(res[1][1][:g].cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="dX vs oracle")


@pytest.mark.parametrize("B,N,F,H,D,drops", [(256, 246, 5, 8, 8, (0.0, 0.0)), (300, 90, 7, 4, 16, (0.6, 0.6)), (256, 90, 4, 2, 32, (0.0, 0.0)),
                                              (256, 7, 3, 8, 8, (0.3, 0.0)), (256, 246, 6, 1, 64, (0.0, 0.0))])
def test_prebuilt_lists_for_graphs_whose_size_is_not_a_multiple_of_four(B, N, F, H, D, drops):
    """The reference's atlases have 246 and 90 nodes (Data_processing.py:46,518).  The mask-reading graph-resident kernels
    need N % 4 == 0 for their 16-byte mask loads; kernels that walk prebuilt lists do not: MaskLists of such a batch
    against the float64 oracle (first graphs) and against the mask given as a tensor (whole batch, other kernels:
    tolerance).  One head of 64 floats has no list-walking backward kernel for these sizes and says so."""
    from gat_amd import ops
    from gat_amd.graph import MaskLists
    dev = torch.device("cuda:0")
    bm = torch.tensor(_masks(B, N, 0.15, 21 + N).astype(np.float32), device=dev)
    ml = MaskLists(bm)
    dp = ops.padded_width(D)
    if H == 1:
        assert not ml.fits(H, dp)
        with pytest.raises(ValueError):
            ops.dense_gat_layer(torch.zeros(B, N, F, device=dev), *[torch.zeros(s, device=dev) for s in ((F, H, D), (H, D), (H, D), (H,), (H,), (H, D))],
                                ml, ops.make_cfg(H, D, "elu"))
        return
    assert ml.fits(H, dp)
    p = layer_params(F, H, D, seed=N + 4)
    x = np.random.default_rng(N + 5).normal(0, 1, (B, N, F)).astype(np.float32)
    dout = np.random.default_rng(N + 6).normal(0, 1, (B, N, H * D)).astype(np.float32)
    res = []
    for mask in (ml, bm):
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, device=dev, requires_grad=True)
        cfg = ops.make_cfg(H, D, "elu", coef_drop=drops[0], in_drop=drops[1], training=True, seed=31, mask_path="lists")
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], mask, cfg)
        out.backward(torch.tensor(dout, device=dev))
        res.append([out.detach(), xt.grad] + [tp[k].grad for k in ("W", "a1", "a2", "bias")])
    for a, b, name in zip(res[0], res[1], ("out", "dX", "dW", "da1", "da2", "dbias")):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol_scale=5e-6, what="prebuilt lists vs mask tensor: " + name)
    g = 2
    coef_mask, ft_mask = dense_dropout_masks(31, B, N, H, D, dp, drops[0], drops[1])
    rp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    rx = torch.tensor(x[:g].astype(np.float64), requires_grad=True)
    ref = ref_torch.layer_dense(rx, rp, bm[:g].cpu().double(), "elu", "concat",
                                coef_mask=None if coef_mask is None else torch.tensor(coef_mask[:g]),
                                ft_mask=None if ft_mask is None else torch.tensor(ft_mask[:g]),
                                coef_keep=1.0 - drops[0], ft_keep=1.0 - drops[1])
    ref.backward(torch.tensor(dout[:g].astype(np.float64)))
    assert_close(res[0][0][:g].cpu().numpy(), ref.detach().numpy(), what="forward vs oracle")
    assert_close(res[0][1][:g].cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="dX vs oracle")


def test_prebuilt_lists_refuse_masks_whose_biases_matter():
    from gat_amd.graph import MaskLists
    dev = torch.device("cuda:0")
    hard = _masks(256, 16, 0.2, 3).astype(np.float32)
    MaskLists(torch.tensor(hard, device=dev))
    soft = _masks(256, 16, 0.2, 3, soft=True).astype(np.float32)
    with pytest.raises(ValueError):
        MaskLists(torch.tensor(soft, device=dev))
    everywhere = hard.copy()
    everywhere[200, 5, :] = -1e9                               # a row masked everywhere: fp32 swallows its logits
    with pytest.raises(ValueError):
        MaskLists(torch.tensor(everywhere, device=dev))
    assert MaskLists(torch.tensor(hard[:, :15, :15].copy(), device=dev)).N == 15      # any N from 4 to 320 ...
    with pytest.raises(ValueError):
        MaskLists(torch.tensor(hard[:, :3, :3].copy(), device=dev))      # ... but not below


def test_mask_memo_switches_a_batch_of_small_graphs_to_prebuilt_lists():
    """GATHeads on one unchanged [B,N,N] mask tensor: the first call reads the mask, the second builds the lists,
    every later call walks them; all calls give the same bits, and a changed mask drops the memo entry."""
    from gat_amd import layers
    from gat_amd.graph import MaskLists
    dev = torch.device("cuda:0")
    B, N, F, H, D = 256, 40, 7, 8, 8
    bt = torch.tensor(_masks(B, N, 0.1, 5).astype(np.float32), device=dev)
    m = layers.GATHeads(F, H, D, device=dev)
    m.load(layer_params(F, H, D, seed=9))
    x = torch.randn(B, N, F, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    layers.mask_graphs._entries.clear()
    calls = [m(x, bt, "elu").detach().clone() for _ in range(4)]
    assert isinstance(layers.mask_graphs.lookup(bt, (H, D)), MaskLists)
    for c in calls[1:]:
        assert torch.equal(c, calls[0])
    narrow = layers.GATHeads(F, 1, 7, device=dev)                # rows of 8 floats: not a shape of the graph-resident kernels
    narrow(x, bt, "identity")
    assert isinstance(layers.mask_graphs.lookup(bt, (1, 8)), layers.CSRGraph)
    bt[0, 0, 1] = 0.0 if float(bt[0, 0, 1]) != 0.0 else -1e9    # in-place change: the version counter moves
    changed = m(x, bt, "elu")
    assert not torch.equal(changed, calls[0])
    assert torch.equal(changed, m(x, bt, "elu")) and torch.equal(changed, m(x, bt, "elu"))
    # graphs of 90 nodes (the reference's AAL atlas, not a multiple of 4): the lists are built on the FIRST sighting
    odd = torch.tensor(_masks(B, 90, 0.1, 6).astype(np.float32), device=dev)
    xo = torch.randn(B, 90, F, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    layers.mask_graphs._entries.clear()
    first = m(xo, odd, "elu").detach().clone()
    assert isinstance(layers.mask_graphs._entries[id(odd)][2], MaskLists)
    assert torch.equal(first, m(xo, odd, "elu"))
    from gat_amd import ops
    tp = {k: getattr(m, k).detach() for k in ("W", "a1", "a2", "b1", "b2", "bias")}
    layers.mask_graphs.enabled = False
    try:
        plain = ops.dense_gat_layer(xo, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], odd, ops.make_cfg(H, D, "elu"))
    finally:
        layers.mask_graphs.enabled = True
    assert_close(first.cpu().numpy(), plain.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="lists at first sighting vs the mask tensor")


@pytest.mark.parametrize("B,N,F,H,D", [(256, 52, 6, 8, 8), (270, 100, 9, 4, 16)])
def test_graph_resident_kernels_with_a_residual_projection(B, N, F, H, D):
    """The residual branch (utils/layers.py:28-32, widths differ: a second projection with bias) through the
    graph-resident list kernels: the activation's derivative then comes from the saved output and g is written
    for dW_res.  Every gradient against the matrix-core kernels, two graphs against the float64 oracle."""
    from gat_amd import ops
    dev = torch.device("cuda:0")
    bm = _masks(B, N, 0.1, 90 + N).astype(np.float32)
    p = layer_params(F, H, D, seed=N + 3, res=True)
    x = np.random.default_rng(N + 4).normal(0, 1, (B, N, F)).astype(np.float32)
    dout = np.random.default_rng(N + 5).normal(0, 1, (B, N, H * D)).astype(np.float32)
    names = ("W", "a1", "a2", "b1", "b2", "bias", "W_res", "b_res")
    res = {}
    for path in ("lists", "dense"):
        tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
        xt = torch.tensor(x, device=dev, requires_grad=True)
        out = ops.dense_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], torch.tensor(bm, device=dev),
                                  ops.make_cfg(H, D, "elu", mask_path=path), tp["W_res"], tp["b_res"])
        out.backward(torch.tensor(dout, device=dev))
        res[path] = [out.detach().cpu().numpy(), xt.grad.cpu().numpy()] + [tp[k].grad.cpu().numpy() for k in names]
    for a, b, name in zip(res["lists"], res["dense"], ("out", "dX") + tuple("d" + k for k in names)):
        assert_close(a, b, rtol=1e-5, atol_scale=5e-6, what="graph-resident vs matrix cores, residual: " + name)
    sl = slice(B - 2, B)
    rp = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    rx = torch.tensor(x[sl].astype(np.float64), requires_grad=True)
    ref = ref_torch.layer_dense(rx, rp, torch.tensor(bm[sl].astype(np.float64)), "elu", "concat", residual=True)
    ref.backward(torch.tensor(dout[sl].astype(np.float64)))
    assert_close(res["lists"][0][sl], ref.detach().numpy(), what="graph-resident forward vs oracle, residual")
    assert_close(res["lists"][1][sl], rx.grad.numpy(), atol_scale=5e-6, what="graph-resident dX vs oracle, residual")
