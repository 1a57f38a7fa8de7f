"""Size-independent properties checked at the benchmark's full size (R-MAT graph at the
ogbn-products shape: 2.45 M nodes, 64.3 M stored entries, hub rows of tens of thousands of
entries that go through the split-and-merge path), where the oracle would take far too long:

  * constant features  => every coefficient row sums to 1, so out = act(Wh + bias) at every node
  * sum_j dWh_j (attention part) == sum_i g_i                 (coefficient rows sum to 1)
  * sum_j df2_j == sum_i df1_i                                (both are the sum of all dlogit_ij;
                                                                df1 comes from the forward pass's
                                                                positive-part sums, df2 from the CSC sweep)
  * the layer is deterministic: two runs agree bit for bit

and, value for value against the oracle on sampled nodes: an output row depends only on the row's
own neighbour list, and dX_j only on the rows that read j, so the oracle evaluated on the sub-graph
induced by a sample of rows (with their complete neighbour lists) reproduces exactly those rows of
the full-size result — hub rows and a column long enough to be split included.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, D, F = 8, 8, 100


@pytest.fixture(scope="module")
def big():
    from gat_amd import CSRGraph, synth
    dev = torch.device("cuda:0")
    n, e = synth.PRODUCTS_NODES, synth.PRODUCTS_EDGES
    dst, src = synth.rmat_graph(n, e, 4, dev)
    g = CSRGraph.from_device_coo(dst, src, n, n)
    del dst, src
    assert g.nnz == e + n and len(g.split_csr) > 0 and len(g.split_csc) > 0
    return g, dev


def test_constant_features_give_constant_output(big):
    from gat_amd import ops, synth
    g, dev = big
    n = g.n_dst
    p = synth.glorot_layer(F, H, D, 7, dev, zero_bias=False)
    x = torch.ones(n, F, device=dev)
    out = ops.sparse_gat_layer(x, p["W"], p["a1"], p["a2"], p["b1"], p["b2"], p["bias"], g, ops.make_cfg(H, D, "elu"))
    want = torch.nn.functional.elu(p["W"].double().sum(0).reshape(-1) + p["bias"].double().reshape(-1))
    err = (out.double() - want[None]).abs().max().item()
    assert err <= 1e-5 * want.abs().max().item(), err


def test_gradient_checksums_and_determinism(big):
    from gat_amd import _lib, ops, synth
    from gat_amd._lib import ptr
    g, dev = big
    n = g.n_dst
    hd = H * D
    p = synth.glorot_layer(F, H, D, 7, dev, zero_bias=False)
    x = synth.normal((n, F), 5, dev)
    dout = synth.normal((n, hd), 8, dev)
    cfg = ops.make_cfg(H, D, "elu")
    st = _lib.current_stream()
    new = lambda *s: torch.empty(*s, device=dev)
    Wp, a1p, a2p, _, Wh, f1, f2, _ = ops._project(x, p["W"], p["a1"], p["a2"], p["b1"], p["b2"], cfg)
    outs = []
    for _ in range(2):
        out, agg, pos, lse, q = new(n, hd), new(n, hd), new(n, hd), new(n, H), new(n, H)
        ws = new(_lib.lib.gatmi_gat_fwd_ws_floats(g.n_slots_csr, H, D))
        _lib.call("gatmi_gat_fwd", n, H, D, ptr(g.items_csr), len(g.items_csr), ptr(g.split_csr), len(g.split_csr),
                  ptr(g.col), ptr(Wh), _lib.DTYPE_F32, hd, ptr(f1), ptr(f2), None, None, ptr(p["bias"].reshape(-1).contiguous()), None,
                  _lib.ACT_ELU, _lib.REDUCE_CONCAT, 0.2, 1.0, 1.0, 0, ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), ptr(ws), st)
        gg, stats, df1, dbias = new(n, hd), new(n, 3 * H), new(n, H), new(hd)
        wsn = new(_lib.lib.gatmi_gat_bwd_node_ws_floats(H, D))
        _lib.call("gatmi_gat_bwd_node", n, H, D, ptr(dout), ptr(out), ptr(agg), ptr(pos), ptr(q), ptr(f1), ptr(lse), None,
                  _lib.ACT_ELU, _lib.REDUCE_CONCAT, 0.2, ptr(gg), hd, None, 0, ptr(stats), 3 * H, ptr(df1), ptr(dbias), ptr(wsn), st)
        # the same from the pre-activation agg + bias instead of the saved output (the form the layers use)
        gg2, stats2, df12, dbias2 = new(n, hd), new(n, 3 * H), new(n, H), new(hd)
        _lib.call("gatmi_gat_bwd_node", n, H, D, ptr(dout), None, ptr(agg), ptr(pos), ptr(q), ptr(f1), ptr(lse),
                  ptr(p["bias"].reshape(-1).contiguous()), _lib.ACT_ELU, _lib.REDUCE_CONCAT, 0.2, ptr(gg2), hd, None, 0, ptr(stats2),
                  3 * H, ptr(df12), ptr(dbias2), ptr(wsn), st)
        # (the saved-output form computes ELU' as out + 1, which loses relative accuracy where out is close to -1)
        assert torch.allclose(gg2, gg, rtol=1e-5, atol=2e-7 * float(gg.abs().max()))
        assert torch.allclose(df12, df1, rtol=1e-3, atol=1e-5 * float(df1.abs().max()))
        dfeat, df2 = new(n, hd), new(n, H)
        wsb = new(_lib.lib.gatmi_gat_bwd_ws_floats(g.n_slots_csc, H, D))
        _lib.call("gatmi_gat_bwd_csc", n, H, D, ptr(g.items_csc), len(g.items_csc), ptr(g.split_csc), len(g.split_csc),
                  ptr(g.rowidx), None, ptr(gg), _lib.DTYPE_F32, hd, ptr(stats), 3 * H, ptr(Wh), ptr(f2), None, None, None, None, None,
                  0.2, 1.0, 1.0, 0, ptr(dfeat), ptr(df2), ptr(wsb), st)
        outs.append((out, dfeat, df2, df1, dbias, gg))
    (out, dfeat, df2, df1, dbias, gg), second = outs
    for a, b in zip(outs[0], second):
        assert torch.equal(a, b), "two runs differ"
    assert torch.isfinite(out).all() and torch.isfinite(dfeat).all()
    # coefficient rows sum to one: column sums of the transposed aggregation equal those of g
    lhs, rhs = dfeat.double().sum(0), gg.double().sum(0)
    scale = gg.double().abs().sum(0)
    assert ((lhs - rhs).abs() <= 1e-6 * scale).all(), ((lhs - rhs).abs() / scale).max().item()
    assert ((dbias.double() - rhs).abs() <= 1e-6 * scale).all()
    # both logit-gradient vectors add up to the same total
    l2, r2 = df2.double().sum(0), df1.double().sum(0)
    s2 = df2.double().abs().sum(0) + df1.double().abs().sum(0)
    assert ((l2 - r2).abs() <= 1e-6 * s2).all(), ((l2 - r2).abs() / s2).max().item()


def _induced(g, rows):
    """Sub-graph that keeps the complete neighbour lists of `rows` and nothing else: (nodes, rowptr,
    col) with `nodes` the sorted global ids of the rows and of every source they read, ids relabelled
    to positions in `nodes`, rows outside `rows` empty."""
    rows = np.unique(np.asarray(rows, dtype=np.int64))
    lo, hi = g.rowptr_host[rows], g.rowptr_host[rows + 1]
    lens = hi - lo
    pos = np.repeat(lo - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(int(lens.sum()))
    cols = g.col[torch.as_tensor(pos, device=g.col.device)].cpu().numpy().astype(np.int64)
    nodes = np.unique(np.concatenate([rows, cols]))
    sub_len = np.zeros(len(nodes), dtype=np.int64)
    sub_len[np.searchsorted(nodes, rows)] = lens
    rowptr = np.concatenate([[0], np.cumsum(sub_len)])
    return nodes, rowptr, np.searchsorted(nodes, cols).astype(np.int32)     # rows ascending: entries already in row order


def _layer_and_input(dev, n):
    from gat_amd import synth
    return synth.glorot_layer(F, H, D, 7, dev, zero_bias=False), synth.normal((n, F), 5, dev)


def test_sampled_rows_match_the_oracle_at_full_size(big):
    """Forward values of 300 random rows and the three longest rows (tens of thousands of entries: the
    split-and-merge path) against oracle.ref_numpy.gat_layer_csr on their induced sub-graph."""
    from gat_amd import ops
    from oracle import ref_numpy
    from tests.util import assert_close
    g, dev = big
    n = g.n_dst
    p, x = _layer_and_input(dev, n)
    out = ops.sparse_gat_layer(x, p["W"], p["a1"], p["a2"], p["b1"], p["b2"], p["bias"], g, ops.make_cfg(H, D, "elu"))
    deg = np.diff(g.rowptr_host)
    hubs = np.argsort(deg)[-3:]
    assert deg[hubs].min() > g.chunk, "the longest rows are expected to be split rows"
    rows = np.unique(np.concatenate([hubs, np.random.default_rng(0).choice(n, 300, replace=False)]))
    nodes, rowptr, col = _induced(g, rows)
    layer = {k: v.detach().cpu().double().numpy() for k, v in p.items()}
    ref = ref_numpy.gat_layer_csr(x[torch.as_tensor(nodes, device=dev)].cpu().double().numpy(), layer, rowptr, col)
    got = out[torch.as_tensor(rows, device=dev)].cpu().numpy()
    assert_close(got, ref[np.searchsorted(nodes, rows)], what="sampled rows of the full-size forward pass")


def _readers(g, js):
    """Every row that reads one of the sources js (their CSC columns), and js themselves."""
    js = np.asarray(js, dtype=np.int64)
    lo, lens = g.colptr_host[js], np.diff(g.colptr_host)[js]
    pos = np.repeat(lo - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(int(lens.sum()))
    return np.unique(np.concatenate([js, g.rowidx[torch.as_tensor(pos, device=g.rowidx.device)].cpu().numpy().astype(np.int64)]))


def _within_budget(g, candidates, budget, at_least):
    """The longest prefix of `candidates` whose readers hold at most `budget` entries between them."""
    rdeg = np.diff(g.rowptr_host)
    kept, total = [], 0
    for j in candidates:
        cost = int(rdeg[_readers(g, [j])].sum())
        if kept and total + cost > budget:
            break
        kept.append(int(j))
        total += cost
    assert len(kept) >= at_least, "only %d of the sampled nodes fit the oracle's budget of %d entries" % (len(kept), budget)
    return kept


def _check_sampled_dx(g, dev, targets, what):
    from gat_amd import ops, synth
    from oracle import ref_torch
    from tests.util import assert_close
    n = g.n_dst
    p, x = _layer_and_input(dev, n)
    x.requires_grad_(True)
    dout = synth.normal((n, H * D), 8, dev)
    out = ops.sparse_gat_layer(x, p["W"], p["a1"], p["a2"], p["b1"], p["b2"], p["bias"], g, ops.make_cfg(H, D, "elu"))
    out.backward(dout)
    targets = np.unique(np.asarray(targets, dtype=np.int64))
    rows = _readers(g, targets)
    nodes, rowptr, col = _induced(g, rows)
    assert len(col) < 3_000_000, "sub-graph of %d entries is too large for the float64 oracle" % len(col)
    pos_rows, pos_t = np.searchsorted(nodes, rows), np.searchsorted(nodes, targets)
    xs = x.detach()[torch.as_tensor(nodes, device=dev)].cpu().double().requires_grad_(True)
    ds = torch.zeros(len(nodes), H * D, dtype=torch.float64)
    ds[pos_rows] = dout[torch.as_tensor(rows, device=dev)].cpu().double()
    pp = {k: v.detach().cpu().double() for k, v in p.items()}
    ref_torch.layer_csr(xs, pp, torch.as_tensor(rowptr), torch.as_tensor(col), "elu", "concat").backward(ds)
    got = x.grad[torch.as_tensor(targets, device=dev)].cpu().numpy()
    assert_close(got, xs.grad[pos_t].numpy(), atol_scale=5e-6, what=what)


def test_sampled_input_gradients_match_the_oracle_at_full_size(big):
    """dX of random nodes (as many as keep the sub-graph under 2 M entries) against the float64 autograd
    oracle on the sub-graph induced by every row that reads them (each with its complete neighbour list)."""
    g, dev = big
    cdeg = np.diff(g.colptr_host)
    targets = np.random.default_rng(1).choice(np.nonzero(cdeg <= 400)[0], 150, replace=False)
    _check_sampled_dx(g, dev, _within_budget(g, targets, 2_000_000, 20), "sampled rows of the full-size input gradient")


def test_sampled_input_gradient_of_a_split_column_at_full_size():
    """The same for a node whose column is cut into pieces and merged.  With the benchmark's work-item
    size the shortest split column has 4 700 readers holding 7.7 M entries between them, too many for
    the float64 oracle; the same graph cut into work items of 128 entries splits columns of a few
    hundred entries, whose readers the oracle handles."""
    from gat_amd import CSRGraph, synth
    dev = torch.device("cuda:0")
    n, e = synth.PRODUCTS_NODES, synth.PRODUCTS_EDGES
    dst, src = synth.rmat_graph(n, e, 4, dev)
    g = CSRGraph.from_device_coo(dst, src, n, n, chunk=128)
    del dst, src
    split = g.split_csc.cpu().numpy()
    assert len(split) > 0
    cdeg = np.diff(g.colptr_host)
    cols = np.unique(split[:, 0])                       # first field of a split record: the column it belongs to
    cols = _within_budget(g, cols[np.argsort(cdeg[cols])][:8], 2_000_000, 1)
    assert cdeg[cols].min() > g.chunk
    _check_sampled_dx(g, dev, cols, "input gradient of split columns at full size")


def test_sampled_rows_and_gradients_at_one_gpus_share_of_the_two_billion_edge_graph():
    """BASELINE.json configs[4] gives each of 8 GPUs 12.5 M rows / 262.5 M entries (feature-table offsets
    pass 2^31 elements, gathered tables are 3 to 5 GB): the same sampled comparisons with the oracle on that graph, built and run on
    one GPU."""
    from gat_amd import CSRGraph, ops, synth
    from oracle import ref_numpy
    from tests.util import assert_close
    dev = torch.device("cuda:0")
    n, e = 12_500_000, 250_000_000
    if torch.cuda.mem_get_info()[0] < 120 * 2 ** 30:
        pytest.skip("needs about 100 GiB of free device memory")
    dst, src = synth.rmat_graph(n, e, 4, dev)
    g = CSRGraph.from_device_coo(dst, src, n, n)
    del dst, src
    assert g.n_dst * H * D * 3 >= 2 ** 31
    p, x = _layer_and_input(dev, n)
    with torch.no_grad():
        out = ops.sparse_gat_layer(x, p["W"], p["a1"], p["a2"], p["b1"], p["b2"], p["bias"], g, ops.make_cfg(H, D, "elu"))
    deg = np.diff(g.rowptr_host)
    last = np.nonzero(deg > 1)[0][-50:]                    # rows at the far end of the entry array
    rows = np.unique(np.concatenate([np.argsort(deg)[-2:], last, np.random.default_rng(2).choice(n, 200, replace=False)]))
    nodes, rowptr, col = _induced(g, rows)
    layer = {k: v.detach().cpu().double().numpy() for k, v in p.items()}
    ref = ref_numpy.gat_layer_csr(x[torch.as_tensor(nodes, device=dev)].cpu().double().numpy(), layer, rowptr, col)
    assert_close(out[torch.as_tensor(rows, device=dev)].cpu().numpy(), ref[np.searchsorted(nodes, rows)],
                 what="sampled rows of the forward pass, 262.5 M entries")
    del out
    cdeg = np.diff(g.colptr_host)
    cand = np.random.default_rng(3).choice(np.nonzero(cdeg <= 400)[0], 100, replace=False)
    cand = np.concatenate([np.nonzero((cdeg > 0) & (cdeg <= 50))[0][-5:], cand])        # columns at the far end too
    _check_sampled_dx(g, dev, _within_budget(g, cand, 2_000_000, 10), "sampled rows of the input gradient, 262.5 M entries")


def test_dense_batch_routes_agree_at_full_size():
    """BASELINE.json configs[2] at its full size (4096 graphs x 264 nodes, density 0.10): the
    matrix-core kernels, the list kernels and the block-diagonal CSR route compute the same layer;
    outputs and gradients are compared with each other (fp32 vs fp32: a few ulp of the row sums)."""
    from gat_amd import CSRGraph, ops, synth
    dev = torch.device("cuda:0")
    B, N, Fd = 4096, 264, 64
    gen = torch.Generator(device=dev).manual_seed(2)
    adj = torch.rand(B, N, N, device=dev, generator=gen) < 0.05
    adj = adj | adj.transpose(1, 2) | torch.eye(N, dtype=torch.bool, device=dev)[None]
    bm = torch.where(adj, 0.0, -1e9).to(torch.float32)
    del adj
    x = synth.normal((B, N, Fd), 3, dev)
    p = synth.glorot_layer(Fd, H, D, 7, dev, zero_bias=False)
    dout = synth.normal((B, N, H * D), 8, dev)
    results = {}
    for route in ("dense", "lists", "csr"):
        leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        if route == "csr":
            graph = CSRGraph.from_bias_mat(bm)
            out = ops.sparse_gat_layer(x.reshape(B * N, Fd), leaves["W"], leaves["a1"], leaves["a2"], leaves["b1"], leaves["b2"],
                                       leaves["bias"], graph, ops.make_cfg(H, D, "elu")).reshape(B, N, H * D)
            del graph
        else:
            out = ops.dense_gat_layer(x, leaves["W"], leaves["a1"], leaves["a2"], leaves["b1"], leaves["b2"], leaves["bias"], bm,
                                      ops.make_cfg(H, D, "elu", mask_path=route))
        out.backward(dout)
        results[route] = (out.detach(), {k: v.grad for k, v in leaves.items()})
    ref_out, ref_g = results["dense"]
    scale = float(ref_out.abs().max())
    for route in ("lists", "csr"):
        o, g = results[route]
        assert float((o - ref_out).abs().max()) <= 5e-6 * scale, route
        for k in ("W", "a1", "a2", "bias"):
            gs = float(ref_g[k].abs().max())
            assert float((g[k] - ref_g[k]).abs().max()) <= 2e-5 * gs, (route, k)
