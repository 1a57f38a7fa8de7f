"""GPU parity of the fused CSR graph-attention layer (through the C ABI) against the oracle.

Forward outputs are compared with the float64 numpy restatement of utils/layers.py:6-34 restricted
to stored entries (oracle/ref_numpy.py); gradients with the float64 torch restatement
(oracle/ref_torch.py).  Tolerance: 1e-5 relative plus 2e-6 of the tensor's scale (fp32 kernels,
different summation order).  The oracle is "parity unpinned" against TensorFlow (not installed).
"""
import numpy as np
import pytest
import torch

from oracle import ref_numpy, ref_torch
from tests.util import assert_close, layer_params, random_graph

pytestmark = pytest.mark.gpu


def _run_layer(n, F, H, D, dst, src, activation, reduce, seed, res=False, chunk=256, need_dx=True, storage="f32"):
    from gat_amd import CSRGraph, ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    p = layer_params(F, H, D, seed + 1, res=res)
    x = rng.normal(0, 1, (n, F))
    graph = CSRGraph.from_coo(dst, src, n, dev, chunk=chunk)
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=need_dx)
    cfg = ops.make_cfg(H, D, activation=activation, reduce=reduce, storage=storage)
    out = ops.sparse_gat_layer(xt, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph, cfg,
                               tp.get("W_res"), tp.get("b_res"))
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    return x, p, tp, xt, out, rowptr, col


def _check_param_grads(tp, rp, keys, n):
    """Parameter gradients.  db1 / db2 are sums of per-node logit gradients that cancel almost
    completely, so their tolerance is relative to the sum of |terms| (the oracle is run with the
    scalar bias broadcast to one leaf per node to expose the terms), not to the tiny result."""
    for k in keys:
        ref = rp[k].grad.numpy()
        if k in ("b1", "b2"):
            terms = ref                       # [n, H] per-node gradients
            ref = terms.sum(0)
            got = tp[k].grad.cpu().numpy().astype(np.float64)
            tol = 1e-5 * np.abs(ref) + 1e-6 * np.abs(terms).sum(0)
            assert (np.abs(got - ref) <= tol).all(), "d%s: got %s ref %s tol %s" % (k, got, ref, tol)
        else:
            assert_close(tp[k].grad.cpu().numpy(), ref, atol_scale=5e-6, what="d" + k)


def _oracle_leaves(p, n):
    rp = {}
    for k, v in p.items():
        if k in ("b1", "b2"):
            v = np.broadcast_to(v, (n, v.shape[0])).copy()
        rp[k] = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    return rp


CASES = [
    # n, F, H, D, avg_deg, activation, reduce
    (257, 20, 8, 8, 6.0, "elu", "concat"),
    (300, 33, 1, 7, 4.0, "identity", "mean"),
    (190, 16, 4, 8, 9.0, "identity", "concat"),
    (128, 64, 1, 64, 5.0, "elu", "concat"),
    (200, 12, 2, 16, 3.0, "elu", "concat"),
    (150, 10, 8, 4, 5.0, "identity", "mean"),
    (210, 14, 8, 8, 6.0, "relu", "concat"),
]


@pytest.mark.parametrize("n,F,H,D,deg,activation,reduce", CASES)
def test_forward_matches_float64_oracle(n, F, H, D, deg, activation, reduce):
    dst, src = random_graph(n, deg, seed=n)
    x, p, tp, xt, out, rowptr, col = _run_layer(n, F, H, D, dst, src, activation, reduce, seed=n + 7)
    act = {"elu": ref_numpy.elu, "relu": ref_numpy.relu, "identity": ref_numpy.identity}[activation]
    ref = ref_numpy.gat_layer_csr(x, p, rowptr, col, activation=act, reduce=reduce)
    assert_close(out.detach().cpu().numpy(), ref, what="forward")


@pytest.mark.parametrize("n,F,H,D,deg,activation,reduce", CASES)
def test_gradients_match_float64_oracle(n, F, H, D, deg, activation, reduce):
    dst, src = random_graph(n, deg, seed=n)
    x, p, tp, xt, out, rowptr, col = _run_layer(n, F, H, D, dst, src, activation, reduce, seed=n + 7)
    rng = np.random.default_rng(99)
    dout = rng.normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=out.device))

    rp = _oracle_leaves(p, n)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rout = ref_torch.layer_csr(rx, rp, torch.tensor(rowptr), torch.tensor(col), activation, reduce)
    rout.backward(torch.tensor(dout, dtype=torch.float64))
    assert_close(out.detach().cpu().numpy(), rout.detach().numpy(), what="forward(torch oracle)")
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="dX")
    _check_param_grads(tp, rp, ("W", "a1", "a2", "b1", "b2", "bias"), n)


def test_hub_rows_are_split_and_merged():
    """Rows and columns far longer than the chunk size go through the partial-result merge."""
    n, F, H, D = 600, 24, 8, 8
    dst, src = random_graph(n, 3.0, seed=5, hubs=3, hub_deg=500)
    x, p, tp, xt, out, rowptr, col = _run_layer(n, F, H, D, dst, src, "elu", "concat", seed=11, chunk=32)
    assert np.diff(rowptr).max() > 32 * 4
    ref = ref_numpy.gat_layer_csr(x, p, rowptr, col, activation=ref_numpy.elu, reduce="concat")
    assert_close(out.detach().cpu().numpy(), ref, what="forward with split rows")
    rng = np.random.default_rng(3)
    dout = rng.normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=out.device))
    rp = _oracle_leaves(p, n)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rout = ref_torch.layer_csr(rx, rp, torch.tensor(rowptr), torch.tensor(col), "elu", "concat")
    rout.backward(torch.tensor(dout, dtype=torch.float64))
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="dX")
    _check_param_grads(tp, rp, ("W", "a1", "a2", "b1", "b2", "bias"), n)


@pytest.mark.parametrize("H,D,activation,reduce", [(4, 8, "elu", "concat"), (4, 8, "identity", "mean"), (2, 7, "identity", "mean")])
def test_residual_projection(H, D, activation, reduce):
    """The residual projection is per head (utils/layers.py:30): with reduce="mean" it is averaged
    over the heads together with the aggregate and the bias."""
    n, F = 220, 18
    dst, src = random_graph(n, 5.0, seed=21)
    x, p, tp, xt, out, rowptr, col = _run_layer(n, F, H, D, dst, src, activation, reduce, seed=31, res=True)
    rp = _oracle_leaves(p, n)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rout = ref_torch.layer_csr(rx, rp, torch.tensor(rowptr), torch.tensor(col), activation, reduce, residual=True)
    assert_close(out.detach().cpu().numpy(), rout.detach().numpy(), what="forward residual")
    dout = np.random.default_rng(8).normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=out.device))
    rout.backward(torch.tensor(dout, dtype=torch.float64))
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), atol_scale=5e-6, what="dX")
    _check_param_grads(tp, rp, ("W", "a1", "a2", "b1", "b2", "bias", "W_res", "b_res"), n)


def test_residual_projection_with_bf16_gather_tables():
    """Residual branch + storage="bf16": the backward pass keeps a contiguous fp32 g for the residual's weight
    gradient next to the bf16 records the per-source kernel gathers.  Tolerance: the bf16 class (2e-2)."""
    n, F, H, D = 260, 20, 4, 8
    dst, src = random_graph(n, 6.0, seed=23, hubs=1, hub_deg=200)
    x, p, tp, xt, out, rowptr, col = _run_layer(n, F, H, D, dst, src, "elu", "concat", seed=33, res=True, chunk=64, storage="bf16")
    rp = _oracle_leaves(p, n)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rout = ref_torch.layer_csr(rx, rp, torch.tensor(rowptr), torch.tensor(col), "elu", "concat", residual=True)
    tol = dict(rtol=2e-2, atol_scale=2e-2)
    assert_close(out.detach().cpu().numpy(), rout.detach().numpy(), what="forward residual bf16", **tol)
    dout = np.random.default_rng(9).normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=out.device))
    rout.backward(torch.tensor(dout, dtype=torch.float64))
    assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), what="dX", **tol)
    for k in ("W", "a1", "a2", "bias", "W_res", "b_res"):
        assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), what="d" + k, **tol)


def test_known_answers():
    """Analytic cases: zero attention vectors => uniform coefficients => neighbour mean;
    a lone node with a self-loop => its own projected features."""
    from gat_amd import CSRGraph, ops
    dev = torch.device("cuda:0")
    n, F, H, D = 40, 6, 2, 4
    dst, src = random_graph(n, 3.0, seed=2)
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, (n, F)).astype(np.float32)
    W = rng.normal(0, 1, (F, H, D)).astype(np.float32)
    z = lambda *s: torch.zeros(*s, device=dev)
    graph = CSRGraph.from_coo(dst, src, n, dev)
    cfg = ops.make_cfg(H, D, activation="identity")
    out = ops.sparse_gat_layer(torch.tensor(x, device=dev), torch.tensor(W, device=dev), z(H, D), z(H, D), z(H), z(H),
                               z(H, D), graph, cfg).cpu().numpy()
    wh = x.astype(np.float64) @ W.reshape(F, H * D).astype(np.float64)
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    ref = np.stack([wh[col[rowptr[i]:rowptr[i + 1]]].mean(0) for i in range(n)])
    assert_close(out, ref, what="uniform attention")
    g1 = CSRGraph.from_coo(np.array([0]), np.array([0]), 1, dev)
    o1 = ops.sparse_gat_layer(torch.tensor(x[:1], device=dev), torch.tensor(W, device=dev),
                              torch.ones(H, D, device=dev), torch.ones(H, D, device=dev), z(H), z(H), z(H, D), g1, cfg)
    assert_close(o1.cpu().numpy(), wh[:1], what="single node")


@pytest.mark.parametrize("H,D,activation,reduce,sparse_x", [(8, 8, "elu", "concat", False), (1, 7, "identity", "mean", False),
                                                            (8, 16, "elu", "concat", False),      # rows wider than one projection tile
                                                            (4, 8, "elu", "concat", True)])       # records converted after a CSR projection
def test_bf16_gather_tables_stay_within_the_bf16_class(H, D, activation, reduce, sparse_x):
    """storage="bf16": the gathered Wh rows (forward) and g rows (backward) are kept in bf16 records that also
    carry the exact fp32 logit halves / row statistics; arithmetic stays fp32.  Tolerance: 2e-2 of the tensor's
    scale against the float64 oracle."""
    from gat_amd import CSRGraph, ops
    from gat_amd.features import SparseFeatures
    dev = torch.device("cuda:0")
    n, F = 400, 24
    dst, src = random_graph(n, 8.0, seed=3, hubs=2, hub_deg=300)
    p = layer_params(F, H, D, seed=4)
    x = np.random.default_rng(5).normal(0, 1, (n, F))
    if sparse_x:
        x = x * (np.random.default_rng(7).random((n, F)) < 0.3)
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    xin = SparseFeatures.from_dense(xt, dev) if sparse_x else xt
    graph = CSRGraph.from_coo(dst, src, n, dev, chunk=64)
    out = ops.sparse_gat_layer(xin, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph,
                               ops.make_cfg(H, D, activation, reduce, storage="bf16"))
    exact = ops.sparse_gat_layer(xin, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], tp["bias"], graph,
                                 ops.make_cfg(H, D, activation, reduce))
    assert not torch.equal(out, exact)
    dout = np.random.default_rng(6).normal(0, 1, tuple(out.shape))
    out.backward(torch.tensor(dout, dtype=torch.float32, device=dev))
    rowptr, col = ref_numpy.coo_to_csr(np.stack([dst, src], 1), n)
    rp = _oracle_leaves(p, n)
    rx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = ref_torch.layer_csr(rx, rp, torch.tensor(rowptr), torch.tensor(col), activation, reduce)
    ref.backward(torch.tensor(dout))
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-2, atol_scale=2e-2, what="bf16 forward")
    if not sparse_x:                # sparse node features are a first layer's constant input: no dX
        assert_close(xt.grad.cpu().numpy(), rx.grad.numpy(), rtol=2e-2, atol_scale=2e-2, what="bf16 dX")
    for k in ("W", "a1", "a2", "bias"):
        assert_close(tp[k].grad.cpu().numpy(), rp[k].grad.numpy(), rtol=2e-2, atol_scale=2e-2, what="bf16 d" + k)


@pytest.mark.parametrize("H,D", [(8, 8), (4, 4), (2, 16), (1, 64)])
def test_source_logit_half_formed_from_the_rows_matches_the_table(H, D):
    """gatmi_gat_fwd / gatmi_gat_bwd_csc take the source half of the logits (utils/layers.py:15) either as
    the table f2 the projection wrote, or as (a2, b2) and form <Wh_j, a2> + b2 from the rows they gather.
    Both forms are the same fp32 sum in a different order: results agree to rounding, and the row form
    is the one the layers use (so it is also what every oracle comparison above exercises)."""
    from gat_amd import CSRGraph, _lib, ops
    from gat_amd._lib import ptr
    dev = torch.device("cuda:0")
    n, F = 700, 20
    dst, src = random_graph(n, 6.0, seed=21, hubs=2, hub_deg=500)
    graph = CSRGraph.from_coo(dst, src, n, dev, chunk=128)
    p = layer_params(F, H, D, seed=22)
    tp = {k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in p.items()}
    x = torch.tensor(np.random.default_rng(23).normal(0, 1, (n, F)), dtype=torch.float32, device=dev)
    cfg = ops.make_cfg(H, D, "elu", "concat")
    dp, hd = cfg.dp, H * cfg.dp
    _, a1p, a2p, _, Wh, f1, f2, _ = ops._project(x, tp["W"], tp["a1"], tp["a2"], tp["b1"], tp["b2"], cfg)
    b2c = tp["b2"].reshape(H).contiguous()
    st = _lib.current_stream()
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    dout = torch.tensor(np.random.default_rng(24).normal(0, 1, (n, hd)), dtype=torch.float32, device=dev)

    def run(form):
        half = (ptr(f2), None, None) if form == "table" else (None, ptr(a2p), ptr(b2c))
        out, agg, pos, lse, q = new(n, hd), new(n, hd), new(n, hd), new(n, H), new(n, H)
        ws = new(max(1, _lib.lib.gatmi_gat_fwd_ws_floats(graph.n_slots_csr, H, dp)))
        _lib.call("gatmi_gat_fwd", n, H, dp, ptr(graph.items_csr), len(graph.items_csr), ptr(graph.split_csr), len(graph.split_csr),
                  ptr(graph.col), ptr(Wh), _lib.DTYPE_F32, hd, ptr(f1), *half, None, None, _lib.ACT_ELU, _lib.REDUCE_CONCAT,
                  0.2, 1.0, 1.0, 0, ptr(out), ptr(agg), ptr(pos), ptr(lse), ptr(q), ptr(ws), st)
        g, stats, df1, dbias = new(n, hd), new(n, 3 * H), new(n, H), new(hd)
        wsn = new(_lib.lib.gatmi_gat_bwd_node_ws_floats(H, dp))
        _lib.call("gatmi_gat_bwd_node", n, H, dp, ptr(dout), None, ptr(agg), ptr(pos), ptr(q), ptr(f1), ptr(lse), None,
                  _lib.ACT_ELU, _lib.REDUCE_CONCAT, 0.2, ptr(g), hd, None, 0, ptr(stats), 3 * H, ptr(df1), ptr(dbias), ptr(wsn), st)
        dfeat, df2 = new(n, hd), new(n, H)
        wsb = new(max(1, _lib.lib.gatmi_gat_bwd_ws_floats(graph.n_slots_csc, H, dp)))
        _lib.call("gatmi_gat_bwd_csc", n, H, dp, ptr(graph.items_csc), len(graph.items_csc), ptr(graph.split_csc), len(graph.split_csc),
                  ptr(graph.rowidx), None, ptr(g), _lib.DTYPE_F32, hd, ptr(stats), 3 * H, ptr(Wh), *half, ptr(df1), ptr(a1p), ptr(a2p),
                  0.2, 1.0, 1.0, 0, ptr(dfeat), ptr(df2), ptr(wsb), st)
        torch.cuda.synchronize()
        return {"out": out, "lse": lse, "q": q, "dfeat": dfeat, "df2": df2, "df1": df1}

    assert len(graph.split_csr) and len(graph.split_csc)
    table, rows = run("table"), run("rows")
    for k in table:
        assert_close(rows[k].cpu().numpy(), table[k].cpu().numpy().astype(np.float64), rtol=2e-5, atol_scale=2e-6, what="row form " + k)
    # the coefficient rows the backward pass rebuilds from the row form sum to one like the forward pass's:
    # column sums of sum_i alpha_ij g_i equal those of g only if both passes saw the same logits
    again = run("rows")
    for k in rows:
        assert torch.equal(rows[k], again[k])
    # giving both forms, or neither, is refused
    rc = _lib.lib.gatmi_gat_fwd(n, H, dp, ptr(graph.items_csr), len(graph.items_csr), None, 0, ptr(graph.col), ptr(Wh), _lib.DTYPE_F32, hd,
                                ptr(f1), ptr(f2), ptr(a2p), None, None, None, _lib.ACT_ELU, _lib.REDUCE_CONCAT, 0.2, 1.0, 1.0, 0,
                                ptr(new(n, hd)), None, None, None, None, None, st)
    assert rc == -1 and "either" in _lib.last_error()          # GATMI_EINVAL
